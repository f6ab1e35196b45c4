"""Hot-path configuration dictionaries.

The host application keeps its own YAML loader (the reference's config.py is out of scope); this
module only provides ready-made dicts holding the keys the mapping hot path reads, with the values
of the reference's shipped configs, so tests and bench.py need no YAML files.  Key names and
nesting are the reference's (SURVEY.md section 5 "Config / flags"):
configs/Replica/replica.yaml:11-37 (mapping), :83-89 (grid), :91-101 (pos/decoder),
:103-114 (cam), :122-142 (training), :152-167 (planes_res/c_planes_res/model),
configs/Replica/office0.yaml:3-4 (bounds).
"""
import copy

_REPLICA = {
    "dataset": "replica",
    "data": {"downsample": 1, "sc_factor": 1, "translation": 0},
    "mapping": {
        "sample": 2048, "iters": 50, "first_iters": 500, "distill_iters": 100, "loop_iters": 100,
        "lr_embed": 0.005, "lr_embed_color": 0.005, "lr_decoder": 0.01,
        "n_pixels": 0.05, "min_pixels_cur": 100, "filter_depth": False,
        "w_sdf_fs": 5, "w_sdf_center": 200, "w_sdf_tail": 30,
        "bound": [[-3, 3], [-4, 2.5], [-2, 2.5]],
        "marching_cubes_bound": [[-2.2, 2.6], [-3.4, 2.1], [-1.4, 2.0]],
    },
    "grid": {"enc": "HashGrid", "tcnn_encoding": True, "hash_size": 16, "voxel_color": 0.08,
             "voxel_sdf": 0.02, "oneGrid": True},
    "pos": {"enc": "OneBlob", "n_bins": 16},
    "decoder": {"geo_feat_dim": 15, "hidden_dim": 32, "num_layers": 2, "num_layers_color": 2,
                "hidden_dim_color": 32, "tcnn_network": False},
    "cam": {"H": 680, "W": 1200, "fx": 600.0, "fy": 600.0, "cx": 599.5, "cy": 339.5,
            "crop_edge": 0, "near": 0, "far": 10, "depth_trunc": 100.0},
    "training": {"rgb_weight": 5.0, "depth_weight": 0.1, "sdf_weight": 1200, "fs_weight": 10,
                 "eikonal_weight": 0, "smooth_weight": 0,
                 "n_samples": 256, "n_samples_d": 32, "range_d": 0.1, "n_range_d": 11,
                 "n_importance": 0, "perturb": 1, "white_bkgd": False, "trunc": 0.1},
    "planes_res": {"coarse": 0.02, "fine": 0.01, "bound_dividable": 0.02},
    "c_planes_res": {"coarse": 0.08, "fine": 0.02},
    "model": {"c_dim": 32, "truncation": 0.1, "input_ch": 64, "input_ch_pos": 48},
    "scale": 1,
    "is_co_sdf": False,
    "enable_loop_detect": False,
}


def replica_office0():
    """Replica office0 as shipped (tri-planes 0.02/0.01 m, 2x32 MLPs, 11+32 samples)."""
    return copy.deepcopy(_REPLICA)


def bench_office0(n_range_d=32, n_samples_d=96, hidden=32):
    """BASELINE.json configs[1] in its as-wired form: office0 planes,
    2048 global rays x 128 samples (n_range_d 32 + n_samples_d 96)."""
    cfg = replica_office0()
    cfg["training"]["n_range_d"] = n_range_d
    cfg["training"]["n_samples_d"] = n_samples_d
    cfg["decoder"]["hidden_dim"] = hidden
    cfg["decoder"]["hidden_dim_color"] = hidden
    return cfg


def bench_office0_hash(hidden=64, hash_size=19, desired_resolution=512):
    """BASELINE.json configs[1] in its LITERAL form (SURVEY.md section 8d, C2 as-north-star): the factory defaults of
    ``get_encoder('HashGrid')`` (model/encodings.py:6-10: 16 levels x 2 features, base 16, T = 2^19, finest 512:
    10,492,048 table entries' floats) + 2x64 MLPs, 2048 global rays x 128 samples.  ``scene_encoding: hash`` selects
    the wiring the reference keeps commented out (model/scene_rep.py:160); ``grid.hash_size`` is then a live key."""
    cfg = bench_office0(hidden=hidden)
    cfg["scene_encoding"] = "hash"
    cfg["grid"]["hash_size"] = hash_size
    cfg["grid"]["desired_resolution"] = desired_resolution
    return cfg


def _overlay(base, over):
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            _overlay(base[k], v)
        else:
            base[k] = copy.deepcopy(v)
    return base


# configs/Replica/apart1_agent{1,2}.yaml:3-4 (inherit replica.yaml; only the bounds differ)
_APART1_BOUNDS = {1: [[-2.8, 8.2], [-1.1, 7.2], [-2, 1.3]], 2: [[-2.8, 7.0], [1.0, 9.0], [-2.5, 1.3]]}


def apartment_agent(agent=1):
    """Replica Apart-1 split into two agents (BASELINE configs[2]): replica.yaml + the agent's bound
    (configs/Replica/apart1_agent1.yaml, apart1_agent2.yaml); agent 1: 62.4 M plane parameters."""
    cfg = replica_office0()
    b = _APART1_BOUNDS[agent]
    cfg["mapping"]["bound"] = copy.deepcopy(b)
    cfg["mapping"]["marching_cubes_bound"] = copy.deepcopy(b)
    cfg["enable_loop_detect"] = True
    return cfg


# configs/ScanNet/scannet.yaml (values that differ from replica.yaml) + configs/ScanNet/scene0000.yaml
_SCANNET = {
    "dataset": "scannet",
    "mapping": {"min_pixels_cur": 20, "w_sdf_tail": 10,
                "bound": [[-0.1, 8.6], [-0.1, 8.9], [-0.3, 3.3]],
                "marching_cubes_bound": [[-0.1, 8.6], [-0.1, 8.9], [-0.3, 3.3]]},
    "grid": {"hash_size": 19, "voxel_sdf": 0.04, "oneGrid": False},
    "cam": {"H": 480, "W": 640, "fx": 577.590698, "fy": 578.729797, "cx": 318.905426, "cy": 242.683609,
            "crop_edge": 10, "near": 0, "far": 8, "depth_trunc": 100.0},
    "training": {"sdf_weight": 1000, "smooth_weight": 0.001, "n_samples_d": 96, "range_d": 0.25, "n_range_d": 21},
}


def scannet_scene0000(hidden=32):
    """ScanNet scene0000_00 (BASELINE configs[3]): colour planes (oneGrid False), 21 + 96 = 117 samples,
    480x640 frames cropped to 460x620, no ``training.n_samples`` key (configs/ScanNet/scannet.yaml:115)."""
    cfg = _overlay(replica_office0(), _SCANNET)
    del cfg["training"]["n_samples"]
    cfg["decoder"]["hidden_dim"] = cfg["decoder"]["hidden_dim_color"] = hidden
    return cfg


# configs/Indoor/indoor.yaml + indoor_agent{0..3}.yaml (bounds = loop_bound.bound_k, indoor.yaml:169-173)
_INDOOR_BOUNDS = {0: [[-6.2, 20], [-15.8, 0], [-1.0, 4.5]], 1: [[-6.2, 56.4], [-15.8, -7.0], [-1.0, 4.5]],
                  2: [[25.0, 56.4], [-13.5, -2.0], [-2.0, 4.5]], 3: [[-6.2, 50.0], [-6.5, -2.2], [-2.0, 4.5]]}
_INDOOR = {
    "dataset": "indoor",
    "mapping": {"iters": 100, "lr_embed": 0.01, "lr_embed_color": 0.01, "w_sdf_fs": 10, "w_sdf_tail": 50},
    "cam": {"H": 720, "W": 1280, "fx": 637.147, "fy": 636.668, "cx": 637.003, "cy": 363.032,
            "crop_edge": 0, "near": 0, "far": 60.0, "depth_trunc": 100.0},
    "training": {"sdf_weight": 1000, "smooth_weight": 0.001, "n_samples": 512, "n_samples_d": 1024,
                 "range_d": 0.2, "n_range_d": 21},
    "planes_res": {"coarse": 0.24, "fine": 0.06, "bound_dividable": 0.24},
    "c_planes_res": {"coarse": 0.24, "fine": 0.06},
    "loop_bound": {f"bound_{k}": v for k, v in _INDOOR_BOUNDS.items()},
}


def indoor_agent(agent=0):
    """INS Indoor agent ``agent`` (BASELINE configs[4]): 21 + 1024 = 1045 samples per ray, far 60 m,
    planes 0.24 / 0.06 m on the agent's bound (configs/Indoor/indoor.yaml:21-23,135-139,169-173;
    configs/Indoor/indoor_agent0.yaml)."""
    cfg = _overlay(replica_office0(), _INDOOR)
    b = _INDOOR_BOUNDS[agent]
    cfg["mapping"]["bound"] = copy.deepcopy(b)
    cfg["mapping"]["marching_cubes_bound"] = copy.deepcopy(b)
    cfg["enable_loop_detect"] = True
    return cfg


WORKLOADS = {
    # name -> (config factory, workload label used in bench.py's config.workload)
    "office0": (lambda hidden=32: bench_office0(hidden=hidden), "replica_office0_triplane_asWired_2048x128"),
    "office0_asShipped": (lambda hidden=32: _overlay(replica_office0(), {"decoder": {"hidden_dim": hidden, "hidden_dim_color": hidden}}),
                          "replica_office0_triplane_asShipped_2048x43"),
    "apartment": (lambda hidden=32: _overlay(apartment_agent(1), {"decoder": {"hidden_dim": hidden, "hidden_dim_color": hidden}}),
                  "replica_apart1_agent1_triplane_2048x43"),
    "scannet": (lambda hidden=32: scannet_scene0000(hidden), "scannet_scene0000_colorplanes_2048x117"),
    "office0_hash": (lambda hidden=64: bench_office0_hash(hidden=hidden), "replica_office0_hashT19_2x64_2048x128"),
    "indoor": (lambda hidden=32: _overlay(indoor_agent(0), {"decoder": {"hidden_dim": hidden, "hidden_dim_color": hidden}}),
               "ins_indoor_agent0_triplane_2048x1045"),
    # BASELINE configs[4] as worded (one of its agents): fp16 feature storage + fp32 accumulate (EXTENSION: grid.plane_dtype);
    # bench.py --graph adds the hipGraph-captured iteration
    "indoor_fp16": (lambda hidden=32: _overlay(indoor_agent(0), {"decoder": {"hidden_dim": hidden, "hidden_dim_color": hidden},
                                                                 "grid": {"plane_dtype": "fp16"}}),
                    "ins_indoor_agent0_triplane_fp16planes_2048x1045"),
    "office0_fp16": (lambda hidden=32: _overlay(bench_office0(hidden=hidden), {"grid": {"plane_dtype": "fp16"}}),
                     "replica_office0_triplane_fp16planes_2048x128"),
}


# ---- EXTENSION: BASELINE configs[2..4] as worded -- ONE scene split over N agents (mneslam_amd/dist.py) ----
# the scene each multi-agent workload covers as a whole: the union of the reference's per-agent bounds
SCENE_BOUNDS = {
    "apartment": [[-2.8, 8.2], [-1.1, 9.0], [-2.5, 1.3]],          # apart1_agent1.yaml U apart1_agent2.yaml
    "scannet": [[-0.1, 8.6], [-0.1, 8.9], [-0.3, 3.3]],            # scene0000.yaml
    "indoor": [[-6.2, 56.4], [-15.8, 0.0], [-2.0, 4.5]],           # indoor.yaml:169-173, bound_0 U .. U bound_3
}


def split_agent_config(cfg, n_agents, rank, overlap=0.5, axis=None):
    """Agent ``rank`` of ``n_agents`` mapping ONE scene (``cfg["mapping"]["bound"]`` = the whole scene): slabs along the
    scene's longest axis that overlap their neighbours by about ``overlap`` metres (the reference's agents overlap too:
    mp_slam/mapper.py:491-509, configs/Indoor/indoor.yaml:169-173), all on ONE lattice -- every slab edge on a multiple of
    the coarsest plane cell from the scene's lower corner, ``planes_res.lattice`` for exact node spacing -- so that the
    cells two neighbours both hold coincide and their gradients can be exchanged (FusedStep(overlap_peers=...)).
    Returns (agent config, slab axis, [extended bound of every agent])."""
    from . import dist as mdist
    cfg = copy.deepcopy(cfg)
    res = [cfg["planes_res"]["coarse"], cfg["planes_res"]["fine"]]
    if not cfg["grid"]["oneGrid"]:
        res += [cfg["c_planes_res"]["coarse"], cfg["c_planes_res"]["fine"]]
    cell = max(res)
    for r in res:
        if abs(cell / r - round(cell / r)) > 1e-6:
            raise ValueError(f"plane resolutions {res} do not nest: no common lattice")
    scene = [[float(lo), float(hi)] for lo, hi in cfg["mapping"]["bound"]]
    ext = [[lo, lo + -(-(hi - lo) // cell) * cell] for lo, hi in scene]          # whole cells per axis
    ext = [[lo, lo + round((hi - lo) / cell) * cell] for lo, hi in ext]
    if axis is None:
        axis = max(range(3), key=lambda k: ext[k][1] - ext[k][0])
    slabs = mdist.aligned_agent_bounds(ext, n_agents, axis, overlap, cell)
    lo_hi = slabs[rank]
    # raw bound whose extension by load_bound (scene_rep.py:72-83: (int(len / bd) + 1) * bd) is exactly the slab
    cfg["mapping"]["bound"] = [[lo, hi - 0.5 * cell] for lo, hi in lo_hi]
    room = [[lo + min(0.3, 0.2 * (hi - lo)), hi - 0.5 * cell - min(0.3, 0.2 * (hi - lo))] for lo, hi in lo_hi]
    cfg["mapping"]["marching_cubes_bound"] = room
    cfg["planes_res"]["bound_dividable"] = cell
    cfg["planes_res"]["lattice"] = True
    return cfg, axis, slabs


def small_test_config(one_grid=True, is_co_sdf=False, n_samples_d=32, n_range_d=11, depth_trunc=100.0):
    """The reduced configuration of the golden fixtures (tests/golden/make_golden.py::small_config)."""
    cfg = replica_office0()
    cfg["mapping"]["bound"] = [[-1.0, 1.0], [-1.2, 1.1], [-0.8, 0.9]]
    cfg["mapping"]["marching_cubes_bound"] = [[-0.8, 0.8], [-1.0, 0.9], [-0.6, 0.7]]
    cfg["planes_res"] = {"coarse": 0.2, "fine": 0.1, "bound_dividable": 0.2}
    cfg["c_planes_res"] = {"coarse": 0.4, "fine": 0.2}
    cfg["grid"]["oneGrid"] = one_grid
    cfg["is_co_sdf"] = is_co_sdf
    cfg["cam"]["far"] = 4.0
    cfg["cam"]["depth_trunc"] = depth_trunc
    cfg["training"]["n_samples_d"] = n_samples_d
    cfg["training"]["n_range_d"] = n_range_d
    cfg["training"]["n_samples"] = 48
    return cfg
