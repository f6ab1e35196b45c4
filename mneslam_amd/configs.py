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
