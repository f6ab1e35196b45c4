// sampler.hip -- per-iteration ray batch on the device (R1/R2 of SURVEY.md section 8a):
// `n_global` rows drawn WITHOUT replacement from the keyframe ray database + `n_cur` pixels of the
// current frame (also without replacement), each rotated by its owning pose
//   rays_d = sum_k dir_cam[k] * c2w[:3,k],  rays_o = c2w[:3,3]        (mp_slam/mapper.py:151-153).
// Reference: KeyFrameDatabase.sample_global_rays (model/keyframe.py:91-103) + Mapper.mapping_optimize
// (mp_slam/mapper.py:135-148), which use python `random.sample` on the host.
//
// Sampling without replacement needs no state here: a keyed 4-round Feistel network is a bijection
// on [0, 2^k); cycle-walking restricts it to a bijection on [0, N), so the images of 0..n-1 are n
// distinct uniformly scrambled indices.  Explicit index arrays (e.g. the host RNG's draws) can be
// supplied instead, which makes the assembly bit-comparable with the reference.
#include "mne_sampler.h"

__global__ __launch_bounds__(256) void sample_rays_kernel(SampleRaysArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_global + a.n_cur) return;
    sample_ray(a, t, true);
}

int mne_half_bits_for(long long n) {
    int bits = 2;
    while ((1ll << bits) < n) ++bits;
    return (bits + 1) / 2;          // even total width >= bits
}

int mne_launch_sample_rays(SampleRaysArgs a, unsigned long long seed, unsigned long long iteration, hipStream_t st) {
    a.half_bits_kf = mne_half_bits_for(a.n_kf_rays);
    a.half_bits_cur = mne_half_bits_for(a.n_cur_rays);
    a.seed = seed; a.iteration = iteration;
    const int R = a.n_global + a.n_cur;
    MNE_LAUNCH(sample_rays_kernel, (R + 255) / 256, 256, 0, st, a);
    return 0;
}
