// mne_sampler.h -- device side of the per-iteration ray batch (shared by sampler.hip and the fused batch kernel of
// render.hip); see sampler.hip for the reference semantics.
#pragma once
#include "mne_device.h"
#include "mne_launch.h"

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint64_t feistel_index(uint64_t i, uint64_t n, int half_bits, uint64_t key) {
    const uint32_t mask = (1u << half_bits) - 1u;
    uint64_t x = i;
    do {
        uint32_t l = (uint32_t)(x >> half_bits) & mask, r = (uint32_t)x & mask;
#pragma unroll
        for (int round = 0; round < 4; ++round) {
            const uint32_t f = mix32(r ^ (uint32_t)(key >> (16 * round)) ^ (0x9E3779B9u * (round + 1)) ^ (uint32_t)(key >> 32)) & mask;
            const uint32_t nl = r;
            r = l ^ f;
            l = nl;
        }
        x = ((uint64_t)l << half_bits) | r;
    } while (x >= n);
    return x;
}

// two independent 64-bit keys from (seed, iteration) -- splitmix64; host and device run the same integer arithmetic
__host__ __device__ inline void ray_keys(unsigned long long seed, unsigned long long iteration, unsigned long long& key_kf,
                                         unsigned long long& key_cur) {
    unsigned long long z = seed * 0x9E3779B97F4A7C15ull + iteration * 0xD1B54A32D192ED03ull + 0x632BE59BD9B4E019ull;
    unsigned long long k[2];
    for (int i = 0; i < 2; ++i) {
        z += 0x9E3779B97F4A7C15ull;
        unsigned long long x = z;
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        k[i] = x ^ (x >> 31);
    }
    key_kf = k[0]; key_cur = k[1];
}

// Ray t of the batch: index draw, pose rotation; `write`: this thread stores the ray (rays_o, rays_d, targets, out_idx).
// Returns the target depth.
__device__ __forceinline__ float sample_ray(const SampleRaysArgs& a, int t, bool write) {
    unsigned long long key_kf, key_cur;
    ray_keys(a.seed, a.iteration + (a.clk.iteration ? *a.clk.iteration : 0ull), key_kf, key_cur);
    const float* src;
    int pose_id;
    long long idx;
    if (t < a.n_global) {
        idx = a.idx_global ? a.idx_global[t] : (long long)feistel_index((uint64_t)t, (uint64_t)a.n_kf_rays, a.half_bits_kf, key_kf);
        src = a.kf_rays + idx * 7;
        pose_id = a.kf_pose_ids ? a.kf_pose_ids[idx / a.n_save] : (int)(idx / a.n_save);
    } else {
        const int j = t - a.n_global;
        idx = a.idx_cur ? a.idx_cur[j] : (long long)feistel_index((uint64_t)j, (uint64_t)a.n_cur_rays, a.half_bits_cur, key_cur);
        src = a.cur_rays + idx * 7;
        pose_id = a.n_poses - 1;                      // id -1 in the reference: poses[-1]
    }
    const float depth = src[6];
    if (write) {
        if (a.out_idx) a.out_idx[t] = idx;
        const float* P = a.poses + (size_t)pose_id * 16;  // row-major 4x4 c2w
        const float d0 = src[0], d1 = src[1], d2 = src[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            a.rays_d[t * 3 + j] = (d0 * P[j * 4 + 0] + d1 * P[j * 4 + 1]) + d2 * P[j * 4 + 2];
            a.rays_o[t * 3 + j] = P[j * 4 + 3];
        }
        a.target_rgb[t * 3 + 0] = src[3]; a.target_rgb[t * 3 + 1] = src[4]; a.target_rgb[t * 3 + 2] = src[5];
        a.target_d[t] = depth;
    }
    return depth;
}

