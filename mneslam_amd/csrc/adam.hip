// adam.hip -- single-pass dense Adam (+ gradient zeroing) over any number of tensors in one
// launch.  Pure HBM streaming: per parameter it reads p, g, m, v and writes p, m, v and g = 0
// (32 B), the dominant byte count of a mapping iteration.
//
// Arithmetic = torch.optim.Adam (single-tensor form, torch >= 2; mneslam_mp.py:459-469 groups):
//   g += wd * p;  m = m + (g - m) * (1 - b1);  v = v * b2 + (1 - b2) * g * g;
//   p = p - step_size * (m / (sqrt(v) / bc2_sqrt + eps)),   step_size = lr / (1 - b1^t).
#include "mne_device.h"
#include "mne_launch.h"

#define ADAM_THREADS 256
#define ADAM_VEC_PER_THREAD 4
#define ADAM_ELEMS_PER_BLOCK (ADAM_THREADS * 4 * ADAM_VEC_PER_THREAD)

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, float wd, float omb1, float b2,
                                         float omb2, float eps, float step_size, float bc2_sqrt) {
    float gg = g;
    if (wd != 0.0f) gg = gg + wd * p;
    m = m + (gg - m) * omb1;
    v = v * b2 + omb2 * (gg * gg);
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(ADAM_THREADS) void adam_kernel(AdamArgs a) {
    int s = 0;
    const long long blk = blockIdx.x;
    while (s + 1 < a.n_seg && blk >= a.blk_start[s + 1]) ++s;
    const mne_adam_seg_t& sg = a.seg[s];
    const float wd = (float)sg.weight_decay, omb1 = (float)(1.0 - sg.beta1), b2 = (float)sg.beta2;
    const float omb2 = (float)(1.0 - sg.beta2), eps = (float)sg.eps;
    float step_size = a.step_size[s], bc2_sqrt = a.bc2_sqrt[s];
    if (a.clk.bias_table) clock_bias(a.clk, sg.lr, sg.step, step_size, bc2_sqrt);        // graph replay: step from device memory
    const long long base = (blk - a.blk_start[s]) * ADAM_ELEMS_PER_BLOCK;
#pragma unroll
    for (int it = 0; it < ADAM_VEC_PER_THREAD; ++it) {
        const long long i = base + ((long long)it * ADAM_THREADS + threadIdx.x) * 4;
        if (i + 3 < sg.n) {
            // parameters: fp32, or half precision (p_f16: p32 = float(p16) -> Adam in fp32 -> round to nearest; g, m, v fp32)
            float4 p = sg.p_f16 ? half4_to_float4(*(const uint2*)((const _Float16*)sg.p + i)) : *(const float4*)((const float*)sg.p + i);
            float4 g = *(const float4*)(sg.g + i);
            float4 m = *(float4*)(sg.m + i), v = *(float4*)(sg.v + i);
            adam_one(p.x, g.x, m.x, v.x, wd, omb1, b2, omb2, eps, step_size, bc2_sqrt);
            adam_one(p.y, g.y, m.y, v.y, wd, omb1, b2, omb2, eps, step_size, bc2_sqrt);
            adam_one(p.z, g.z, m.z, v.z, wd, omb1, b2, omb2, eps, step_size, bc2_sqrt);
            adam_one(p.w, g.w, m.w, v.w, wd, omb1, b2, omb2, eps, step_size, bc2_sqrt);
            if (sg.p_f16) *(uint2*)((_Float16*)sg.p + i) = float4_to_half4(p);
            else *(float4*)((float*)sg.p + i) = p;
            *(float4*)(sg.m + i) = m; *(float4*)(sg.v + i) = v;
            if (a.zero_grad) *(float4*)(sg.g + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (long long j = i; j < sg.n && j < i + 4; ++j) {
                float p = sg.p_f16 ? (float)((const _Float16*)sg.p)[j] : ((const float*)sg.p)[j];
                float g = sg.g[j], m = sg.m[j], v = sg.v[j];
                adam_one(p, g, m, v, wd, omb1, b2, omb2, eps, step_size, bc2_sqrt);
                if (sg.p_f16) ((_Float16*)sg.p)[j] = (_Float16)p; else ((float*)sg.p)[j] = p;
                sg.m[j] = m; sg.v[j] = v;
                if (a.zero_grad) sg.g[j] = 0.0f;
            }
        }
    }
}

int mne_launch_adam(const AdamArgs& a, hipStream_t st) {
    const long long nblk = a.blk_start[a.n_seg];
    if (nblk <= 0) return 0;
    MNE_LAUNCH(adam_kernel, (unsigned)nblk, ADAM_THREADS, 0, st, a);
    return 0;
}

__global__ void clock_advance_kernel(unsigned long long* iteration, int* step_offset) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (iteration) *iteration += 1ull;
        if (step_offset) *step_offset += 1;
    }
}

int mne_launch_clock_advance(unsigned long long* iteration, int* step_offset, hipStream_t st) {
    MNE_LAUNCH(clock_advance_kernel, 1, 64, 0, st, iteration, step_offset);
    return 0;
}

long long mne_adam_blocks_for(long long n) { return (n + ADAM_ELEMS_PER_BLOCK - 1) / ADAM_ELEMS_PER_BLOCK; }
