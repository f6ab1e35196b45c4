// encodings.hip -- the three remaining branches of the reference's encoder factory (model/encodings.py:48-58, 73-95):
// tcnn.Encoding(otype = "SphericalHarmonics" | "Frequency" | "Identity"), forward and the gradient with respect to the
// input.  None of them is reached by the reference's mapping path (model/scene_rep.py:157 requests OneBlob) -- they are here
// so that get_encoder() covers the reference's whole surface.  The arithmetic is tinycudann's (not in the reference tree:
// PARITY UNPINNED); frozen spec and source map: oracle/encodings_misc.py.  One thread per (point, output group); memory-bound
// elementwise kernels.
#include "mne_launch.h"

#define ENC_PI 3.14159265358979323846f

// Frequency: out[j], j in [0, dims * 2 F): dim = j / (2 F), log2 frequency = (j / 2) % F, phase = (j % 2) * pi / 2;
// out = sin(2^f * x * pi + phase)            (tiny-cuda-nn encodings/frequency.h, kernel_frequency)
__global__ __launch_bounds__(256) void frequency_kernel(long long n_out, int dims, int F, const float* x, float* out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_out) return;
    const int per = dims * 2 * F;
    const long long i = t / per;
    const int j = (int)(t % per);
    const int dim = j / (2 * F), f = (j / 2) % F;
    const float phase = (float)(j & 1) * (ENC_PI / 2.0f);
    const float v = scalbnf(x[i * dims + dim], f);
    out[t] = sinf(v * ENC_PI + phase);
}
// d(total)/dx[dim] = sum over the dim's 2 F outputs of dout * cos(arg) * 2^f * pi      (kernel_frequency_backward)
__global__ __launch_bounds__(256) void frequency_backward_kernel(long long n_in, int dims, int F, const float* x, const float* dout, float* dx) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_in) return;
    const long long i = t / dims;
    const int dim = (int)(t % dims);
    const float xv = x[t];
    const float* d = dout + i * (long long)(dims * 2 * F) + (long long)dim * 2 * F;
    float acc = 0.0f;
    for (int j = 0; j < 2 * F; ++j) {
        const int f = j / 2;
        const float phase = (float)(j & 1) * (ENC_PI / 2.0f);
        const float arg = scalbnf(xv, f) * ENC_PI + phase;
        acc += d[j] * (cosf(arg) * scalbnf(ENC_PI, f));
    }
    dx[t] = acc;
}

// Spherical harmonics of degree <= 4 (16 coefficients) of the direction 2 * in - 1      (encodings/spherical_harmonics.h, sh_enc)
struct Sh16 { float v[16]; };
__device__ __forceinline__ Sh16 sh_values(float x, float y, float z) {
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    Sh16 o;
    o.v[0] = 0.28209479177387814f;
    o.v[1] = -0.48860251190291987f * y;
    o.v[2] = 0.48860251190291987f * z;
    o.v[3] = -0.48860251190291987f * x;
    o.v[4] = 1.0925484305920792f * xy;
    o.v[5] = -1.0925484305920792f * yz;
    o.v[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o.v[7] = -1.0925484305920792f * xz;
    o.v[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o.v[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o.v[10] = 2.8906114426405538f * xy * z;
    o.v[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o.v[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o.v[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o.v[14] = 1.4453057213202769f * z * (x2 - y2);
    o.v[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
    return o;
}
__global__ __launch_bounds__(256) void sh_kernel(long long n, int n_coef, const float* in, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Sh16 o = sh_values(in[i * 3] * 2.0f - 1.0f, in[i * 3 + 1] * 2.0f - 1.0f, in[i * 3 + 2] * 2.0f - 1.0f);
    for (int k = 0; k < n_coef; ++k) out[i * n_coef + k] = o.v[k];
}
__global__ __launch_bounds__(256) void sh_backward_kernel(long long n, int n_coef, const float* in, const float* dout, float* din) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i * 3] * 2.0f - 1.0f, y = in[i * 3 + 1] * 2.0f - 1.0f, z = in[i * 3 + 2] * 2.0f - 1.0f;
    const float x2 = x * x, y2 = y * y, z2 = z * z;
    float g[16];
    for (int k = 0; k < 16; ++k) g[k] = k < n_coef ? dout[i * n_coef + k] : 0.0f;
    const float c1 = 0.48860251190291987f, c2 = 1.0925484305920792f, c3 = 0.54627421529603959f, c4 = 0.59004358992664352f;
    const float c5 = 2.8906114426405538f, c6 = 0.45704579946446572f, c7 = 1.4453057213202769f, c8 = 0.3731763325901154f;
    const float dxv = -c1 * g[3] + c2 * y * g[4] - c2 * z * g[7] + 2.0f * c3 * x * g[8] - 6.0f * c4 * x * y * g[9] + c5 * y * z * g[10]
                      + c6 * (1.0f - 5.0f * z2) * g[13] + 2.0f * c7 * x * z * g[14] + c4 * (-3.0f * x2 + 3.0f * y2) * g[15];
    const float dyv = -c1 * g[1] + c2 * x * g[4] - c2 * z * g[5] - 2.0f * c3 * y * g[8] + c4 * (-3.0f * x2 + 3.0f * y2) * g[9] + c5 * x * z * g[10]
                      + c6 * (1.0f - 5.0f * z2) * g[11] - 2.0f * c7 * y * z * g[14] + 6.0f * c4 * x * y * g[15];
    const float dzv = c1 * g[2] - c2 * y * g[5] + 2.0f * 0.94617469575755997f * z * g[6] - c2 * x * g[7] + c5 * x * y * g[10]
                      - 10.0f * c6 * y * z * g[11] + c8 * (15.0f * z2 - 3.0f) * g[12] - 10.0f * c6 * x * z * g[13] + c7 * (x2 - y2) * g[14];
    din[i * 3] = 2.0f * dxv; din[i * 3 + 1] = 2.0f * dyv; din[i * 3 + 2] = 2.0f * dzv;       // d(2 in - 1) / d in
}

// Identity: out = in * scale + offset      (encodings/identity.h)
__global__ __launch_bounds__(256) void identity_kernel(long long n_elems, float scale, float offset, const float* x, float* out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_elems) out[t] = x[t] * scale + offset;
}

static unsigned blocks_for(long long n) { return (unsigned)((n + 255) / 256); }

int mne_launch_frequency(long long n, int dims, int F, const float* x, float* out, hipStream_t st) {
    const long long n_out = n * dims * 2 * F;
    MNE_LAUNCH(frequency_kernel, blocks_for(n_out), 256, 0, st, n_out, dims, F, x, out);
    return 0;
}
int mne_launch_frequency_backward(long long n, int dims, int F, const float* x, const float* dout, float* dx, hipStream_t st) {
    MNE_LAUNCH(frequency_backward_kernel, blocks_for(n * dims), 256, 0, st, n * dims, dims, F, x, dout, dx);
    return 0;
}
int mne_launch_sh(long long n, int n_coef, const float* in, float* out, hipStream_t st) {
    MNE_LAUNCH(sh_kernel, blocks_for(n), 256, 0, st, n, n_coef, in, out);
    return 0;
}
int mne_launch_sh_backward(long long n, int n_coef, const float* in, const float* dout, float* din, hipStream_t st) {
    MNE_LAUNCH(sh_backward_kernel, blocks_for(n), 256, 0, st, n, n_coef, in, dout, din);
    return 0;
}
int mne_launch_identity(long long n_elems, float scale, float offset, const float* x, float* out, hipStream_t st) {
    MNE_LAUNCH(identity_kernel, blocks_for(n_elems), 256, 0, st, n_elems, scale, offset, x, out);
    return 0;
}
