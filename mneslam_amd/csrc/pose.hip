// pose.hip -- the pose-alignment loop of loop closure (N2) as device work only.
//
// Reference: the loop inside Mapper.handle_loop_closure, mp_slam/mapper.py:362-412 -- per iteration
//     c2w   = SLAM.matrix_from_tensor(cur_rot, cur_trans)                 (optimization/utils.py:161-197, axis-angle)
//     rays  = dirs @ c2w[:3,:3]^T, origin c2w[:3,3]                        (mp_slam/mapper.py:388-392)
//     out   = model.render_rays(rays_o, rays_d, target_d=None)
//     loss  = w_rgb * mse(out.rgb, teacher.rgb) + w_depth * mse(out.depth, teacher.depth)
//     loss.backward(); pose_optimizer.step()                               (torch.optim.Adam, two groups: lr_rot, lr_trans)
// Here the ray gradients come from mne_render_backward (R13) and everything around them is three small kernels:
//   pose_rays_kernel    parameters -> c2w, rays
//   pose_loss_kernel    maps -> d(loss)/d(maps) + per-workgroup loss partials
//   pose_update_kernel  ray gradients -> d/d(c2w) -> d/d(rot, trans) (analytic Jacobian of the axis-angle map),
//                       best-pose tracking, Adam on the six parameters with the step count in device memory
// so that an iteration is six launches with no autograd graph, no torch.optim step and no host synchronisation.
//
// Parameterisation: R = Rot(rot) * R_base with a constant R_base (identity for the reference; a host that optimises a
// rotation relative to the initial pose passes that pose's rotation).  Rot = Rodrigues' formula for the reference's
// rot_rep 'axis_angle' (optimization/utils.py:161-177) or the normalising quaternion map of rot_rep 'quat'
// (pytorch3d.transforms.quaternion_to_matrix behind optimization/utils.py:199-210; real part first).
#include "mne_device.h"
#include "mne_launch.h"

// R = I + sin(t) K + (1 - cos(t)) K K,  K = cross-product matrix of rot / t,  t = |rot|   (optimization/utils.py:161-177)
__device__ __forceinline__ void rodrigues(const float w[3], float R[9], float K[9], float& th) {
    th = sqrtf(w[0] * w[0] + w[1] * w[1] + w[2] * w[2] + 1e-24f);          // (the reference divides by zero at rot = 0)
    const float o0 = w[0] / th, o1 = w[1] / th, o2 = w[2] / th;
    K[0] = 0.f; K[1] = -o2; K[2] = o1;
    K[3] = o2; K[4] = 0.f; K[5] = -o0;
    K[6] = -o1; K[7] = o0; K[8] = 0.f;
    const float s = sinf(th), c1 = 1.0f - cosf(th);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float kk = 0.f;
            for (int k = 0; k < 3; ++k) kk += K[i * 3 + k] * K[k * 3 + j];
            R[i * 3 + j] = (i == j ? 1.0f : 0.0f) + s * K[i * 3 + j] + c1 * kk;
        }
}

// R = I + s A(q), s = 2 / (q . q)   (pytorch3d quaternion_to_matrix; q = (r, i, j, k))
__device__ __forceinline__ void quat_matrix(const float q[4], float R[9], float A[9], float& s) {
    const float r = q[0], i = q[1], j = q[2], k = q[3];
    s = 2.0f / (r * r + i * i + j * j + k * k);
    A[0] = -(j * j + k * k); A[1] = i * j - k * r;    A[2] = i * k + j * r;
    A[3] = i * j + k * r;    A[4] = -(i * i + k * k); A[5] = j * k - i * r;
    A[6] = i * k - j * r;    A[7] = j * k + i * r;    A[8] = -(i * i + j * j);
    for (int e = 0; e < 9; ++e) R[e] = ((e % 4 == 0) ? 1.0f : 0.0f) + s * A[e];
}

__device__ __forceinline__ void pose_matrix(const PoseArgs& a, float M[9]) {
    float R[9], K[9], th;
    if (a.n_rot == 4) {
        const float q[4] = {a.rot[0], a.rot[1], a.rot[2], a.rot[3]};
        quat_matrix(q, R, K, th);
    } else {
        const float w[3] = {a.rot[0], a.rot[1], a.rot[2]};
        rodrigues(w, R, K, th);
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0.f;
            for (int k = 0; k < 3; ++k) s += R[i * 3 + k] * a.r_base[k * 3 + j];
            M[i * 3 + j] = s;
        }
}

__global__ __launch_bounds__(256) void pose_rays_kernel(PoseArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float M[9];
    pose_matrix(a, M);
    if (t == 0)
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) a.c2w[i * 4 + j] = M[i * 3 + j];
            a.c2w[i * 4 + 3] = a.trans[i];
        }
    if (t >= a.n) return;
    const float d0 = a.dirs[t * 3], d1 = a.dirs[t * 3 + 1], d2 = a.dirs[t * 3 + 2];
    for (int i = 0; i < 3; ++i) {
        // rays_d = sum_k dir[k] * c2w[i][k]   (mp_slam/mapper.py:391, the same sum the mapping sampler forms)
        a.rays_d[t * 3 + i] = __fadd_rn(__fadd_rn(__fmul_rn(d0, M[i * 3]), __fmul_rn(d1, M[i * 3 + 1])), __fmul_rn(d2, M[i * 3 + 2]));
        a.rays_o[t * 3 + i] = a.trans[i];
    }
}

// loss = w_rgb * mean((rgb - want)^2 over n*3) + w_depth * mean((depth - want)^2 over n)  (F.mse_loss, reduction mean)
__global__ __launch_bounds__(256) void pose_loss_kernel(PoseArgs a) {
    __shared__ float red[256];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float part = 0.f;
    if (t < a.n) {
        const float cr = a.w_rgb / (3.0f * (float)a.n), cd = a.w_depth / (float)a.n;
        for (int k = 0; k < 3; ++k) {
            const float e = a.rgb[t * 3 + k] - a.want_rgb[t * 3 + k];
            a.d_rgb[t * 3 + k] = 2.0f * cr * e;
            part += cr * e * e;
        }
        const float e = a.depth[t] - a.want_depth[t];
        a.d_depth[t] = 2.0f * cd * e;
        part += cd * e * e;
    }
    red[threadIdx.x] = part;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) a.partials[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void pose_update_kernel(PoseArgs a) {
    __shared__ float red[12][256];          // 9 entries of G = sum_r d_rays_d (x) dir, 3 of sum_r d_rays_o
    __shared__ float lred[256];
    const int tid = threadIdx.x;
    float acc[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int r = tid; r < a.n; r += 256) {
        const float d0 = a.dirs[r * 3], d1 = a.dirs[r * 3 + 1], d2 = a.dirs[r * 3 + 2];
        for (int i = 0; i < 3; ++i) {
            const float g = a.d_rays_d[r * 3 + i];
            acc[i * 3] += g * d0; acc[i * 3 + 1] += g * d1; acc[i * 3 + 2] += g * d2;
            acc[9 + i] += a.d_rays_o[r * 3 + i];
        }
    }
    for (int k = 0; k < 12; ++k) red[k][tid] = acc[k];
    float lp = 0.f;
    for (int p = tid; p < a.n_partials; p += 256) lp += a.partials[p];
    lred[tid] = lp;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            for (int k = 0; k < 12; ++k) red[k][tid] += red[k][tid + s];
            lred[tid] += lred[tid + s];
        }
        __syncthreads();
    }
    if (tid != 0) return;
    // ---- best pose so far: the pose the loss was evaluated at (mp_slam/mapper.py:399-403)
    const float loss = lred[0];
    *a.last_loss = loss;
    if (loss < *a.best_loss) {
        *a.best_loss = loss;
        for (int k = 0; k < 12; ++k) a.best_c2w[k] = a.c2w[k];
    }
    // ---- d/dM (M = c2w rotation) -> d/dR = dM * R_base^T -> d/d(rot) through the axis-angle map
    float GM[9], G[9];
    for (int k = 0; k < 9; ++k) GM[k] = red[k][0];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            float s = 0.f;
            for (int k = 0; k < 3; ++k) s += GM[i * 3 + k] * a.r_base[j * 3 + k];
            G[i * 3 + j] = s;
        }
    float grad[7];
    const int nr = a.n_rot;
    if (nr == 4) {
        // d/dq of R = I + s A(q):  s <G, dA/dq_m> - s^2 q_m <G, A>
        const float q[4] = {a.rot[0], a.rot[1], a.rot[2], a.rot[3]};
        float R[9], A[9], sc;
        quat_matrix(q, R, A, sc);
        const float r = q[0], i = q[1], j = q[2], k = q[3];
        float ga = 0.f;
        for (int e = 0; e < 9; ++e) ga += G[e] * A[e];
        const float d[4] = {
            -k * G[1] + j * G[2] + k * G[3] - i * G[5] - j * G[6] + i * G[7],
            j * G[1] + k * G[2] + j * G[3] - 2.0f * i * G[4] - r * G[5] + k * G[6] + r * G[7] - 2.0f * i * G[8],
            -2.0f * j * G[0] + i * G[1] + r * G[2] + i * G[3] + k * G[5] - r * G[6] + k * G[7] - 2.0f * j * G[8],
            -2.0f * k * G[0] - r * G[1] + i * G[2] + r * G[3] - 2.0f * k * G[4] + j * G[5] + i * G[6] + j * G[7]};
        for (int m = 0; m < 4; ++m) grad[m] = sc * d[m] - sc * sc * q[m] * ga;
    } else {
        float w[3] = {a.rot[0], a.rot[1], a.rot[2]}, R[9], K[9], th;
        rodrigues(w, R, K, th);
        const float s = sinf(th), c = cosf(th);
        float gk = 0.f, gkk = 0.f;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                float kk = 0.f;
                for (int k = 0; k < 3; ++k) kk += K[i * 3 + k] * K[k * 3 + j];
                gk += G[i * 3 + j] * K[i * 3 + j];
                gkk += G[i * 3 + j] * kk;
            }
        const float g_th = c * gk + s * gkk;                         // through sin / (1 - cos)
        float dK[9];                                                 // dL/dK = s G + (1 - c)(G K^T + K^T G)
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                float t1 = 0.f, t2 = 0.f;
                for (int k = 0; k < 3; ++k) { t1 += G[i * 3 + k] * K[j * 3 + k]; t2 += K[k * 3 + i] * G[k * 3 + j]; }
                dK[i * 3 + j] = s * G[i * 3 + j] + (1.0f - c) * (t1 + t2);
            }
        const float go[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};       // d/d(omega hat)
        const float o[3] = {w[0] / th, w[1] / th, w[2] / th};
        const float og = o[0] * go[0] + o[1] * go[1] + o[2] * go[2];
        for (int j = 0; j < 3; ++j) grad[j] = (go[j] - o[j] * og) / th + g_th * o[j];
    }
    for (int j = 0; j < 3; ++j) grad[nr + j] = red[9 + j][0];
    // ---- torch.optim.Adam (amsgrad off, no weight decay), two groups
    const int t = *a.step + 1;
    *a.step = t;
    const double bc1 = 1.0 - pow(a.beta1, (double)t), bc2 = 1.0 - pow(a.beta2, (double)t);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float omb1 = (float)(1.0 - a.beta1), b2 = (float)a.beta2, omb2 = (float)(1.0 - a.beta2), eps = (float)a.eps;
    for (int k = 0; k < nr + 3; ++k) {
        const float step_size = (float)((k < nr ? a.lr_rot : a.lr_trans) / bc1);
        float* p = k < nr ? a.rot + k : a.trans + (k - nr);
        const float g = grad[k];
        float m = a.m[k], v = a.v[k];
        m = m + (g - m) * omb1;
        v = b2 * v + omb2 * g * g;
        a.m[k] = m; a.v[k] = v;
        *p = *p - step_size * (m / (sqrtf(v) / bc2_sqrt + eps));
    }
}

int mne_launch_pose(const PoseArgs& a, int what, hipStream_t st) {
    const int blocks = (a.n + 255) / 256;
    if (what == 0) MNE_LAUNCH(pose_rays_kernel, blocks, 256, 0, st, a);
    else if (what == 1) MNE_LAUNCH(pose_loss_kernel, blocks, 256, 0, st, a);
    else MNE_LAUNCH(pose_update_kernel, 1, 256, 0, st, a);
    return 0;
}
