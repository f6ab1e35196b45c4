// r03_lds_atomic_bench.hip -- stand-alone microbenchmark (not part of the library): how fast does one CU add
// 32-channel gradient rows into a 16x16x32 fp32 tile in LDS with ds_add_f32, in the access pattern a per-tile plane
// update would use (half-wave = one list entry, lane = channel, 4 bilinear corners per entry)?
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics profiles/r03_lds_atomic_bench.hip -o profiles/_bin/lds_atomic_bench
// Decides whether tile_adam_kernel's counting sort + single-writer accumulate (11-46 us on heavy tiles,
// profiles/r01_tile_adam_phases.txt) can be replaced by plain LDS float atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// mode 0: ds_add_f32 (atomicAdd, result unused)   1: plain read-add-write (racy; upper bound of the LDS data path)
// mode 2: ds_add_u32 on fixed-point values   3: ds_add_u64
// spread: cells are drawn from [0, spread) -- 256 = whole tile, 16 = one cell row (a wall seen edge-on), 1 = one cell
template <int MODE>
__global__ __launch_bounds__(512) void accum_kernel(const unsigned* cells, const float* rows, float* out, int n_entries, int reps) {
    __shared__ float g[256 * 32];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, hw = tid >> 5;       // 16 half-waves
    for (int i = tid; i < 256 * 32; i += 512) g[i] = 0.f;
    __syncthreads();
    const unsigned* my_cells = cells + (size_t)blockIdx.x * n_entries;
    for (int r = 0; r < reps; ++r) {
        for (int e0 = 0; e0 < n_entries; e0 += 16 * 8) {
            float v[8];
            unsigned cell[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = e0 + j * 16 + hw;
                cell[j] = my_cells[e];
                v[j] = rows[(size_t)(e & 1023) * 32 + c];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int x = cell[j] & 15, y = (cell[j] >> 4) & 15;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int xx = (x + (q & 1)) & 15, yy = (y + (q >> 1)) & 15;
                    float* p = g + (yy * 16 + xx) * 32 + c;
                    const float w = 0.25f * v[j];
                    if (MODE == 0) atomicAdd(p, w);
                    else if (MODE == 1) *p += w;
                    else if (MODE == 2) atomicAdd((int*)p, (int)(w * 1048576.0f));                      // ds_add_u32, fixed point
                    else atomicAdd((unsigned long long*)g + ((yy * 16 + xx) * 32 + c) / 2, (unsigned long long)(long long)(w * 1048576.0f));   // ds_add_u64 (half the cells)
                }
            }
        }
    }
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < 256 * 32; i += 512) s += g[i];
    if (s == 12345.678f) out[blockIdx.x] = s;
}

int main() {
    const int n_entries = 4096, blocks = 512, reps = 8;
    std::vector<float> rows(1024 * 32, 1.0f);
    float *d_rows, *d_out;
    unsigned* d_cells;
    CHECK(hipMalloc(&d_rows, rows.size() * 4));
    CHECK(hipMalloc(&d_out, blocks * 4));
    CHECK(hipMalloc(&d_cells, (size_t)blocks * n_entries * 4));
    CHECK(hipMemcpy(d_rows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int spread : {256, 16, 1}) {
        std::vector<unsigned> cells((size_t)blocks * n_entries);
        unsigned s = 12345u;
        for (auto& c : cells) { s = s * 1664525u + 1013904223u; c = (s >> 8) % spread; }
        CHECK(hipMemcpy(d_cells, cells.data(), cells.size() * 4, hipMemcpyHostToDevice));
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9f;
            for (int t = 0; t < 3; ++t) {
                CHECK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(accum_kernel<0>, dim3(blocks), dim3(512), 0, 0, d_cells, d_rows, d_out, n_entries, reps);
                else if (mode == 1) hipLaunchKernelGGL(accum_kernel<1>, dim3(blocks), dim3(512), 0, 0, d_cells, d_rows, d_out, n_entries, reps);
                else if (mode == 2) hipLaunchKernelGGL(accum_kernel<2>, dim3(blocks), dim3(512), 0, 0, d_cells, d_rows, d_out, n_entries, reps);
                else hipLaunchKernelGGL(accum_kernel<3>, dim3(blocks), dim3(512), 0, 0, d_cells, d_rows, d_out, n_entries, reps);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
            }
            const double entries = (double)blocks * n_entries * reps;
            printf("cells spread %3d  %s: %.3f ms  -> %.2f G entry-corners/s chip-wide, %.2f us per 512-entry pass per workgroup (512 workgroups resident), %.1f G lane-ops/s per CU\n",
                   spread, mode == 0 ? "ds_add_f32      " : mode == 1 ? "read-add-write  " : mode == 2 ? "ds_add_u32 fixed" : "ds_add_u64 fixed", best, entries * 4 / best / 1e6,
                   best * 1e3 / ((double)n_entries * reps / 512.0), entries * 4 * 32 / best / 1e6 / 256.0);
        }
    }
    return 0;
}
