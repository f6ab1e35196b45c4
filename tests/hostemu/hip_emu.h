// hip_emu.h -- TEST-ONLY host emulation of the small HIP subset the mneslam kernels use.
//
// Purpose: the build container has no GPU, and GPU time is scarce.  Compiling the *same* kernel
// sources with g++ -DMNE_HOST_EMU against this header lets tests/ run them on CPU tensors (slowly):
// one workgroup at a time, one OS thread per wave with its work-items as fibers (or, for sanitizer builds,
// -DMNE_EMU_OS_THREADS: one OS thread per work-item), wave64 collectives (__shfl*, __ballot, MFMA) and
// __syncthreads as barriers.  It is never part of the shipped library:
// mneslam_amd/ loads only the hipcc-built libmneslam_hip.so and fails loudly without it.
//
// Emulated semantics follow /opt/skills/guides (wave = 64 lanes; MFMA f32 32x32x2 fragment layout:
// A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D col=l&31,row=(reg&3)+8*(reg>>2)+4*(l>>5)).
#pragma once
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <sys/mman.h>
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define __shared__ static

struct emu_dim3 { unsigned x = 1, y = 1, z = 1; };
inline thread_local emu_dim3 threadIdx, blockIdx;
inline emu_dim3 blockDim, gridDim;

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
inline float2 make_float2(float a, float b) { return float2{a, b}; }

typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "emu"; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return 0; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { std::memcpy(d, s, n); return 0; }
#define hipMemcpyHostToDevice 1
#define hipMemcpyDeviceToDevice 3

namespace hipemu {
constexpr int WAVE = 64;
struct BlockCtx {
    unsigned nthreads;
    std::unique_ptr<std::barrier<>> block_bar;
    std::unique_ptr<std::barrier<>> end_bar;
    std::vector<std::unique_ptr<std::barrier<>>> wave_bar;      // (OS-thread model only)
    std::vector<uint64_t> xchg;      // one 8-byte slot per work-item
    std::vector<uint64_t> xchg2;
    std::vector<unsigned char> dyn_lds;
};
inline BlockCtx* g_ctx = nullptr;

inline int lane() { return threadIdx.x & 63; }
inline int wave() { return threadIdx.x >> 6; }
inline int wave_lanes() {           // live lanes in this (possibly partial) wave
    int base = wave() * WAVE;
    int n = (int)g_ctx->nthreads - base;
    return n > WAVE ? WAVE : n;
}

#ifdef MNE_EMU_OS_THREADS
// ---- model A (sanitizer builds): one OS thread per work-item, std::barrier per wave and per block
inline void wave_sync() { g_ctx->wave_bar[wave()]->arrive_and_wait(); }
inline void block_sync() { g_ctx->block_bar->arrive_and_wait(); }

template <class K, class... A>
void launch(K kern, unsigned grid, unsigned block, size_t lds, A... args) {
    BlockCtx ctx;
    ctx.nthreads = block;
    ctx.block_bar = std::make_unique<std::barrier<>>(block);
    ctx.end_bar = std::make_unique<std::barrier<>>(block);
    unsigned nw = (block + WAVE - 1) / WAVE;
    for (unsigned w = 0; w < nw; ++w) {
        unsigned n = std::min<unsigned>(WAVE, block - w * WAVE);
        ctx.wave_bar.emplace_back(std::make_unique<std::barrier<>>(n));
    }
    ctx.xchg.assign(block, 0);
    ctx.xchg2.assign(block, 0);
    ctx.dyn_lds.assign(lds + 64, 0);
    g_ctx = &ctx;
    blockDim.x = block;
    gridDim.x = grid;
    std::vector<std::thread> pool;
    pool.reserve(block);
    for (unsigned t = 0; t < block; ++t) {
        pool.emplace_back([&, t]() {
            for (unsigned b = 0; b < grid; ++b) {
                threadIdx.x = t;
                blockIdx.x = b;
                kern(args...);
                ctx.end_bar->arrive_and_wait();        // static __shared__ is reused by the next block
            }
        });
    }
    for (auto& th : pool) th.join();
    g_ctx = nullptr;
}
#else
// ---- model B (default): one OS thread per WAVE, its work-items are fibers on that thread (user-space context switch).
// A wave collective is a turn of the round robin instead of a futex barrier among 64 oversubscribed threads: the same
// kernels run 10-30x faster.  Lanes of a wave still execute independently between collectives, exactly as far as the
// kernels are allowed to assume (they mark every cross-lane dependency with MNE_WAVE_SYNC / a shuffle / __syncthreads).
constexpr size_t FIBER_STACK = 1u << 20;
struct WaveCtx {
    int n = 0, cur = 0, wave = 0, n_done = 0;
    void* sp[WAVE];
    bool done[WAVE];
    void* sched_sp = nullptr;
    int arrived = 0, arrived_b = 0;
    unsigned gen = 0, gen_b = 0;
    char* stacks = nullptr;
    std::function<void()> body;
};
inline thread_local WaveCtx* t_wave = nullptr;

// save callee-saved registers + stack pointer of the running fiber, continue the other one (System V x86-64)
__attribute__((naked, noinline)) static void emu_switch(void** /*save_sp: rdi*/, void* /*load_sp: rsi*/) {
    __asm__ volatile(
        "pushq %rbp\n pushq %rbx\n pushq %r12\n pushq %r13\n pushq %r14\n pushq %r15\n"
        "movq %rsp, (%rdi)\n movq %rsi, %rsp\n"
        "popq %r15\n popq %r14\n popq %r13\n popq %r12\n popq %rbx\n popq %rbp\n ret\n");
}
inline void fiber_resume_from(WaveCtx& w, int from, int to) {
    w.cur = to;
    threadIdx.x = (unsigned)(w.wave * WAVE + to);
    emu_switch(&w.sp[from], w.sp[to]);
}
inline void fiber_yield() {
    WaveCtx& w = *t_wave;
    int nxt = w.cur;
    for (int k = 1; k <= w.n; ++k) { nxt = (w.cur + k) % w.n; if (!w.done[nxt]) break; }
    if (nxt == w.cur) {       // every other lane of the wave has left the kernel while this one waits in a collective
        std::fprintf(stderr, "hip_emu: lane %d of wave %d waits in a collective its wave has abandoned\n", w.cur, w.wave);
        std::abort();
    }
    fiber_resume_from(w, w.cur, nxt);
}
inline void wave_sync() {
    WaveCtx& w = *t_wave;
    if (w.n == 1) return;
    const unsigned g = w.gen;
    if (++w.arrived == w.n) { w.arrived = 0; ++w.gen; return; }
    while (*(volatile unsigned*)&w.gen == g) fiber_yield();
}
inline void block_sync() {
    WaveCtx& w = *t_wave;
    const unsigned g = w.gen_b;
    if (++w.arrived_b == w.n) {            // the wave's last lane meets the other waves, then releases its own
        w.arrived_b = 0;
        g_ctx->block_bar->arrive_and_wait();
        ++w.gen_b;
        return;
    }
    while (*(volatile unsigned*)&w.gen_b == g) fiber_yield();
}
static void fiber_entry() {
    WaveCtx& w = *t_wave;
    w.body();
    WaveCtx& v = *t_wave;
    v.done[v.cur] = true;
    void* dead;
    if (++v.n_done == v.n) emu_switch(&dead, v.sched_sp);          // back to the wave's OS thread: block finished
    int nxt = v.cur;
    for (int k = 1; k <= v.n; ++k) { nxt = (v.cur + k) % v.n; if (!v.done[nxt]) break; }
    v.cur = nxt;
    threadIdx.x = (unsigned)(v.wave * WAVE + nxt);
    emu_switch(&dead, v.sp[nxt]);
    __builtin_unreachable();
}
inline void run_wave(WaveCtx& w, unsigned grid, BlockCtx& ctx) {
    t_wave = &w;
    w.stacks = (char*)mmap(nullptr, FIBER_STACK * w.n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (w.stacks == (char*)MAP_FAILED) { std::perror("hip_emu: mmap"); std::abort(); }
    for (unsigned b = 0; b < grid; ++b) {
        blockIdx.x = b;
        w.n_done = 0; w.arrived = w.arrived_b = 0;
        for (int i = 0; i < w.n; ++i) {
            w.done[i] = false;
            uintptr_t top = ((uintptr_t)(w.stacks + FIBER_STACK * (size_t)(i + 1))) & ~(uintptr_t)15;
            void** s = (void**)top;
            *--s = nullptr;                      // keeps the entry frame 16-byte aligned as after a call
            *--s = (void*)&fiber_entry;          // `ret` target of the first switch
            for (int r = 0; r < 6; ++r) *--s = nullptr;
            w.sp[i] = (void*)s;
        }
        w.cur = 0;
        threadIdx.x = (unsigned)(w.wave * WAVE);
        emu_switch(&w.sched_sp, w.sp[0]);
        ctx.end_bar->arrive_and_wait();          // static __shared__ is reused by the next block
    }
    munmap(w.stacks, FIBER_STACK * w.n);
    t_wave = nullptr;
}

template <class K, class... A>
void launch(K kern, unsigned grid, unsigned block, size_t lds, A... args) {
    BlockCtx ctx;
    ctx.nthreads = block;
    const unsigned nw = (block + WAVE - 1) / WAVE;
    ctx.block_bar = std::make_unique<std::barrier<>>(nw);
    ctx.end_bar = std::make_unique<std::barrier<>>(nw);
    ctx.xchg.assign(block, 0);
    ctx.xchg2.assign(block, 0);
    ctx.dyn_lds.assign(lds + 64, 0);
    g_ctx = &ctx;
    blockDim.x = block;
    gridDim.x = grid;
    std::vector<std::thread> pool;
    pool.reserve(nw);
    for (unsigned wv = 0; wv < nw; ++wv) {
        pool.emplace_back([&, wv]() {
            WaveCtx w;
            w.wave = (int)wv;
            w.n = (int)std::min<unsigned>(WAVE, block - wv * WAVE);
            w.body = [&]() { kern(args...); };
            run_wave(w, grid, ctx);
        });
    }
    for (auto& th : pool) th.join();
    g_ctx = nullptr;
}
#endif

template <class T> inline uint64_t to_bits(T v) { uint64_t b = 0; std::memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; std::memcpy(&v, &b, sizeof(T)); return v; }

template <class T> inline T shfl_idx(T v, int src) {
    static_assert(sizeof(T) <= 8, "shuffle payload");
    int base = wave() * WAVE;
    g_ctx->xchg[threadIdx.x] = to_bits(v);
    wave_sync();
    int s = src & 63;
    T r = (s < wave_lanes()) ? from_bits<T>(g_ctx->xchg[base + s]) : v;
    wave_sync();
    return r;
}

template <class K, class... A>
void launch2d(K kern, unsigned gx, unsigned gy, unsigned block, A... args) {
    // collective-free kernels only: run work-items sequentially on the calling thread
    blockDim.x = block; gridDim.x = gx; gridDim.y = gy;
    for (unsigned by = 0; by < gy; ++by)
        for (unsigned bx = 0; bx < gx; ++bx)
            for (unsigned t = 0; t < block; ++t) {
                threadIdx.x = t; blockIdx.x = bx; blockIdx.y = by;
                kern(args...);
            }
}
}  // namespace hipemu

#define hipLaunchOrEmu2D(kern, gx, gy, block, stream, ...) hipemu::launch2d(kern, (unsigned)(gx), (unsigned)(gy), (unsigned)(block), __VA_ARGS__)
#define MNE_LAUNCH(kern, grid, block, lds, stream, ...) hipemu::launch(kern, (unsigned)(grid), (unsigned)(block), (size_t)(lds), __VA_ARGS__)
#define MNE_DYN_LDS(name) unsigned char* name = (unsigned char*)(((uintptr_t)hipemu::g_ctx->dyn_lds.data() + 15) & ~(uintptr_t)15)
typedef const float* mne_cptr;
#define MNE_CPTR(p) ((const float*)(p))

#define MNE_WAVE_SYNC() hipemu::wave_sync()
#define MNE_SCHED_BARRIER() do { } while (0)
inline int mne_xcc_id() { return (int)(blockIdx.x & 7u); }          // workgroups go round the XCDs
#define MNE_LDS_MAX (160 * 1024)
#define MNE_DRAIN_STORES() do { } while (0)
#define MNE_FENCE_RELEASE_AGENT() std::atomic_thread_fence(std::memory_order_seq_cst)
#define MNE_FENCE_ACQUIRE_AGENT() std::atomic_thread_fence(std::memory_order_seq_cst)
#define MNE_SET_MAX_LDS(kern, bytes) ((void)0)
inline void __syncthreads() { hipemu::block_sync(); }
inline void __threadfence() { std::atomic_thread_fence(std::memory_order_seq_cst); }
template <class T> inline T __shfl(T v, int src, int = 64) { return hipemu::shfl_idx(v, src); }
template <class T> inline T mne_bcast8(T v, int k) { return __shfl(v, (int)((threadIdx.x & 63u) & ~7u) | k); }
template <class T> inline T __shfl_xor(T v, int m, int = 64) { return hipemu::shfl_idx(v, hipemu::lane() ^ m); }
template <class T> inline T __shfl_down(T v, unsigned d, int = 64) {
    int s = hipemu::lane() + (int)d;
    return hipemu::shfl_idx(v, s < 64 ? s : hipemu::lane());
}
template <class T> inline T __shfl_up(T v, unsigned d, int = 64) {
    int s = hipemu::lane() - (int)d;
    return hipemu::shfl_idx(v, s >= 0 ? s : hipemu::lane());
}
inline unsigned long long __ballot(int pred) {
    using namespace hipemu;
    int base = wave() * WAVE;
    g_ctx->xchg2[threadIdx.x] = pred ? 1 : 0;
    wave_sync();
    unsigned long long m = 0;
    for (int l = 0; l < wave_lanes(); ++l) m |= (unsigned long long)(g_ctx->xchg2[base + l] & 1) << l;
    wave_sync();
    return m;
}
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
inline int __clzll(unsigned long long v) { return v ? __builtin_clzll(v) : 64; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }

inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
inline float __uint_as_float(unsigned u) { float f; std::memcpy(&f, &u, 4); return f; }
inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }
inline long long __double_as_longlong(double d) { long long u; std::memcpy(&u, &d, 8); return u; }
inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

template <class T> inline T atomic_add_impl(T* addr, T val) {
    static_assert(sizeof(T) == 4, "4-byte atomics only");
    auto* a = reinterpret_cast<std::atomic<uint32_t>*>(addr);
    uint32_t old = a->load(std::memory_order_relaxed);
    for (;;) {
        T cur; std::memcpy(&cur, &old, 4);
        T nv = cur + val;
        uint32_t nb; std::memcpy(&nb, &nv, 4);
        if (a->compare_exchange_weak(old, nb, std::memory_order_relaxed)) return cur;
    }
}
inline float atomicAdd(float* a, float v) { return atomic_add_impl(a, v); }
inline int atomicAdd(int* a, int v) { return atomic_add_impl(a, v); }
inline unsigned atomicAdd(unsigned* a, unsigned v) { return atomic_add_impl(a, v); }
inline unsigned long long atomicAdd(unsigned long long* a, unsigned long long v) {
    return reinterpret_cast<std::atomic<unsigned long long>*>(a)->fetch_add(v, std::memory_order_relaxed);
}
inline float unsafeAtomicAdd(float* a, float v) { return atomic_add_impl(a, v); }

// MFMA f32 32x32x2: every lane of a full wave must call it together.
typedef float emu_f32x16 __attribute__((ext_vector_type(16)));
inline emu_f32x16 emu_mfma_f32_32x32x2f32(float a, float b, emu_f32x16 c) {
    using namespace hipemu;
    int base = wave() * WAVE, l = lane();
    g_ctx->xchg[threadIdx.x] = to_bits(a);
    g_ctx->xchg2[threadIdx.x] = to_bits(b);
    wave_sync();
    int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av = from_bits<float>(g_ctx->xchg[base + i + 32 * k]);
            float bv = from_bits<float>(g_ctx->xchg2[base + j + 32 * k]);
            acc = std::fmaf(av, bv, acc);
        }
        c[r] = acc;
    }
    wave_sync();
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) emu_mfma_f32_32x32x2f32(a, b, c)
