// mne_platform.h -- the one place that decides between the real HIP toolchain (the product:
// hipcc --offload-arch=gfx950) and the test-only host emulator (tests/hostemu/hip_emu.h, used by
// the CPU test-suite to run these same kernel sources without a GPU).
#pragma once
#ifdef MNE_HOST_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#define MNE_LAUNCH(kern, grid, block, lds, stream, ...) \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, stream, __VA_ARGS__)
#define hipLaunchOrEmu2D(kern, gx, gy, block, stream, ...) hipLaunchKernelGGL(kern, dim3(gx, gy), dim3(block), 0, stream, __VA_ARGS__)
#define MNE_LDS_MAX (160 * 1024)      // LDS per CU on gfx950
// inter-workgroup hand-off inside one launch (cdna_hip_programming.md, Guideline 16): drain this wave's stores, agent-scope
// release / acquire by one lane
#define MNE_DRAIN_STORES() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define MNE_FENCE_RELEASE_AGENT() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define MNE_FENCE_ACQUIRE_AGENT() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define MNE_SET_MAX_LDS(kern, bytes) (void)hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))
#define MNE_DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
// Decoder weights are wave-uniform: reading them through the constant address space makes the
// compiler use scalar (SMEM) loads, so the tiny-MLP FMAs take their weight operand from SGPRs.
typedef const __attribute__((address_space(4))) float* mne_cptr;
#define MNE_CPTR(p) ((mne_cptr)(unsigned long long)(p))
// Broadcast of lane k (0..7, compile-time) inside every aligned group of 8 lanes: ds_swizzle in bit-mask mode,
// new_lane = (lane & 0x18) | k within each half-wave; goes through the LDS crossbar, touches no LDS memory.
#define MNE_SWZ8(k) (0x18 | ((k) << 5))
static __device__ __forceinline__ int mne_bcast8(int v, int k) {
    switch (k) {      // the pattern must be an immediate
        case 0: return __builtin_amdgcn_ds_swizzle(v, MNE_SWZ8(0));
        case 1: return __builtin_amdgcn_ds_swizzle(v, MNE_SWZ8(1));
        case 2: return __builtin_amdgcn_ds_swizzle(v, MNE_SWZ8(2));
        case 3: return __builtin_amdgcn_ds_swizzle(v, MNE_SWZ8(3));
        case 4: return __builtin_amdgcn_ds_swizzle(v, MNE_SWZ8(4));
        case 5: return __builtin_amdgcn_ds_swizzle(v, MNE_SWZ8(5));
        case 6: return __builtin_amdgcn_ds_swizzle(v, MNE_SWZ8(6));
        default: return __builtin_amdgcn_ds_swizzle(v, MNE_SWZ8(7));
    }
}
static __device__ __forceinline__ float mne_bcast8(float v, int k) { return __int_as_float(mne_bcast8(__float_as_int(v), k)); }
// The XCD (accelerator complex die, 0..7 on MI355X) this wave runs on: hardware register XCC_ID.
static __device__ __forceinline__ int mne_xcc_id() {
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7;
}
// LDS hand-off between the lanes of ONE wave (each wave owns a private LDS region): DS operations of
// a wave complete in issue order, so draining lgkmcnt and pinning the compiler's order is enough --
// no s_barrier, the four waves of a workgroup never wait for each other.
// pin the instruction scheduler at this point (nothing moves across it)
#define MNE_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#define MNE_WAVE_SYNC()                                         \
    do {                                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)
#endif
