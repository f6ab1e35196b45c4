// r06_queue_handoff.hip -- what a dependency between two kernels costs on MI355X / ROCm 7.2, by the way it is expressed.
//   hipcc --offload-arch=gfx950 -O2 profiles/r06_queue_handoff.hip -o gpurun_out/r06_queue_handoff && gpurun_out/r06_queue_handoff
// Cases (kernel = 64 workgroups spinning for ~40 us on the realtime counter; N repetitions, wall clock per pair):
//   same      A then B in ONE stream (barrier bit between them)
//   anyorder  A then B in one stream, B launched with hipExtAnyOrderLaunch (no barrier bit: B may run beside A)
//   event     A on stream 1, event record, stream 2 waits for the event, B on stream 2
//   value     A on stream 1, hipStreamWriteValue32, stream 2 hipStreamWaitValue32 on signal memory, B on stream 2
//   pingpong  the mapping iteration's shape: A(s1) -> B(s2) -> A(s1) -> ... joined by events both ways
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spin_kernel(long long ticks, int* sink) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1);
}

static double now_us() {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    int rate_khz = 0;
    CHECK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const double spin_us = 40.0;
    const long long ticks = (long long)(spin_us * rate_khz / 1000.0);
    int can_wait = 0;
    (void)hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("wall clock %d kHz, spin %lld ticks = %.0f us, stream wait value supported: %d\n", rate_khz, ticks, spin_us, can_wait);
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    int* sink;
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(sink, 0, 4));
    const int N = 200;
    hipEvent_t ev[2 * N];
    for (int i = 0; i < 2 * N; ++i) CHECK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
    auto launch = [&](hipStream_t s, unsigned flags) {
        if (flags) hipExtLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s, nullptr, nullptr, flags, ticks, sink);
        else hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s, ticks, sink);
    };
    // warm-up
    for (int i = 0; i < 10; ++i) { launch(s1, 0); launch(s2, 0); }
    CHECK(hipDeviceSynchronize());
    auto report = [&](const char* name, double t_us, int pairs) {
        printf("%-9s %8.1f us per pair (two kernels of %.0f us): hand-off overhead %6.1f us vs serial, %6.1f us vs concurrent\n", name,
               t_us / pairs, spin_us, t_us / pairs - 2 * spin_us, t_us / pairs - spin_us);
    };
    {
        double t0 = now_us();
        for (int i = 0; i < N; ++i) { launch(s1, 0); launch(s1, 0); }
        CHECK(hipStreamSynchronize(s1));
        report("same", now_us() - t0, N);
    }
    {
        double t0 = now_us();
        for (int i = 0; i < N; ++i) { launch(s1, 0); launch(s1, hipExtAnyOrderLaunch); }
        CHECK(hipStreamSynchronize(s1));
        report("anyorder", now_us() - t0, N);
    }
    {
        double t0 = now_us();
        for (int i = 0; i < N; ++i) {
            launch(s1, 0);
            CHECK(hipEventRecord(ev[2 * i], s1));
            CHECK(hipStreamWaitEvent(s2, ev[2 * i], 0));
            launch(s2, 0);
            CHECK(hipEventRecord(ev[2 * i + 1], s2));
            CHECK(hipStreamWaitEvent(s1, ev[2 * i + 1], 0));
        }
        CHECK(hipStreamSynchronize(s1));
        CHECK(hipStreamSynchronize(s2));
        report("pingpong", now_us() - t0, N);
    }
    if (can_wait) {
        unsigned* sig = nullptr;
        if (hipExtMallocWithFlags((void**)&sig, 8, hipMallocSignalMemory) == hipSuccess) {
            CHECK(hipMemset(sig, 0, 8));
            CHECK(hipDeviceSynchronize());
            double t0 = now_us();
            bool ok = true;
            for (int i = 0; i < N && ok; ++i) {
                launch(s1, 0);
                ok = ok && hipStreamWriteValue32(s1, sig, 2 * i + 1, 0) == hipSuccess;
                ok = ok && hipStreamWaitValue32(s2, sig, 2 * i + 1, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
                launch(s2, 0);
                ok = ok && hipStreamWriteValue32(s2, sig, 2 * i + 2, 0) == hipSuccess;
                ok = ok && hipStreamWaitValue32(s1, sig, 2 * i + 2, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess;
            }
            CHECK(hipStreamSynchronize(s1));
            CHECK(hipStreamSynchronize(s2));
            if (ok) report("value", now_us() - t0, N);
            else printf("value     stream write / wait value calls failed\n");
        } else printf("value     hipMallocSignalMemory allocation failed\n");
    }
    // a wait for an event that fired long ago: A(s1), [s2: short kernel + record, done while A spins], s1 waits for it, B(s1)
    {
        long long short_ticks = ticks / 20;
        double t0 = now_us();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s2, short_ticks, sink);
            CHECK(hipEventRecord(ev[i], s2));
            launch(s1, 0);
            CHECK(hipStreamWaitEvent(s1, ev[i], 0));
            launch(s1, 0);
        }
        CHECK(hipStreamSynchronize(s1));
        CHECK(hipStreamSynchronize(s2));
        report("satisfied", now_us() - t0, N);
    }
    // the fused step's shape: chain C on s1 (A -> B), helper H on s2 that starts after A (event) and ends before B would (shorter),
    // B waits for H: A(40) -> [H(20) on s2] ; B after A and H.  Critical path if hops were free: A + B.
    {
        long long half = ticks / 2;
        double t0 = now_us();
        for (int i = 0; i < N; ++i) {
            launch(s1, 0);                                        // A
            CHECK(hipEventRecord(ev[2 * i], s1));
            CHECK(hipStreamWaitEvent(s2, ev[2 * i], 0));
            hipLaunchKernelGGL(spin_kernel, dim3(64), dim3(64), 0, s2, half, sink);      // H (20 us) beside A2
            CHECK(hipEventRecord(ev[2 * i + 1], s2));
            launch(s1, 0);                                        // A2 (40 us), same queue as A
            CHECK(hipStreamWaitEvent(s1, ev[2 * i + 1], 0));      // H finished ~20 us ago
            launch(s1, 0);                                        // B
        }
        CHECK(hipStreamSynchronize(s1));
        CHECK(hipStreamSynchronize(s2));
        const double t = (now_us() - t0) / N;
        printf("sidecar   %8.1f us per round (A, A2, B of %.0f us in one queue = %.0f; helper of %.0f us beside A2, joined before B)\n", t, spin_us,
               3 * spin_us, spin_us / 2);
    }
    // three kernels, the middle one any-order, a barrier kernel behind: A[B] B[any] C[B] -- C must wait for both
    {
        double t0 = now_us();
        for (int i = 0; i < N; ++i) { launch(s1, 0); launch(s1, hipExtAnyOrderLaunch); launch(s1, 0); }
        CHECK(hipStreamSynchronize(s1));
        const double t = (now_us() - t0) / N;
        printf("A,B*,C    %8.1f us per triple (serial %.0f, with B beside A %.0f)\n", t, 3 * spin_us, 2 * spin_us);
    }
    int h = 0;
    CHECK(hipMemcpy(&h, sink, 4, hipMemcpyDeviceToHost));
    printf("kernels run: %d\n", h);
    return 0;
}
