// Probe (not part of the product): latency of a cross-queue dependency through
//   (a) a one-wave polling kernel (k_gate, what dqn.hip uses), (b) hipStreamWaitValue32 on hipMallocSignalMemory,
//   (c) hipEventRecord + hipStreamWaitEvent.
// Queue A: work kernel, then a publishing kernel; queue B: wait, then a work kernel.  Reported: mean time from the end of
// A's work kernel to the start of B's work kernel, and B's total time per iteration.
// Build: hipcc --offload-arch=gfx950 -O3 tools/probes/waitvalue_probe.hip -o tools/probes/waitvalue_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void k_work(unsigned long long* ts, int spin)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) ts[0] = wall_clock64();
    float x = threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    if (x == 123.f) ts[2] = 1;
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == 0) ts[1] = wall_clock64();
}
__global__ void k_pub(unsigned* sig, unsigned v) { if (threadIdx.x == 0) __hip_atomic_store(sig, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void k_gate(unsigned* sig, unsigned v)
{
    if (threadIdx.x != 0) return;
    while ((int)(__hip_atomic_load(sig, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - v) < 0) __builtin_amdgcn_s_sleep(4);
}

int main()
{
    const int N = 300, SPIN = 4000;
    hipStream_t A, B;
    CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    unsigned* sig_mem; unsigned* sig_sig;
    CK(hipMalloc((void**)&sig_mem, 64)); CK(hipMemset(sig_mem, 0, 64));
    hipError_t e = hipExtMallocWithFlags((void**)&sig_sig, 8, hipMallocSignalMemory);
    printf("hipMallocSignalMemory: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) CK(hipMemset(sig_sig, 0, 8));
    int can = 0; (void)hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("CanUseStreamWaitValue: %d\n", can);
    unsigned long long *tsA, *tsB;
    CK(hipMalloc((void**)&tsA, N * 32)); CK(hipMalloc((void**)&tsB, N * 32));
    hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming | hipEventDisableSystemFence));
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 1 && (e != hipSuccess || !can)) { printf("mode 1 skipped\n"); continue; }
        unsigned* sig = mode == 1 ? sig_sig : sig_mem;
        unsigned base = 1000u * (mode + 1);
        CK(hipDeviceSynchronize());
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, A, tsA + i * 4, SPIN);
            if (mode == 2) { CK(hipEventRecord(ev, A)); CK(hipStreamWaitEvent(B, ev, 0)); }
            else hipLaunchKernelGGL(k_pub, dim3(1), dim3(64), 0, A, sig, base + i + 1);
            if (mode == 0) hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, B, sig, base + i + 1);
            if (mode == 1) CK(hipStreamWaitValue32(B, sig, base + i + 1, hipStreamWaitValueGte, 0xFFFFFFFFu));
            hipLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, B, tsB + i * 4, SPIN);
            // keep the two queues in lock step so that the measured latency is the dependency's, not queueing
            CK(hipStreamSynchronize(B));
        }
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> a(N * 4), b(N * 4);
        CK(hipMemcpy(a.data(), tsA, N * 32, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), tsB, N * 32, hipMemcpyDeviceToHost));
        double lat = 0, work = 0; int n = 0;
        for (int i = 20; i < N; ++i) { lat += (double)(long long)(b[i * 4] - a[i * 4 + 1]); work += (double)(a[i * 4 + 1] - a[i * 4]); ++n; }
        printf("mode %d (%s): A.work end -> B.work start %.2f us  (work kernel %.2f us)\n", mode,
               mode == 0 ? "polling gate kernel" : mode == 1 ? "hipStreamWaitValue32 on signal memory" : "event record + wait", lat / n / 100.0, work / n / 100.0);
    }
    return 0;
}
