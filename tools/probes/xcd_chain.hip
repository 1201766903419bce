// Stage-0 probe for a persistent SAC step: a chain of L dense layers [1024 x 256] x [256 x 256] + bias + ReLU, either as L launches
// (the product's k_dense_small form) or as ONE launch of 256 workgroups in eight XCD-local groups (rows of XCD x: [128 x, 128 x + 128)),
// the groups ordered layer to layer by an XCD-local barrier: plain stores stay in the XCD's L2, the counter is an L2 atomic, the
// readers invalidate their L1 (modes 0, 1) or read the activations with nt loads, which bypass L1 (modes 2: L2 counter, 3: agent counter, no sleep).  Question: how many microseconds per layer in the chain?
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iborder_amd/csrc tools/probes/xcd_chain.hip -o tools/probes/xcd_chain.bin
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dense.hpp"

using namespace bdr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static float* dev_rand(size_t n, float lo, float hi, unsigned seed)
{
    std::vector<float> h(n);
    unsigned s = seed * 2654435761u + 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = lo + (hi - lo) * ((s >> 8) * (1.0f / 16777216.0f)); }
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
    return d;
}

constexpr int Bn = 1024, H = 256;

struct ChainArgs {
    const float* w; const float* b;   // [L][H][H], [L][H]
    float* x0; float* x1;             // ping-pong activations [Bn][H]
    unsigned* cnt;                    // [8][32] one counter line per XCD
    unsigned* bad;                    // [4]: xcc mismatches, time-outs
    int L, mode;                      // mode 0: L2-local counter (workgroup-scope atomic + sc1 poll); 1: agent-scope atomic + sc1 poll
    unsigned base;                    // counter value at launch (counters are monotonic across launches)
};

template <bool NT = false>
__device__ __forceinline__ void layer_tile(const float* xin, float* xout, const float* w, const float* bias, int m0, int n0, float (*red)[32][33])
{
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* arow = xin + (size_t)(m0 + (lane & 31)) * H;
    const int r = tid >> 3, c4 = (tid & 7) * 4;
    const f32x4 e0 = *reinterpret_cast<const f32x4*>(bias + n0 + c4);
    dense_small_tile<false>([&](int k) { if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(arow + k)); else return *reinterpret_cast<const f32x4*>(arow + k); }, w, H, n0, H, wave, lane, red);
    __syncthreads();
    f32x4 v;
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = dense_small_sum(red, r, c4 + q) + e0[q]; v[q] = v[q] > 0.f ? v[q] : 0.f; }
    *reinterpret_cast<f32x4*>(xout + (size_t)(m0 + r) * H + n0 + c4) = v;
}

__global__ __launch_bounds__(256) void k_layer(const float* xin, float* xout, const float* w, const float* bias)
{
    __shared__ float red[4][32][33];
    const int m0 = ((int)blockIdx.x / 8) * 32, n0 = ((int)blockIdx.x % 8) * 32;
    layer_tile(xin, xout, w, bias, m0, n0, red);
}

__global__ __launch_bounds__(256) void k_chain(ChainArgs a)
{
    __shared__ float red[4][32][33];
    __shared__ unsigned s_bail;
    const int x = (int)blockIdx.x & 7, t = (int)blockIdx.x >> 3;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    if (threadIdx.x == 0) { s_bail = 0; if ((int)xcc != x) atomicAdd(a.bad, 1u); }
    __syncthreads();
    const int m0 = x * 128 + (t >> 3) * 32, n0 = (t & 7) * 32;
    unsigned* cnt = a.cnt + x * 32;
    const float* xin = a.x0; float* xout = a.x1;
    for (int l = 0; l < a.L; ++l) {
        if (a.mode >= 2) layer_tile<true>(xin, xout, a.w + (size_t)l * H * H, a.b + (size_t)l * H, m0, n0, red);
        else layer_tile(xin, xout, a.w + (size_t)l * H * H, a.b + (size_t)l * H, m0, n0, red);
        if (l + 1 == a.L) break;
        // ---- XCD-local barrier ----
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned want = a.base + 32u * (unsigned)(l + 1);
            if (a.mode == 0 || a.mode == 2) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long t0 = wall_clock64();
            while ((int)(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) < 0) {
                if (wall_clock64() - t0 > 100000000ll / 50) { atomicAdd(a.bad + 1, 1u); s_bail = 1; break; }   // 20 ms at 100 MHz
                if (a.mode < 3) __builtin_amdgcn_s_sleep(1);
            }
            if (a.mode < 2) asm volatile("buffer_inv sc1" ::: "memory");
        }
        __syncthreads();
        if (s_bail) return;
        const float* tmp = xout; xout = const_cast<float*>(xin); xin = tmp;
    }
}

int main(int argc, char** argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 20;
    float* w = dev_rand((size_t)L * H * H, -0.108f, 0.108f, 1);
    float* b = dev_rand((size_t)L * H, -0.05f, 0.05f, 2);
    float* x_init = dev_rand((size_t)Bn * H, 0.f, 1.f, 3);
    float *x0, *x1; CK(hipMalloc(&x0, (size_t)Bn * H * 4)); CK(hipMalloc(&x1, (size_t)Bn * H * 4));
    unsigned *cnt, *bad; CK(hipMalloc(&cnt, 8 * 32 * 4)); CK(hipMalloc(&bad, 16)); CK(hipMemset(cnt, 0, 8 * 32 * 4)); CK(hipMemset(bad, 0, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ref((size_t)Bn * H), got((size_t)Bn * H);
    const int reps = 200;
    float ms;

    auto run_launches = [&]() {
        const float* xin = x0; float* xout = x1;
        for (int l = 0; l < L; ++l) {
            hipLaunchKernelGGL(k_layer, dim3(256), dim3(256), 0, 0, xin, xout, w + (size_t)l * H * H, b + (size_t)l * H);
            const float* t = xout; xout = const_cast<float*>(xin); xin = t;
        }
    };
    CK(hipMemcpy(x0, x_init, (size_t)Bn * H * 4, hipMemcpyDeviceToDevice));
    run_launches(); CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref.data(), (L & 1) ? x1 : x0, (size_t)Bn * H * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < 20; ++i) run_launches();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) run_launches();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("L=%d launches:            %8.2f us per chain, %6.2f us per layer\n", L, ms * 1000 / reps, ms * 1000 / reps / L);

    unsigned base = 0;
    for (int mode = 0; mode < 4; ++mode) {
        ChainArgs a{w, b, x0, x1, cnt, bad, L, mode, 0};
        auto launch = [&]() { a.base = base; hipLaunchKernelGGL(k_chain, dim3(256), dim3(256), 0, 0, a); base += 32u * (unsigned)(L - 1); };
        size_t wrong = 0;
        for (int trial = 0; trial < 5; ++trial) {   // correctness under repetition (L1-warm readers: same addresses every launch)
            CK(hipMemcpy(x0, x_init, (size_t)Bn * H * 4, hipMemcpyDeviceToDevice));
            launch(); CK(hipDeviceSynchronize());
            CK(hipMemcpy(got.data(), (L & 1) ? x1 : x0, (size_t)Bn * H * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < got.size(); ++i) wrong += got[i] != ref[i];
        }
        for (int i = 0; i < 20; ++i) launch();
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned hb[4]; CK(hipMemcpy(hb, bad, 16, hipMemcpyDeviceToHost));
        printf("L=%d persistent (mode %d): %8.2f us per chain, %6.2f us per layer   wrong words %zu, xcc mismatches %u, time-outs %u\n", L, mode,
               ms * 1000 / reps, ms * 1000 / reps / L, wrong, hb[0], hb[1]);
        CK(hipMemset(bad, 0, 16));
    }
    return 0;
}
