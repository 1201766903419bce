// Where does k_dense_chain2 (dense_chain.hpp) spend its time?  SAC shapes: x [1024][64] -> 256 -> 256, nz networks per launch.
//   (a) two k_dense_small launches per pass  (b) the chain kernel, a wave per tile  (c) a wave per k-slice; back-to-back launches (every figure carries
//   the ~2.4 us of a dependent launch), bit comparison of h0 / h1 against (a), shader-clock stamps of workgroup 0 / wave 0.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DC2_STAMPS -Iinclude -Iborder_amd/csrc tools/probes/chain2_probe.hip -o tools/probes/chain2_probe.bin
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dense.hpp"
#include "dense_chain.hpp"

namespace bdr { thread_local char g_err[512]; thread_local int g_err_deferred; }
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

int main(int argc, char** argv)
{
    const int M = argc > 1 ? atoi(argv[1]) : 1024;
    const int reps = 300;
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int units[2] = {256, 256};
    MlpLayout net = make_mlp(23, units, 2, 1, false);
    const DenseLayer &l0 = net.L[0], &l1 = net.L[1];
    printf("layers: %d x %d, %d x %d\n", l0.Kp, l0.Np, l1.Kp, l1.Np);
    float* params[4]; float* x[4]; float *h0a[4], *h1a[4], *h0b[4], *h1b[4];
    for (int z = 0; z < 4; ++z) {
        params[z] = dev_rand(net.total, -0.1f, 0.1f, 10 + z);
        {   // input rows as the agents pack them: 23 columns, zero padding up to Kp
            std::vector<float> hx((size_t)M * l0.Kp, 0.f);
            unsigned sd = (20 + z) * 2654435761u + 12345u;
            for (int m = 0; m < M; ++m) for (int k = 0; k < 23; ++k) { sd = sd * 1664525u + 1013904223u; hx[(size_t)m * l0.Kp + k] = -1.f + 2.f * ((sd >> 8) * (1.0f / 16777216.0f)); }
            CK(hipMalloc(&x[z], hx.size() * 4)); CK(hipMemcpy(x[z], hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
        }
        CK(hipMalloc(&h0a[z], (size_t)M * 256 * 4)); CK(hipMalloc(&h1a[z], (size_t)M * 256 * 4));
        CK(hipMalloc(&h0b[z], (size_t)M * 256 * 4)); CK(hipMalloc(&h1b[z], (size_t)M * 256 * 4));
    }
    long long* stamps; CK(hipMalloc(&stamps, 64 * 8));
    std::vector<float> ra((size_t)M * 256), rb((size_t)M * 256);
    for (int nz : {1, 2, 4}) {
        DenseSrc in[4]; for (int z = 0; z < 4; ++z) in[z] = DenseSrc{x[z], l0.Kp};
        auto two = [&]() {
            DenseSrc mid[4]; for (int z = 0; z < 4; ++z) mid[z] = DenseSrc{h0a[z], l0.Np};
            (void)dense_forward_z(st, l0, nz, params, in, h0a, M, true);
            (void)dense_forward_z(st, l1, nz, params, mid, h1a, M, true);
        };
        float ms;
        two(); CK(hipStreamSynchronize(st));
        for (int i = 0; i < 20; ++i) two();
        CK(hipEventRecord(e0, st)); for (int i = 0; i < reps; ++i) two(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        printf("nz=%d two launches:        %7.2f us per pass\n", nz, ms * 1000 / reps);
        for (int tpw : {4, 1}) {
            Chain2Args c{};
            for (int z = 0; z < nz; ++z) c.n[z] = Chain2Net{x[z], l0.Kp, params[z] + l0.w, params[z] + l0.b, params[z] + l1.w, params[z] + l1.b, h0b[z], h1b[z]};
            c.M = M; c.n1 = l1.Np; c.relu0 = l0.relu; c.relu1 = l1.relu; c.stamps = stamps;
            const int rbk = (M + 31) / 32;
            auto launch = [&]() {
                if (tpw == 4) hipLaunchKernelGGL((k_dense_chain2<4, 2>), dim3(rbk * (l1.Np / 128), 1, nz), dim3(256), 0, st, c);
                else hipLaunchKernelGGL((k_dense_chain2<1, 2>), dim3(rbk * (l1.Np / 32), 1, nz), dim3(256), 0, st, c);
            };
            launch(); CK(hipStreamSynchronize(st));
            size_t wrong = 0;
            for (int z = 0; z < nz; ++z) {
                CK(hipMemcpy(ra.data(), h0a[z], ra.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), h0b[z], rb.size() * 4, hipMemcpyDeviceToHost));
                wrong += memcmp(ra.data(), rb.data(), ra.size() * 4) != 0;
                CK(hipMemcpy(ra.data(), h1a[z], ra.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(rb.data(), h1b[z], rb.size() * 4, hipMemcpyDeviceToHost));
                wrong += memcmp(ra.data(), rb.data(), ra.size() * 4) != 0;
            }
            for (int i = 0; i < 20; ++i) launch();
            CK(hipEventRecord(e0, st)); for (int i = 0; i < reps; ++i) launch(); CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
            long long hs[8]; CK(hipMemcpy(hs, stamps, 64, hipMemcpyDeviceToHost));
            printf("nz=%d chain, tiles/wg %d:   %7.2f us per pass   differing buffers %zu   stamps (cycles from start): loads issued %lld, layer 0 done %lld, barrier %lld, layer 1 done %lld (h0 stores issued before it)",
                   nz, tpw, ms * 1000 / reps, wrong, hs[1] - hs[0], hs[2] - hs[0], hs[3] - hs[0], hs[4] - hs[0]);
            if (tpw == 4) printf(", stores issued %lld", hs[6] - hs[0]);
            printf("\n");
        }
    }
    return 0;
}
