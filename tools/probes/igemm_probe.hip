// Tile-shape sweep for the conv2/conv3 forward and input-gradient implicit GEMMs (B=256 Atari shapes).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iborder_amd/csrc tools/probes/igemm_probe.hip -o tools/probes/igemm_probe.bin
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cnn_layers.hpp"
#include "igemm_dl.hpp"

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
static double checksum(const float* d, size_t n)
{
    std::vector<float> h(n); CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
    double s = 0; for (size_t i = 0; i < n; ++i) s += (double)h[i] * (double)((i % 97) + 1);
    return s;
}

// forward policy with k-major (per-tap transposed) weights [tap][cout][cin]: timing experiment only
template <class G, int WM_, int WN_>
struct FwdPT : FwdP<G, AFwd<G>, WM_, WN_, false> {
    static constexpr bool B_TR = true;
    __device__ static constexpr int KP(const FwdArgs&) { return G::CIN; }
};

template <class P, int STAGES>
static void run_dl(const char* name, dim3 grid, const typename P::Args& args, const float* out, size_t nout)
{
    hipStream_t st = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK((launch_igemm_dl<P, STAGES>(st, grid, args)));
    CK(hipDeviceSynchronize());
    const int IT = 50;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < IT; ++i) CK((launch_igemm_dl<P, STAGES>(st, grid, args)));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s grid=(%4u,%u,%u) thr=%4d  %7.2f us   sum=%.9e  [direct-to-LDS, %d stages]\n", name, grid.x, grid.y, grid.z, 64 * P::WM * P::WN,
           ms * 1000.0 / IT, checksum(out, nout), STAGES);
    fflush(stdout);
}

template <class P, int TEAMS>
static void run(const char* name, dim3 grid, const typename P::Args& args, const float* out, size_t nout)
{
    hipStream_t st = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) CK((launch_igemm<P, TEAMS>(st, grid, args)));
    CK(hipDeviceSynchronize());
    const int IT = 50;
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < IT; ++i) CK((launch_igemm<P, TEAMS>(st, grid, args)));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s grid=(%4u,%u,%u) thr=%4d  %7.2f us   sum=%.9e\n", name, grid.x, grid.y, grid.z, 64 * P::WM * P::WN * TEAMS,
           ms * 1000.0 / IT, checksum(out, nout));
    fflush(stdout);
}

int main()
{
    const int B = 256, NZ = 2;
    // forward conv2: x [B][20][20][32] -> [B*81][64]; conv3: [B][9][9][64] -> [B*49][64]
    float* x1 = dev_rand((size_t)B * 400 * 32, 0.f, 1.f, 1);
    float* w2 = dev_rand(512 * 64, -0.05f, 0.05f, 2);
    float* b2 = dev_rand(64, -0.1f, 0.1f, 3);
    float* h2[2]; for (int z = 0; z < 2; ++z) CK(hipMalloc(&h2[z], (size_t)B * 81 * 64 * 4));
    float* w3 = dev_rand(576 * 64, -0.05f, 0.05f, 4);
    float* h3[2]; for (int z = 0; z < 2; ++z) CK(hipMalloc(&h3[z], (size_t)B * 49 * 64 * 4));
    float* x2 = dev_rand((size_t)B * 81 * 64, 0.f, 1.f, 5);

    FwdArgs f2{}; for (int z = 0; z < NZ; ++z) { f2.x[z] = x1; f2.w[z] = w2; f2.bias[z] = b2; f2.out[z] = h2[z]; }
    f2.M = B * 81;
    FwdArgs f3{}; for (int z = 0; z < NZ; ++z) { f3.x[z] = x2; f3.w[z] = w3; f3.bias[z] = b2; f3.out[z] = h3[z]; }
    f3.M = B * 49;
    const size_t n2 = (size_t)B * 81 * 64, n3 = (size_t)B * 49 * 64;

#define FWD2(WM, WN, RP, TM, TN, T) { using P = FwdP<GeomC2, AFwd<GeomC2>, WM, WN, false, RP, TM, TN>; \
    CK(hipMemset(h2[1], 0, n2 * 4)); run<P, T>("fwd_c2 " #WM "x" #WN " rp" #RP " t" #TM #TN " teams" #T, dim3(m_tiles<P>(f2.M) * (64 / (WN * TN * 32)), 1, NZ), f2, h2[1], n2); }
#define FWD3(WM, WN, RP, TM, TN, T) { using P = FwdP<GeomC3, AFwd<GeomC3>, WM, WN, false, RP, TM, TN>; \
    CK(hipMemset(h3[1], 0, n3 * 4)); run<P, T>("fwd_c3 " #WM "x" #WN " rp" #RP " t" #TM #TN " teams" #T, dim3(m_tiles<P>(f3.M) * (64 / (WN * TN * 32)), 1, NZ), f3, h3[1], n3); }

#define FWD2D(WM, WN, RP, TM, TN, S) { using P = FwdP<GeomC2, AFwd<GeomC2>, WM, WN, false, RP, TM, TN>; \
    CK(hipMemset(h2[1], 0, n2 * 4)); run_dl<P, S>("fwd_c2 " #WM "x" #WN " rp" #RP " t" #TM #TN, dim3(m_tiles<P>(f2.M) * (64 / (WN * TN * 32)), 1, NZ), f2, h2[1], n2); }
#define FWD3D(WM, WN, RP, TM, TN, S) { using P = FwdP<GeomC3, AFwd<GeomC3>, WM, WN, false, RP, TM, TN>; \
    CK(hipMemset(h3[1], 0, n3 * 4)); run_dl<P, S>("fwd_c3 " #WM "x" #WN " rp" #RP " t" #TM #TN, dim3(m_tiles<P>(f3.M) * (64 / (WN * TN * 32)), 1, NZ), f3, h3[1], n3); }
    FWD2(2, 2, 0, 1, 1, 1)   // current
    { using P = FwdPT<GeomC2, 2, 2>; run<P, 1>("fwd_c2 2x2 k-major B", dim3(m_tiles<P>(f2.M), 1, NZ), f2, h2[1], n2); }
    { using P = FwdPT<GeomC2, 2, 2>; run<P, 2>("fwd_c2 2x2 k-major B teams2", dim3(m_tiles<P>(f2.M), 1, NZ), f2, h2[1], n2); }
    FWD2D(2, 2, 0, 1, 1, 3)
    FWD2D(2, 2, 0, 1, 1, 4)
    FWD2D(2, 2, 0, 2, 1, 3)
    FWD2D(4, 2, 0, 1, 1, 3)
    FWD2D(2, 1, 0, 1, 2, 3)
    FWD2D(4, 1, 0, 1, 2, 3)
    FWD2D(4, 1, 0, 1, 2, 4)
    FWD2D(2, 2, 0, 2, 1, 4)
    FWD3(2, 2, 0, 1, 1, 1)
    { using P = FwdPT<GeomC3, 2, 2>; run<P, 1>("fwd_c3 2x2 k-major B", dim3(m_tiles<P>(f3.M), 1, NZ), f3, h3[1], n3); }
    { using P = FwdPT<GeomC3, 2, 2>; run<P, 2>("fwd_c3 2x2 k-major B teams2", dim3(m_tiles<P>(f3.M), 1, NZ), f3, h3[1], n3); }
    FWD3(2, 2, 0, 1, 1, 2)   // current
    FWD3D(2, 2, 0, 1, 1, 3)
    FWD3D(2, 2, 0, 1, 1, 4)
    FWD3D(2, 2, 64, 1, 1, 3)
    FWD3D(4, 2, 0, 1, 1, 3)
    FWD3D(4, 1, 0, 1, 2, 3)
    FWD3D(2, 2, 0, 2, 1, 3)


    // dX conv3: dy [B][7][7][64] -> dx over [B][9][9][64]; dX conv2: dy [B][9][9][64] -> [B][20][20][32]
    float* dy3 = dev_rand(n3, -1.f, 1.f, 6);
    float* mask2 = dev_rand(n2, -1.f, 1.f, 7);
    float* dx2; CK(hipMalloc(&dx2, n2 * 4));
    DxArgs d3{dy3, w3, mask2, dx2, B * 81};
    float* dy2 = dev_rand(n2, -1.f, 1.f, 8);
    const size_t n1 = (size_t)B * 400 * 32;
    float* mask1 = dev_rand(n1, -1.f, 1.f, 9);
    float* dx1; CK(hipMalloc(&dx1, n1 * 4));
    DxArgs d2{dy2, w2, mask1, dx1, B * 100};

#define DX3(WM, WN, RP, TM, TN, T) { using P = DxC3P<WM, WN, RP, TM, TN>; \
    CK(hipMemset(dx2, 0, n2 * 4)); run<P, T>("dx_c3 " #WM "x" #WN " rp" #RP " t" #TM #TN " teams" #T, dim3(m_tiles<P>(d3.M) * (64 / (WN * TN * 32)), 1, 1), d3, dx2, n2); }
#define DX2(WM, WN, RP, TM, TN, T) { using P = DxC2P<WM, WN, RP, TM, TN>; \
    CK(hipMemset(dx1, 0, n1 * 4)); run<P, T>("dx_c2 " #WM "x" #WN " rp" #RP " t" #TM #TN " teams" #T, dim3(m_tiles<P>(d2.M) * (32 / (WN * TN * 32)), 4, 1), d2, dx1, n1); }

#define DX3D(WM, WN, RP, TM, TN, S) { using P = DxC3P<WM, WN, RP, TM, TN>; \
    CK(hipMemset(dx2, 0, n2 * 4)); run_dl<P, S>("dx_c3 " #WM "x" #WN " rp" #RP " t" #TM #TN, dim3(m_tiles<P>(d3.M) * (64 / (WN * TN * 32)), 1, 1), d3, dx2, n2); }
#define DX2D(WM, WN, RP, TM, TN, S) { using P = DxC2P<WM, WN, RP, TM, TN>; \
    CK(hipMemset(dx1, 0, n1 * 4)); run_dl<P, S>("dx_c2 " #WM "x" #WN " rp" #RP " t" #TM #TN, dim3(m_tiles<P>(d2.M) * (32 / (WN * TN * 32)), 4, 1), d2, dx1, n1); }
    DX3(2, 2, 0, 1, 1, 1)
    DX3D(2, 2, 0, 1, 1, 3)
    DX3D(2, 2, 0, 1, 1, 4)
    DX3D(4, 2, 0, 1, 1, 3)
    DX3D(4, 1, 0, 1, 2, 3)
    DX3D(2, 2, 0, 2, 1, 3)
    DX2D(4, 1, 0, 1, 1, 3)
    DX2D(4, 1, 0, 1, 1, 4)
    DX2D(4, 1, 0, 2, 1, 3)
    DX3(2, 2, 0, 1, 1, 2)    // current
    DX3(3, 2, 96, 1, 1, 1)   // one image (81 rows -> 96) per workgroup: 256 workgroups
    DX3(3, 2, 96, 1, 1, 2)
    DX3(3, 1, 96, 1, 2, 1)
    DX3(3, 1, 96, 1, 2, 2)
    DX3(1, 2, 96, 3, 1, 2)

    DX2(4, 1, 0, 1, 1, 1)    // current
    // ---- quantisation study: time per tile at balanced grids (multiples of 256 workgroups) vs the real grids
    printf("\n# balance study (us per launch; tiles = workgroups)\n");
    auto fwd3 = [&](int wgs_per_net) { FwdArgs g = f3; g.M = 64 * wgs_per_net; using P = FwdP<GeomC3, AFwd<GeomC3>, 2, 2, false, 0, 1, 1>;
        char nm[64]; snprintf(nm, 64, "fwd_c3 t2 %d tiles", 2 * wgs_per_net); run<P, 2>(nm, dim3(wgs_per_net, 1, NZ), g, h3[1], (size_t)g.M * 64); };
    fwd3(128); fwd3(196);
    auto fwd2 = [&](int wgs_per_net) { FwdArgs g = f2; g.M = 64 * wgs_per_net; using P = FwdP<GeomC2, AFwd<GeomC2>, 2, 2, false, 0, 1, 1>;
        char nm[64]; snprintf(nm, 64, "fwd_c2 %d tiles", 2 * wgs_per_net); run<P, 1>(nm, dim3(wgs_per_net, 1, NZ), g, h2[1], (size_t)g.M * 64); };
    fwd2(256); fwd2(324);
    auto dx3 = [&](int wgs) { DxArgs g = d3; g.M = 64 * wgs; using P = DxC3P<2, 2>;
        char nm[64]; snprintf(nm, 64, "dx_c3 t2 %d tiles", wgs); run<P, 2>(nm, dim3(wgs, 1, 1), g, dx2, (size_t)g.M * 64); };
    dx3(256); dx3(324);
    auto dx2f = [&](int wgs_per_class) { DxArgs g = d2; g.M = 128 * wgs_per_class; using P = DxC2P<4, 1>;
        char nm[64]; snprintf(nm, 64, "dx_c2 %d tiles", 4 * wgs_per_class); run<P, 1>(nm, dim3(wgs_per_class, 4, 1), g, dx1, (size_t)1); };
    dx2f(128); dx2f(192); dx2f(200);
    return 0;
}
