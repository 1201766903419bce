// Phase timing of the one-workgroup DQN step (border_amd/csrc/mlp_fused.hpp) on CartPole shapes: where do the microseconds go?
#define MF_TRACE 1
#include "mlp_fused.hpp"
#include <cstdio>
#include <vector>
namespace bdr { thread_local char g_err[512]; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    const int B = 32, L = 3, in_dim = 4, A = 2;
    int Kp[3] = {64, 64, 64}, Np[3] = {64, 64, 64};
    MlpFusedArgs f{};
    f.L = L; f.nz = 2; f.B = B; f.A = A; f.in_dim = in_dim;
    size_t o = 0;
    for (int i = 0; i < L; ++i) { f.Kp[i] = Kp[i]; f.Np[i] = Np[i]; f.relu[i] = i < L - 1; f.w[i] = o; o += Kp[i] * Np[i]; f.b[i] = o; o += Np[i]; f.in_rows_l[i] = i == 0 ? in_dim : 64; }
    f.total = o;
    auto dev = [&](size_t n, float fill) { float* p; hipMalloc(&p, n * 4); std::vector<float> h(n, fill); for (size_t i = 0; i < n; ++i) h[i] = fill * (float)((i * 37 % 101) - 50) / 50.f; hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice); return p; };
    f.q = dev(o, 0.1f); f.q_tgt = dev(o, 0.1f); f.grad = dev(o, 0.f); f.m = dev(o, 0.f); f.v = dev(o, 0.f);
    float* obs = dev(B * in_dim, 1.f); float* nobs = dev(B * in_dim, 1.f);
    const float* par[2] = {f.q, f.q_tgt}; const float* rows[2] = {obs, nobs};
    for (int z = 0; z < 2; ++z) { f.params[z] = par[z]; f.in_rows[z] = rows[z]; f.x_in[z] = dev(B * 64, 0.f); for (int i = 0; i < L; ++i) f.act[z][i] = dev(B * 64, 0.f); }
    for (int i = 0; i < L; ++i) f.dy[i] = dev(B * 64, 0.f);
    long long* act; hipMalloc(&act, B * 8); hipMemset(act, 0, B * 8);
    f.actions = (const uint8_t*)act; f.act_bytes = 8; f.reward = dev(B, 1.f); int8_t* term; hipMalloc(&term, B); hipMemset(term, 0, B); f.term = term;
    f.pred = dev(B, 0.f); f.tgt = dev(B, 0.f); f.loss_row = dev(B, 0.f); f.loss = dev(4, 0.f);
    f.gamma = 0.99f; f.loss_kind = 0; f.td_abs = dev(B, 0.f);
    unsigned* err; hipMalloc(&err, 16); hipMemset(err, 0, 16); f.err = err;
    f.adam = adam_scalars_for(false, 1e-3, 0, 0, 0, 0, 1); f.do_adam = 1; f.do_track = 1; f.tau = 0.01f; f.omt = 0.99f;
    unsigned long long* tr; CK(hipMalloc(&tr, 32 * 8)); f.trace = tr;
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t lds_bytes = mf_lds_plan(f);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_dqn_mlp_step_lds), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  for (int variant = 0; variant < 2; ++variant) {
    printf("---- %s\n", variant ? "k_dqn_mlp_step_lds (matrices resident in LDS)" : "k_dqn_mlp_step (phases exchange through global memory)");
    double sum[12] = {0};
    const int reps = 200;
    for (int r = 0; r < reps + 20; ++r) {
        if (variant) hipLaunchKernelGGL(k_dqn_mlp_step_lds, dim3(1), dim3(512), lds_bytes, st, f);
        else hipLaunchKernelGGL(k_dqn_mlp_step, dim3(1), dim3(512), 0, st, f);
        CK(hipStreamSynchronize(st));
        unsigned long long h[18];
        CK(hipMemcpy(h, tr, sizeof h, hipMemcpyDeviceToHost));
        const int idx[10] = {0, 1, 2, 3, 4, 6, 7, 8, 9, 11};
        if (r == reps + 19) printf("shader clock over the kernel: %.0f MHz\n", (double)(h[13] - h[12]) / ((double)(h[11] - h[0]) / 100.0));
        if (r >= 20) for (int k = 1; k < 10; ++k) sum[idx[k]] += (double)(h[idx[k]] - h[idx[k - 1]]) / 100.0;
    }
    const char* nm[12] = {"", "pack", "fwd0", "fwd1", "fwd2", "-", "td+loss", "bwd2", "bwd1", "bwd0", "-", "adam+track"};
    double tot = 0;
    for (int k = 1; k < 12; ++k) { if (k == 5 || k == 10) continue; printf("%-10s %.2f us\n", nm[k], sum[k] / reps); tot += sum[k] / reps; }
    printf("total in-kernel %.2f us\n", tot);
  }
    return 0;
}
