// Probe: issue rate of v_mfma_f32_32x32x16_bf16 from one wave per SIMD, NC independent accumulator chains, operands in registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NC, int NW>
__global__ __launch_bounds__(256, 1) void k_rate(const uint4* in, float* out, unsigned long long* cyc, int reps)
{
    bf16x8 a[NC], w[NW];
    for (int q = 0; q < NC; ++q) a[q] = __builtin_bit_cast(bf16x8, in[threadIdx.x + 256 * q]);
    for (int q = 0; q < NW; ++q) w[q] = __builtin_bit_cast(bf16x8, in[threadIdx.x + 256 * (NC + q)]);
    f32x16 acc[NC];
    for (int q = 0; q < NC; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < reps; ++it) {
#pragma unroll
        for (int s = 0; s < NW; ++s)
#pragma unroll
            for (int q = 0; q < NC; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[s], a[q], acc[q], 0, 0, 0);
    }
    float sum = 0.f;
    for (int q = 0; q < NC; ++q) for (int r = 0; r < 16; ++r) sum += acc[q][r];
    const unsigned long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int NC, int NW> void run(const uint4* in, float* out, unsigned long long* cyc, int grid)
{
    const int reps = 8;
    for (int k = 0; k < 3; ++k) { hipLaunchKernelGGL((k_rate<NC, NW>), dim3(grid), dim3(256), 0, 0, in, out, cyc, reps); hipDeviceSynchronize(); }
    std::vector<unsigned long long> h(grid); hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("chains %d, %2d weight fragments, grid %3d: %.1f cycles per MFMA\n", NC, NW, grid, s / grid / (double)(reps * NC * NW));
}
int main()
{
    uint4* in; float* out; unsigned long long* cyc;
    hipMalloc(&in, 256 * 64 * 16); hipMemset(in, 0x3c, 256 * 64 * 16); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
    for (int grid : {1, 256}) {
        run<1, 16>(in, out, cyc, grid); run<2, 16>(in, out, cyc, grid); run<4, 16>(in, out, cyc, grid); run<4, 48>(in, out, cyc, grid); run<2, 48>(in, out, cyc, grid);
    }
    return 0;
}
