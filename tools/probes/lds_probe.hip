// LDS fragment-read cost of the igemm tile layouts (cycles per wave-instruction with 4 waves/CU reading).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/lds_probe.hip -o tools/probes/lds_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// mode 0: b128 linear (lane*16B)   1: b128 rows of 36 floats (row = lane&31, k-half = lane>>5)
// mode 2: b128 rows of 32 floats XOR-swizzled   3: b32 [k][64] B pattern (row 4h+s, col i)
// mode 4: b128 rows of 32 floats, no swizzle (worst case)   5: b64 rows of 36   6: b128 rows of 40 floats
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) float s[16384];
    for (int i = threadIdx.x; i < 16384; i += 256) s[i] = (float)i;
    __syncthreads();
    const int lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    int off;  // floats
    if (MODE == 0) off = lane * 4;
    else if (MODE == 1) off = i * 36 + 4 * h;
    else if (MODE == 2) off = i * 32 + ((h ^ (i & 7)) * 4);
    else if (MODE == 3) off = 4 * h * 64 + i;
    else if (MODE == 4) off = i * 32 + 4 * h;
    else if (MODE == 5) off = i * 36 + 4 * h;
    else off = i * 40 + 4 * h;
    off += (threadIdx.x >> 6) * 2048;
    f32x4 acc = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 3) {
                float v;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((off + u * 64) * 4), "n"(0));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc[0] += 0.f * v;
            } else if (MODE == 5) {
                float2 v;
                asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"((off + (u & 3) * 8) * 4));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc[0] += 0.f * v.x;
            } else {
                f32x4 v;
                asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"((off + ((MODE == 0) ? u * 256 : (u & 3) * 8)) * 4));
                asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
                acc[0] += 0.f * v[0];
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const long long t1 = clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int bytes_per_lane)
{
    float* out; long long* cyc; const int NB = 256, IT = 2000;
    CK(hipMalloc(&out, NB * 256 * 4)); CK(hipMalloc(&cyc, NB * 8));
    hipLaunchKernelGGL(k<MODE>, dim3(NB), dim3(256), 0, 0, out, cyc, IT);
    hipLaunchKernelGGL(k<MODE>, dim3(NB), dim3(256), 0, 0, out, cyc, IT);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(NB); CK(hipMemcpy(h.data(), cyc, NB * 8, hipMemcpyDeviceToHost));
    double m = 0; for (auto c : h) m += c; m /= NB;
    const double per = m / (IT * 8.0);   // cycles per instruction per wave, 4 waves sharing the LDS
    printf("%-44s %7.2f cycles / wave-instr (4 waves)  => %6.1f B/clk/CU\n", name, per, 4.0 * 64 * bytes_per_lane / per);
    CK(hipFree(out)); CK(hipFree(cyc));
}
int main()
{
    run<0>("b128 linear", 16);
    run<1>("b128 rows of 36 floats (igemm A tile)", 16);
    run<6>("b128 rows of 40 floats", 16);
    run<2>("b128 rows of 32 floats, XOR swizzle (dl)", 16);
    run<4>("b128 rows of 32 floats, no swizzle", 16);
    run<5>("b64  rows of 36 floats", 8);
    run<3>("b32  [k][64] B tile (row 4h+s, col i)", 4);
    return 0;
}
