// What can a 2 x 14.45 MB copy cost at all on this chip?  The yardstick for k_gather (csrc/replay.hip): the same bytes, the same
// grid (512 workgroups x 256 threads, 8 x 16 B in flight per thread, all loads before the first store), but
//   A  contiguous source (no random rows, no TLB reach problem)
//   B  256 random 56 KB rows out of a 56 GB region (what the gather reads), no ChaCha, no tail fields
//   C  B with the row index from a dependent load (an index array) instead of a kernel argument computation
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/gather_floor.hip -o tools/probes/gather_floor.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint64_t ROW = 28224, STRIDE = 56576, NVEC = ROW / 16;

__global__ __launch_bounds__(256) void k_copy_rows(const uint8_t* ring, const uint64_t* rows, int use_rows, uint64_t stride, uint8_t* d0, uint8_t* d1)
{
    const uint32_t sample = blockIdx.x >> 1, chunk = blockIdx.x & 1;
    const uint64_t row = use_rows ? rows[sample] : sample;
    const u32x4* s0 = reinterpret_cast<const u32x4*>(ring + row * stride);
    const u32x4* s1 = reinterpret_cast<const u32x4*>(ring + row * stride + 28224);
    u32x4* o0 = reinterpret_cast<u32x4*>(d0 + (uint64_t)sample * ROW);
    u32x4* o1 = reinterpret_cast<u32x4*>(d1 + (uint64_t)sample * ROW);
    const uint64_t v0 = chunk * 882, v1 = v0 + 882;
    for (uint64_t v = v0 + threadIdx.x; v < v1; v += 1024) {
        u32x4 x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t w = min(v + (uint64_t)u * 256, v1 - 1);
            x[u] = __builtin_nontemporal_load(s0 + w);
            y[u] = __builtin_nontemporal_load(s1 + w);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const uint64_t w = v + (uint64_t)u * 256;
            if (w < v1) { o0[w] = x[u]; o1[w] = y[u]; }
        }
    }
}

template <class F>
static double time_us(F f, int reps = 200)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) f(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) f(i);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0 / reps;
}

int main()
{
    const uint64_t cap = 1000000;
    uint8_t *ring, *d0, *d1;
    CK(hipMalloc(&ring, cap * STRIDE));
    CK(hipMemset(ring, 1, cap * STRIDE));
    CK(hipMalloc(&d0, 256 * ROW)); CK(hipMalloc(&d1, 256 * ROW));
    // 64 different index sets (so that no launch re-reads rows a previous one left in the caches)
    std::vector<uint64_t> h(64 * 256);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = s % cap; }
    uint64_t* rows; CK(hipMalloc(&rows, h.size() * 8)); CK(hipMemcpy(rows, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    printf("A contiguous rows (stride = row bytes x 2)     : %6.2f us\n", time_us([&](int) { hipLaunchKernelGGL(k_copy_rows, dim3(512), dim3(256), 0, 0, ring, rows, 0, (uint64_t)2 * 28224, d0, d1); }));
    printf("B random rows of the 56 GB ring (index array)   : %6.2f us\n", time_us([&](int i) { hipLaunchKernelGGL(k_copy_rows, dim3(512), dim3(256), 0, 0, ring, rows + (i % 64) * 256, 1, STRIDE, d0, d1); }));
    printf("C the same 256 rows every launch (cache resident): %6.2f us\n", time_us([&](int) { hipLaunchKernelGGL(k_copy_rows, dim3(512), dim3(256), 0, 0, ring, rows, 1, STRIDE, d0, d1); }));
    // back-to-back independent launches overlap their tails; serialised launches (an empty kernel's worth of dependency) do not:
    printf("D empty-kernel launch interval                  : %6.2f us\n", time_us([&](int) { hipLaunchKernelGGL(k_copy_rows, dim3(1), dim3(64), 0, 0, ring, rows, 0, (uint64_t)2 * 28224, d0, d1); }));
    return 0;
}
