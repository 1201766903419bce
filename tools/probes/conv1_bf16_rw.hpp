// conv1 forward, second form: the three bf16 weight planes live in each wave's REGISTERS.
//
// Same arithmetic as conv1_bf16.hpp (u8 pixels exact in bf16, every f32 weight split exactly into three bf16 terms, products exact,
// f32 accumulation in the MFMA; per output the same k-steps in the same order: 16 k per step, planes lo, mid, hi) - bit-identical
// results - but a different machine mapping.  What bound the first form (tools/probes/conv1_probe.hip, rocprofv3 PMC): every MFMA
// fetched its 1 KB weight fragment from LDS (4 SIMDs x 1 KB per 32 matrix cycles = the CU's whole 128 B/clk), every wave did one or
// two items, so load -> MFMA -> store ran as one latency chain per wave with 4 waves per SIMD finishing - and storing - together.
// Here a workgroup is ONE wave per SIMD (256 threads, one workgroup per CU, up to 512 registers per lane): after the prologue has
// split the weights into LDS once, each wave copies ALL 4 * NS * 3 fragments into registers (192 VGPRs at n_stack = 4) and the item
// loop touches no LDS at all.  A wave walks a contiguous run of 32-pixel units two at a time (two independent accumulator chains
// sharing every weight fragment), with the next pair's pixel rows in flight during the MFMAs and the previous pair's stores draining
// behind them.  cnn/base.rs:26-28.
#pragma once
#include <type_traits>
#include "../../border_amd/csrc/conv1_bf16.hpp"

namespace bdr {

constexpr int C1RW_MAX_STACK = 4;   // 48 * n_stack weight registers: n_stack 5 ... 8 stay on the LDS form (conv1_bf16.hpp)

template <int NS>
static __global__ __launch_bounds__(256, 1) void k_conv1_bf16_rw(Conv1Args a)
{
    constexpr int PV = c1_plane_vecs(NS), KS = 4 * NS;
    __shared__ uint4 wl[3 * PV];   // prologue only: three bf16 weight planes in B-fragment order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.x % a.nz, wg = blockIdx.x / a.nz, nwg = gridDim.x / a.nz;
    const int i = lane & 31, h = lane >> 5;
    const int units = (a.M + 31) / 32;
    const uint8_t* x = a.x[z];

    // this wave's run of units: [u0, u1)
    const int nw = nwg * 4, w = wg * 4 + wave;
    const int u0 = (int)((long long)units * w / nw), u1 = (int)((long long)units * (w + 1) / nw);

    auto load_unit = [&](int unit, uint2 (&r)[KS]) {   // 2 * KS patch rows of 8 pixels (lane = pixel i, row parity h)
        int m = unit * 32 + i;
        m = m < a.M ? m : a.M - 1;
        const int b = m / 400, rem = m - b * 400;
        const int oh = rem / 20, ow = rem - oh * 20;
        const uint8_t* p = x + (size_t)b * (NS * 7056) + (oh * 4 + h) * 84 + ow * 4;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(p + (s >> 2) * 7056 + ((s & 3) * 2) * 84);
            r[s] = uint2{q[0], q[1]};
        }
    };

    // the first pair's pixels are in flight while the weights are split
    uint2 n0[KS], n1[KS];
    int u = u0;
    if (u < u1) load_unit(u, n0);
    if (u + 1 < u1) load_unit(u + 1, n1);
    {
        // weight split: all of a thread's 8 * WPT loads are in flight before the first one is used (one round trip, not WPT)
        constexpr int WPT = PV / 256;
        const float* w1 = a.w1[z];
        float wv[WPT][8];
#pragma unroll
        for (int q = 0; q < WPT; ++q) {
            const int t = tid + 256 * q, n = t & 31, hh = (t >> 5) & 1, sx = t >> 6;
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[q][j] = w1[(size_t)(16 * sx + 8 * hh + j) * 32 + n];
        }
#pragma unroll
        for (int q = 0; q < WPT; ++q) {
            const int t = tid + 256 * q;
            uint32_t hi[8], mid[8], lo[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                hi[j] = __float_as_uint(wv[q][j]) & 0xffff0000u;
                const float r1 = wv[q][j] - __uint_as_float(hi[j]);     // exact
                mid[j] = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(mid[j]);          // exact, <= 8 significant bits
                lo[j] = __float_as_uint(r2) & 0xffff0000u;
            }
            wl[0 * PV + t] = uint4{(hi[0] >> 16) | hi[1], (hi[2] >> 16) | hi[3], (hi[4] >> 16) | hi[5], (hi[6] >> 16) | hi[7]};
            wl[1 * PV + t] = uint4{(mid[0] >> 16) | mid[1], (mid[2] >> 16) | mid[3], (mid[4] >> 16) | mid[5], (mid[6] >> 16) | mid[7]};
            wl[2 * PV + t] = uint4{(lo[0] >> 16) | lo[1], (lo[2] >> 16) | lo[3], (lo[4] >> 16) | lo[5], (lo[6] >> 16) | lo[7]};
        }
    }
    __syncthreads();
    bf16x8 wr[3][KS];   // every weight fragment of this lane: [plane][k-step]
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int s = 0; s < KS; ++s) wr[pl][s] = __builtin_bit_cast(bf16x8, wl[pl * PV + (s * 2 + h) * 32 + i]);

    const float bias = a.bias[z][i];
    float* out = a.out[z];
    asm volatile("" ::"v"(bias));   // the bias lands here, not in front of the first store (stores share vmcnt with the pixel loads)

    auto store_unit = [&](int unit, const f32x16& acc) {
        const int m0 = unit * 32;
        if (m0 + 32 <= a.M) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mo = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = acc[r] * (1.0f / 255.0f) + bias;
                out[(size_t)mo * 32 + i] = v > 0.f ? v : 0.f;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mo = m0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = acc[r] * (1.0f / 255.0f) + bias;
                if (mo < a.M) out[(size_t)mo * 32 + i] = v > 0.f ? v : 0.f;
            }
        }
    };

    for (; u + 1 < u1; u += 2) {   // pairs of units: two accumulator chains share every weight fragment
        uint2 c0[KS], c1[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) { c0[s] = n0[s]; c1[s] = n1[s]; }
        if (u + 2 < u1) load_unit(u + 2, n0);
        if (u + 3 < u1) load_unit(u + 3, n1);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 a0 = u8x8_to_bf16(c0[s].x, c0[s].y), a1 = u8x8_to_bf16(c1[s].x, c1[s].y);
#pragma unroll
            for (int pl = 2; pl >= 0; --pl) {   // small terms first
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wr[pl][s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, wr[pl][s], acc1, 0, 0, 0);
            }
        }
        store_unit(u, acc0);
        store_unit(u + 1, acc1);
    }
    if (u < u1) {   // odd tail: one unit
        f32x16 acc0;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bf16x8 a0 = u8x8_to_bf16(n0[s].x, n0[s].y);
#pragma unroll
            for (int pl = 2; pl >= 0; --pl) acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, wr[pl][s], acc0, 0, 0, 0);
        }
        store_unit(u, acc0);
    }
}

// grid for the register form: one workgroup per CU over all instances
inline int conv1_rw_groups(int nz, int M, int cus = 256)
{
    const int units = (M + 31) / 32;
    int g = cus / nz;
    if (g < 1) g = 1;
    const int need = (units + 3) / 4;   // no more waves than units
    return g < need ? g : (need < 1 ? 1 : need);
}

inline hipError_t launch_conv1_bf16_rw(int ns, dim3 grid, hipStream_t st, const Conv1Args& c)
{
    switch (ns) {
#define BDR_C1RW_CASE(N) case N: hipLaunchKernelGGL(k_conv1_bf16_rw<N>, grid, dim3(256), 0, st, c); break;
        BDR_C1RW_CASE(1) BDR_C1RW_CASE(2) BDR_C1RW_CASE(3) BDR_C1RW_CASE(4)
#undef BDR_C1RW_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace bdr
