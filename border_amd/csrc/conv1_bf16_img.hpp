// conv1 forward, third form: the input images staged through LDS as bf16, the output stored as 16-byte vectors.
//
// Same arithmetic as conv1_bf16.hpp (u8 pixels exact in bf16, every f32 weight split exactly into three bf16 terms, products exact,
// f32 accumulation in the MFMA; per output the same k-steps in the same order: 16 k per step, planes lo, mid, hi): bit-identical
// results (tools/probes/conv1_rw_probe.hip compares the words).  What differs is what the kernel asks of the memory pipes:
//  * a workgroup copies its image(s) into LDS with coalesced 16-byte loads (1 764 per n_stack = 4 image instead of 200 overlapping
//    eight-byte gathers per 32-pixel unit, every pixel fetched four times) and converts every pixel to bf16 ONCE on the way (the
//    first form converted it once per fragment: ~4 VALU operations per MFMA in the item loop).  A fragment - 8 pixels of a patch row -
//    is 16 bytes at an 8-byte aligned LDS address (ds_read2_b64); pixel m of an image sits at dword 2 m + 128 oh of its plane, so the
//    32 lanes of a half wave always fall into different banks;
//  * the MFMA operand roles are swapped - rows = the 32 output channels, columns = 32 pixels - so a lane ends up with four runs of 4
//    CONSECUTIVE channels of one pixel: the NHWC epilogue is 4 global_store_dwordx4 per unit instead of 16 dword stores (the MFMA
//    computes D^T = B^T A^T element by element: same sums);
//  * IPW images per workgroup pass (1 or 2): with two images the units of a pass spread more evenly over the waves.
// NT threads = NT / 64 waves take the pass's 32-pixel units round robin; the weight fragments are read from LDS one k-step ahead, the
// pixel fragments two.  Measured and dropped (tools/probes/conv1_rw_probe.hip, conv1_bf16_rw.hpp): the weight planes in REGISTERS with one
// wave per SIMD and two or four accumulator chains per wave (11.3 us: an in-order wave pays ~12 cycles of matrix-pipe time for every LDS
// read it issues between its MFMAs, two waves per SIMD hide them), two units per wave with shared weight fragments (10.1), the
// un-swapped roles with sixteen full-row dword stores (9.4-10.0: no difference).  Where the launch's 9.6 us are at one instance of
// B = 256 (stamps of the probe): ~1.8 launch, 3.2 until the first MFMA (first-touch loads of image and weights, split, conversion,
// barrier), ~2.3 of MFMAs with the 13 units of an image spread 4 / 3 / 3 / 3 over the SIMDs, ~2.3 for the 13 MB of stores, which
// only start when the first units are done.  cnn/base.rs:26-28.
#pragma once
#include <algorithm>
#include <cstdlib>
#include "conv1_bf16.hpp"

namespace bdr {

constexpr int C1IMG_MAX_STACK = 6;   // bf16 image (14 112 B per frame) + three weight planes (12 KiB per frame) in 160 KB of LDS

#ifdef C1_TRACE   // tools/probes only: wall-clock stamps (100 MHz) of the first and the last wave of every workgroup
__device__ unsigned long long* g_c1_trace;
#define C1_TP(slot) do { if ((threadIdx.x & 63) == 0 && g_c1_trace && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == (blockDim.x >> 6) - 1)) { unsigned long long* t__ = g_c1_trace + (blockIdx.x * 2 + ((threadIdx.x >> 6) ? 1 : 0)) * 8; t__[slot] = wall_clock64(); if ((slot) == 4) t__[5] = clock64(); if ((slot) == 6) t__[7] = clock64(); } } while (0)
#else
#define C1_TP(slot) do { } while (0)
#endif

template <int NS, int IPW, int NT>
static __global__ __launch_bounds__(NT, NT / 256) void k_conv1_bf16_img(Conv1Args a)
{
    C1_TP(0);
    constexpr int PV = c1_plane_vecs(NS), KS = 4 * NS, IMG = NS * 7056, IMG16 = IMG / 16, NW = NT / 64;
    __shared__ uint4 wl[3 * PV];                 // three bf16 weight planes in fragment order
    __shared__ uint4 img[IPW * IMG16 * 2];       // the pass's images as bf16, [image][c][84][84]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.x % a.nz, wg = blockIdx.x / a.nz, nwg = gridDim.x / a.nz;
    const int i = lane & 31, h = lane >> 5;
    const int B = a.M / 400;                     // (M = B * 400 always)
    const uint8_t* x = a.x[z];

    constexpr int LPT = (IPW * IMG16 + NT - 1) / NT;   // 16-byte image loads per thread and pass
    int img0 = wg * IPW;
    // the first pass's images: every load in flight while the weights are split (unconditional, clamped: a predicated element would
    // keep the whole array out of registers)
    uint4 stage[LPT];
    const int n16_first = img0 < B ? min(IPW, B - img0) * IMG16 : 0;
    {
        const uint4* src = reinterpret_cast<const uint4*>(x + (size_t)(img0 < B ? img0 : 0) * IMG);
#pragma unroll
        for (int k = 0; k < LPT; ++k) { const int t = tid + NT * k; stage[k] = src[t < n16_first ? t : 0]; }
    }
    {
        // weight split: all of a thread's 8 * WPT loads are in flight before the first one is used (one round trip, not WPT)
        constexpr int WPT = (PV + NT - 1) / NT;
        const float* w1 = a.w1[z];
        float wv[WPT][8];
#pragma unroll
        for (int q = 0; q < WPT; ++q) {
            int t = tid + NT * q;
            t = t < PV ? t : PV - 1;
            const int n = t & 31, hh = (t >> 5) & 1, sx = t >> 6;
#pragma unroll
            for (int j = 0; j < 8; ++j) wv[q][j] = w1[(size_t)(16 * sx + 8 * hh + j) * 32 + n];
        }
#pragma unroll
        for (int q = 0; q < WPT; ++q) {
            const int t = tid + NT * q;
            uint32_t hi[8], mid[8], lo[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                hi[j] = __float_as_uint(wv[q][j]) & 0xffff0000u;
                const float r1 = wv[q][j] - __uint_as_float(hi[j]);     // exact
                mid[j] = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(mid[j]);          // exact, <= 8 significant bits
                lo[j] = __float_as_uint(r2) & 0xffff0000u;
            }
            if (t < PV) {
                wl[0 * PV + t] = uint4{(hi[0] >> 16) | hi[1], (hi[2] >> 16) | hi[3], (hi[4] >> 16) | hi[5], (hi[6] >> 16) | hi[7]};
                wl[1 * PV + t] = uint4{(mid[0] >> 16) | mid[1], (mid[2] >> 16) | mid[3], (mid[4] >> 16) | mid[5], (mid[6] >> 16) | mid[7]};
                wl[2 * PV + t] = uint4{(lo[0] >> 16) | lo[1], (lo[2] >> 16) | lo[3], (lo[4] >> 16) | lo[5], (lo[6] >> 16) | lo[7]};
            }
        }
    }
    // bias of this lane's four channel runs [8 q + 4 h, + 4)
    f32x4 bq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const f32x4*>(a.bias[z] + 8 * q + 4 * h);
    C1_TP(1);
#pragma unroll
    for (int k = 0; k < LPT; ++k) {
        const int t = tid + NT * k;
        if (t < n16_first) {
            img[2 * t] = __builtin_bit_cast(uint4, u8x8_to_bf16(stage[k].x, stage[k].y));
            img[2 * t + 1] = __builtin_bit_cast(uint4, u8x8_to_bf16(stage[k].z, stage[k].w));
        }
    }
    C1_TP(2);
    __syncthreads();
    C1_TP(3);
    const uint4* wlane = wl + h * 32 + i;        // fragment (plane pl, k-step s) of this lane (output channel i, k half h): wlane[pl * PV + 64 s]

    float* out = a.out[z];
    const uint8_t* imgb = reinterpret_cast<const uint8_t*>(img);
    C1_TP(4);

    for (bool first = true; img0 < B; img0 += nwg * IPW, first = false) {
        const int nimg = min(IPW, B - img0), Mloc = nimg * 400, units = (Mloc + 31) / 32;
        if (!first) {   // more images than workgroup passes: stage the next ones where the last ones were
            __syncthreads();                 // every wave has read its last fragment of the previous pass
            const uint4* src = reinterpret_cast<const uint4*>(x + (size_t)img0 * IMG);
            for (int t = tid; t < nimg * IMG16; t += NT) {
                const uint4 v = src[t];
                img[2 * t] = __builtin_bit_cast(uint4, u8x8_to_bf16(v.x, v.y));
                img[2 * t + 1] = __builtin_bit_cast(uint4, u8x8_to_bf16(v.z, v.w));
            }
            __syncthreads();
        }
        // a unit's fragments: lane = pixel i of the unit, row parity h; k-step s = patch rows 2 (s % 4) + h of channel s / 4
        auto unit_base = [&](int unit) -> const uint8_t* {
            int m = unit * 32 + i;
            m = m < Mloc ? m : Mloc - 1;
            const int bl = m >= 400 ? 1 : 0, rem = m - bl * 400;
            const int oh = rem / 20, ow = rem - oh * 20;
            return imgb + 2 * (bl * IMG + (oh * 4 + h) * 84 + ow * 4);
        };
        auto frag = [&](const uint8_t* p, int s) -> bf16x8 {
            const uint2* q = reinterpret_cast<const uint2*>(p + 2 * ((s >> 2) * 7056 + ((s & 3) * 2) * 84));
            const uint2 lo = q[0], hi = q[1];
            return __builtin_bit_cast(bf16x8, uint4{lo.x, lo.y, hi.x, hi.y});
        };
        auto store_unit = [&](int unit, const f32x16& acc) {   // lane = pixel i, four runs of 4 consecutive channels [8 q + 4 h, + 4)
            const int m = unit * 32 + i;
#ifdef C1_ABL_NOSTORE
            if (m < Mloc && acc[0] == 12345.678f) {
#else
            if (m < Mloc) {
#endif
                float* o = out + ((size_t)img0 * 400 + m) * 32 + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { const float t = acc[4 * q + e] * (1.0f / 255.0f) + bq[q][e]; v[e] = t > 0.f ? t : 0.f; }
                    *reinterpret_cast<f32x4*>(o + 8 * q) = v;
                }
            }
        };
        for (int u = wave; u < units; u += NW) {
            const uint8_t* p = unit_base(u);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            // pixel fragments two k-steps ahead of the MFMAs that use them, weight fragments one (sched_barrier keeps hipcc from hoisting
            // all the reads of a unit: 4 VGPRs each)
            bf16x8 fa = frag(p, 0), fb = frag(p, 1), wa[3], wb[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wa[pl] = __builtin_bit_cast(bf16x8, wlane[pl * PV]);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const bf16x8 fc = s + 2 < KS ? frag(p, s + 2) : fa;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wb[pl] = s + 1 < KS ? __builtin_bit_cast(bf16x8, wlane[pl * PV + 64 * (s + 1)]) : wa[pl];
#pragma unroll
                for (int pl = 2; pl >= 0; --pl) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wa[pl], fa, acc, 0, 0, 0);   // small terms first
                fa = fb; fb = fc;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wa[pl] = wb[pl];
                __builtin_amdgcn_sched_barrier(0);
            }
            store_unit(u, acc);
        }
        C1_TP(6);
    }
}

// waves = 8 or 16 per workgroup
inline hipError_t launch_conv1_bf16_img(int ns, int ipw, int waves, dim3 grid, hipStream_t st, const Conv1Args& c)
{
    if (ns < 1 || ns > C1IMG_MAX_STACK || ipw < 1 || ipw > 2 || (waves != 8 && waves != 16) || (ns > 4 && ipw == 2)) return hipErrorInvalidValue;
    switch (ns * 4 + (ipw - 1) * 2 + (waves == 16)) {
#define BDR_C1IMG_CASE1(N) \
        case N * 4 + 0: hipLaunchKernelGGL((k_conv1_bf16_img<N, 1, 512>), grid, dim3(512), 0, st, c); break; \
        case N * 4 + 1: hipLaunchKernelGGL((k_conv1_bf16_img<N, 1, 1024>), grid, dim3(1024), 0, st, c); break;
#define BDR_C1IMG_CASE(N) BDR_C1IMG_CASE1(N) \
        case N * 4 + 2: hipLaunchKernelGGL((k_conv1_bf16_img<N, 2, 512>), grid, dim3(512), 0, st, c); break; \
        case N * 4 + 3: hipLaunchKernelGGL((k_conv1_bf16_img<N, 2, 1024>), grid, dim3(1024), 0, st, c); break;
        BDR_C1IMG_CASE(1) BDR_C1IMG_CASE(2) BDR_C1IMG_CASE(3) BDR_C1IMG_CASE(4) BDR_C1IMG_CASE1(5) BDR_C1IMG_CASE1(6)
#undef BDR_C1IMG_CASE
#undef BDR_C1IMG_CASE1
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// conv1 forward of nz network instances on B images each: the staged-image form where its LDS image fits (bf16 image(s) + three weight
// planes <= 160 KB: n_stack <= 6, two images per pass up to n_stack 4), the direct form of conv1_bf16.hpp otherwise.  Same bits either way.
// BDR_C1_FORM (diagnostic, read once): -1 = always the direct form, 8 / 16 = waves per workgroup of the staged form (default 8).
inline int conv1_form_env()
{
    static const int form = [] { const char* e = getenv("BDR_C1_FORM"); return e ? atoi(e) : 8; }();
    return form;
}
inline hipError_t conv1_forward(int ns, int B, hipStream_t st, const Conv1Args& c, int cus = 256)
{
    const int form = conv1_form_env();
    if (ns <= C1IMG_MAX_STACK && form >= 0 && c.M == B * 400) {
        const int per_inst = std::max(1, cus / c.nz);
        const int ipw = (ns <= 4 && B > per_inst) ? 2 : 1;
        const int g = std::max(1, std::min(per_inst, (B + ipw - 1) / ipw));
        return launch_conv1_bf16_img(ns, ipw, form == 16 ? 16 : 8, dim3(g * c.nz), st, c);
    }
    const int items = (c.M + 31) / 32;
    const int g = std::max(1, std::min(512 / c.nz, (items + 7) / 8));
    return launch_conv1_bf16(ns, dim3(g * c.nz), st, c);
}

}  // namespace bdr
