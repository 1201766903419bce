// conv1 weight gradient on the bf16 matrix cores with EXACT operands.
//
//   G[ko][n] = sum_m X[m][ko] * dY[m][n],   m = (b, oh, ow) over B*400 positions, ko = (c, kh, kw), n = 32
//   X[m][ko] = img[b][c][4*oh + kh][4*ow + kw]   (u8, exact in bf16)
//   dY       = gradient w.r.t. conv1's pre-activation, f32, split into three bf16 terms (exact 3 x 8-bit
//              truncation split, as conv1_bf16.hpp does for the weights)
// so every product is exact and only the f32 accumulation of the MFMA rounds: the error class of the FP32
// path at 16x the matrix rate (cnn/base.rs:26-28 backward; the 1/255 is applied by the partial reduction).
//
// v_mfma_f32_32x32x16_bf16 wants, per lane, 8 CONSECUTIVE reduction elements of its row / column:
//  * A (rows ko): for fixed (c,kh,kw) four consecutive ow are the bytes 4*ow + kw of one image row, i.e. byte
//    kw%4 of four consecutive dwords -> one dwordx4 load + a per-lane byte extract.  20 = 5 x 4: the reduction is
//    walked in QUADS of 4 ow that never straddle an image row; a lane's 8 elements are two quads.
//  * B (columns n): dY is [m][32]; the workgroup splits its image's 400 x 32 block once into three k-major bf16
//    planes in LDS ([plane][n][m], rows padded to 408 for conflict-free ds_read_b128).
// One workgroup = one image at a time (400 positions = 25 MFMA k-steps), 8 waves = (input channel c, kh half):
// one 32-row tile each; workgroups stride over the batch and write one partial [256*32 + 32] each
// (bias gradient = column sums of dY, fixed order), summed by k_reduce_partials(3).
#pragma once
#include "../../border_amd/csrc/conv1_dw_bf16.hpp"

namespace bdr {

struct Conv1DwArgs_v1_unused {
    const uint8_t* x;     // [B][n_stack][84][84] u8
    const float* dy;      // [B*400][32] f32
    float* part;          // [gridDim.x][part_stride]
    size_t part_stride;   // floats (64*n_stack*32 + 32)
    int B;
};

constexpr int C1DW_LDM_V1 = 408;   // bf16 per plane row (400 + 8)
constexpr int C1DW_PF_V1 = 4;      // k-steps of pixel loads in flight

struct __attribute__((packed, aligned(4))) U32x4A4_v1 { uint32_t x, y, z, w; };   // dword-aligned 16-byte load

// bf16 pair from the bytes (d0 >> sh) & 255, (d1 >> sh) & 255 (integers 0..255: bf16 = upper half of the f32)
__device__ __forceinline__ uint32_t u8pair_to_bf16(uint32_t d0, uint32_t d1, uint32_t sh)
{
    const uint32_t f0 = __float_as_uint((float)((d0 >> sh) & 255u)), f1 = __float_as_uint((float)((d1 >> sh) & 255u));
    return __builtin_amdgcn_perm(f1, f0, 0x07060302u);
}

#ifdef C1DW_TRACE   // tools/probes only
__device__ unsigned long long* g_c1dw_trace;
#define C1DW_TP(slot) do { if (threadIdx.x == 0 && g_c1dw_trace) g_c1dw_trace[blockIdx.x * 4 + (slot)] = wall_clock64(); } while (0)
#else
#define C1DW_TP(slot) do { } while (0)
#endif

// NS = n_stack.  The 2 * NS (input channel, kh half) row tiles of the image are dealt to the 8 waves: one each for NS = 4 (the
// reference's examples), TPW = 2 per wave for NS = 5 ... 8, idle waves (they still split dY and meet the barriers) below 4.
template <int NS>
static __global__ __launch_bounds__(512) void k_conv1_dw_bf16_v1(Conv1DwArgs a)
{
    C1DW_TP(0);
    constexpr int TASKS = 2 * NS, TPW = (TASKS + 7) / 8;
    __shared__ __attribute__((aligned(16))) uint16_t planes[3 * 32 * C1DW_LDM];   // 78 336 B
    __shared__ float sred[16][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    const int pn = tid & 31, pmg = tid >> 5;                      // prologue role: column n, quad group (16 groups)

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    // per-lane A geometry of task t = wave + 8 * j -> (input channel c = t / 2, kh half tile = t % 2): kh = 4*tile + i/8, kw = i%8
    const uint32_t sh = 8u * (uint32_t)(i & 3);
    int c_of[TPW], rowoff[TPW]; bool live[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int task = wave + 8 * t;
        live[t] = task < TASKS;
        c_of[t] = (live[t] ? task : 0) >> 1;
        rowoff[t] = (4 * (task & 1) + (i >> 3)) * 84 + (i & 4);
    }

    for (int img = blockIdx.x; img < a.B; img += gridDim.x) {
        const uint8_t* ximg = a.x + (size_t)img * (NS * 7056);
        auto load_step = [&](int t, int s, U32x4A4 (&d)[2]) {   // step s: quads 4s + 2g, 4s + 2g + 1
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int quad = 4 * s + 2 * g + q;
                const int oh = quad / 5, ow0 = 4 * (quad - 5 * oh);
                d[q] = *reinterpret_cast<const U32x4A4*>(ximg + c_of[t] * 7056 + (4 * oh) * 84 + 4 * ow0 + rowoff[t]);
            }
        };
        // pixel fragments run C1DW_PF k-steps ahead of the MFMAs through a register ring (a step is ~0.15 us of
        // matrix work, an L2 hit is longer); the first ones are in flight while dY is split
        U32x4A4 ring[TPW][C1DW_PF + 1][2];
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int s0 = 0; s0 < C1DW_PF; ++s0) load_step(t, s0, ring[t][s0]);

        // ---- dY block of this image -> three k-major bf16 planes; bias partial sums
        {
            const float* dyb = a.dy + (size_t)img * 400 * 32;
            float s = 0.f;
            for (int quad = pmg; quad < 100; quad += 16) {
                uint32_t hi[4], mid[4], lo[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float y = dyb[(size_t)(4 * quad + j) * 32 + pn];
                    s += y;
                    hi[j] = __float_as_uint(y) & 0xffff0000u;
                    const float r1 = y - __uint_as_float(hi[j]);            // exact
                    mid[j] = __float_as_uint(r1) & 0xffff0000u;
                    const float r2 = r1 - __uint_as_float(mid[j]);          // exact, <= 8 significant bits
                    lo[j] = __float_as_uint(r2) & 0xffff0000u;
                }
                const int o = pn * C1DW_LDM + 4 * quad;
                *reinterpret_cast<uint2*>(&planes[0 * 32 * C1DW_LDM + o]) = uint2{(hi[0] >> 16) | hi[1], (hi[2] >> 16) | hi[3]};
                *reinterpret_cast<uint2*>(&planes[1 * 32 * C1DW_LDM + o]) = uint2{(mid[0] >> 16) | mid[1], (mid[2] >> 16) | mid[3]};
                *reinterpret_cast<uint2*>(&planes[2 * 32 * C1DW_LDM + o]) = uint2{(lo[0] >> 16) | lo[1], (lo[2] >> 16) | lo[3]};
            }
            sred[pmg][pn] = s;
        }
        __syncthreads();
        C1DW_TP(1);
        if (tid < 32) {
            float t = sred[0][tid];
#pragma unroll
            for (int k = 1; k < 16; ++k) t += sred[k][tid];
            bsum += t;
        }

        // ---- 25 k-steps of 16 positions
#pragma unroll
        for (int s = 0; s < 25; ++s) {
            uint4 bq[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl)
                bq[pl] = *reinterpret_cast<const uint4*>(&planes[(pl * 32 + i) * C1DW_LDM + 16 * s + 8 * g]);
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                U32x4A4 (&cur)[2] = ring[t][s % (C1DW_PF + 1)];
                if (s + C1DW_PF < 25) load_step(t, s + C1DW_PF, ring[t][(s + C1DW_PF) % (C1DW_PF + 1)]);
                uint4 av;
                av.x = u8pair_to_bf16(cur[0].x, cur[0].y, sh);
                av.y = u8pair_to_bf16(cur[0].z, cur[0].w, sh);
                av.z = u8pair_to_bf16(cur[1].x, cur[1].y, sh);
                av.w = u8pair_to_bf16(cur[1].z, cur[1].w, sh);
                const bf16x8 af = __builtin_bit_cast(bf16x8, av);
#pragma unroll
                for (int pl = 2; pl >= 0; --pl)   // small terms first
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bq[pl]), acc[t], 0, 0, 0);
            }
        }
        C1DW_TP(2);
        __syncthreads();   // planes / sred are rewritten by the next image
    }

    float* part = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        if (!live[t]) continue;   // (wave-uniform)
        const int task = wave + 8 * t;                            // rows [32 * task, 32 * task + 32) = (c * 64 + tile * 32 + row)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
            part[(size_t)(task * 32 + row) * 32 + i] = acc[t][r];
        }
    }
    if (tid < 32) part[64 * NS * 32 + tid] = bsum;
    C1DW_TP(3);
}

inline hipError_t launch_conv1_dw_bf16_v1(int ns, dim3 grid, hipStream_t st, const Conv1DwArgs& d)
{
    switch (ns) {
#define BDR_C1DW_CASE(N) case N: hipLaunchKernelGGL(k_conv1_dw_bf16_v1<N>, grid, dim3(512), 0, st, d); break;
        BDR_C1DW_CASE(1) BDR_C1DW_CASE(2) BDR_C1DW_CASE(3) BDR_C1DW_CASE(4) BDR_C1DW_CASE(5) BDR_C1DW_CASE(6) BDR_C1DW_CASE(7) BDR_C1DW_CASE(8)
#undef BDR_C1DW_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace bdr
