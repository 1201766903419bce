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
// One workgroup = one image at a time (400 positions = 25 MFMA k-steps).  A wave's 32-row tile is (channel pair, column phase):
// rows = 2 channels x 8 kh x the two kw with kw % 4 = phase.  The byte a lane extracts from its dwords is then the SAME for the whole
// wave - v_cvt_f32_ubyte<phase> straight from the loaded dword, 1.5 VALU operations per element instead of 2.5 with a per-lane shift -
// and the k loop, which was bound by those conversions (2 x the matrix time: tools/probes/c1dw_trace.hip), runs a phase-specialised
// body.  The image itself is staged through LDS (coalesced 16-byte loads, rows padded to a conflict-free stride): the fragments are
// strided gathers over 2 channels x 8 rows, free in LDS and 4 x the cache lines per load straight from memory.  The 16 reduction
// elements of a k-step are 8 positions of the image's upper half (lane half 0) and 8 of its lower half (lane half 1), so every LDS
// offset of the loop is a compile-time constant.  Each gradient element is the sum of the same exact products as with the (channel,
// kh half) tiling of rounds 1-4, added in a different order (5e-7 relative: tools/probes/c1dw_probe.hip).  k loop 5.2 -> 4.1 us, launch
// 11.6-12.2 -> 9.6 us on the same box.  Workgroups stride over the batch and write one partial
// [64 * n_stack * 32 + 32] each (bias gradient = column sums of dY, fixed order), summed by k_reduce_adam / k_reduce_partials.
#pragma once
#include "conv1_bf16.hpp"

namespace bdr {

struct Conv1DwArgs {
    const uint8_t* x;     // [B][n_stack][84][84] u8
    const float* dy;      // [B*400][32] f32
    float* part;          // [gridDim.x][part_stride]
    size_t part_stride;   // floats (64*n_stack*32 + 32)
    int B;
};

constexpr int C1DW_LDM = 408;   // bf16 per plane row (400 + 8)
constexpr int C1DW_ROW = 88, C1DW_CH = 84 * 88 + 96;   // the image in LDS: rows padded 84 -> 88 bytes, frames 7 392 -> 7 488 (see k_conv1_dw_bf16)
#ifndef C1DW_WAVES_PER_EU
#define C1DW_WAVES_PER_EU 3   // register budget of a third wave per SIMD (<= 168 VGPRs): the kernel itself runs two, the rest of the register file
#endif                        // is what lets a wave of the OTHER queue's kernel share the SIMD (two-queue schedule of the DQN step)
constexpr int C1DW_PF = 2;      // k-steps of pixel fragments (LDS reads) in flight

struct __attribute__((packed, aligned(4))) U32x4A4 { uint32_t x, y, z, w; };   // dword-aligned 16-byte load

// bf16 pair from byte PH of d0 and d1 (integers 0..255: bf16 = upper half of the f32)
template <int PH>
__device__ __forceinline__ uint32_t u8pair_to_bf16(uint32_t d0, uint32_t d1)
{
    const uint32_t f0 = __float_as_uint((float)((d0 >> (8 * PH)) & 255u)), f1 = __float_as_uint((float)((d1 >> (8 * PH)) & 255u));
    return __builtin_amdgcn_perm(f1, f0, 0x07060302u);
}

#ifdef C1DW_TRACE   // tools/probes only
__device__ unsigned long long* g_c1dw_trace;
#define C1DW_TP(slot) do { if (threadIdx.x == 0 && g_c1dw_trace) g_c1dw_trace[blockIdx.x * 4 + (slot)] = wall_clock64(); } while (0)
#else
#define C1DW_TP(slot) do { } while (0)
#endif

// pixel fragments of k-step s of a tile: quads 2s, 2s + 1 (four ow each) of the lane half's 10 output rows, from the image in LDS
// (16 bytes at a 4-byte aligned address: two ds_read2_b32)
__device__ __forceinline__ void c1dw_load_step(const uint8_t* src, int s, int g, U32x4A4 (&d)[2])
{
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int quad = 2 * s + q;                                  // (of the lane's half of the image: src includes 10 output rows for g = 1)
        const int oh = quad / 5, ow0 = 4 * (quad - 5 * oh);
        const uint32_t* p = reinterpret_cast<const uint32_t*>(src + (4 * oh) * C1DW_ROW + 4 * ow0);
        d[q].x = p[0]; d[q].y = p[1]; d[q].z = p[2]; d[q].w = p[3];
    }
}

// the 25 k-steps of one image for the TPW tiles of a wave; PH = the wave's column phase (kw % 4).  The pixel fragments run C1DW_PF
// k-steps ahead of the MFMAs through a register ring; the caller has requested the first C1DW_PF steps.
template <int PH, int TPW>
__device__ __forceinline__ void c1dw_steps(const uint8_t* const (&src)[TPW], U32x4A4 (&ring)[TPW][C1DW_PF + 1][2], const uint16_t* planes, int i, int g,
                                           f32x16 (&acc)[TPW])
{
    // dY fragments one k-step ahead of their MFMAs as well
    uint4 bq[3], bn[3];
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) bq[pl] = *reinterpret_cast<const uint4*>(&planes[(pl * 32 + i) * C1DW_LDM + 200 * g]);
#pragma unroll
    for (int s = 0; s < 25; ++s) {
        if (s + 1 < 25) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bn[pl] = *reinterpret_cast<const uint4*>(&planes[(pl * 32 + i) * C1DW_LDM + 8 * (s + 1) + 200 * g]);
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            U32x4A4 (&cur)[2] = ring[t][s % (C1DW_PF + 1)];
            if (s + C1DW_PF < 25) c1dw_load_step(src[t], s + C1DW_PF, g, ring[t][(s + C1DW_PF) % (C1DW_PF + 1)]);
            uint4 av;
            av.x = u8pair_to_bf16<PH>(cur[0].x, cur[0].y);
            av.y = u8pair_to_bf16<PH>(cur[0].z, cur[0].w);
            av.z = u8pair_to_bf16<PH>(cur[1].x, cur[1].y);
            av.w = u8pair_to_bf16<PH>(cur[1].z, cur[1].w);
            const bf16x8 af = __builtin_bit_cast(bf16x8, av);
#pragma unroll
            for (int pl = 2; pl >= 0; --pl)   // small terms first
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bq[pl]), acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bq[pl] = bn[pl];
        __builtin_amdgcn_sched_barrier(0);
    }
}

// NS = n_stack.  The 4 * ceil(NS / 2) (channel pair, phase) row tiles of the image are dealt to the 8 waves: task = wave + 8 j ->
// phase = task % 4 (the same for every task of a wave), pair = task / 4; one tile each for NS = 3, 4 (the reference's examples use 4),
// TPW = 2 per wave for NS = 5 ... 8, idle waves (they still split dY and meet the barriers) below 3.  An odd NS leaves the second
// half of its last pair's tile unused (its lanes read channel NS - 1 again and store nothing).
template <int NS>
static __global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(NS <= 4 ? C1DW_WAVES_PER_EU : 2, NS <= 4 ? C1DW_WAVES_PER_EU : 2))) void k_conv1_dw_bf16(Conv1DwArgs a)
{
    C1DW_TP(0);
    constexpr int TASKS = 4 * ((NS + 1) / 2), TPW = (TASKS + 7) / 8;
    // LDS, dynamic (c1dw_lds_bytes): with a static size hipcc sees that only one workgroup fits a CU and spends the whole register file on
    // its two waves per SIMD (218 VGPRs) - registers that a wave of the OTHER queue's kernel needs to share the SIMD in the two-queue
    // schedule of the DQN step (same kernel, 218 instead of <= 168 VGPRs: the step was 4 us slower although the launch was 2.5 us faster)
    extern __shared__ __attribute__((aligned(16))) uint8_t c1dw_lds[];
    uint16_t* planes = reinterpret_cast<uint16_t*>(c1dw_lds);                                  // [3][32][C1DW_LDM] bf16, 78 336 B
    // the image, [c][84][88] u8: row stride 22 dwords and frame stride 1 872 = 16 (mod 32) dwords put the 2 channels x 8 rows x 2 columns
    // a half wave's ds_read2_b32 touches into 32 different banks (the batch's 21-dword rows would be 2-way conflicts: the k loop is LDS-bound)
    uint32_t* ximg_l = reinterpret_cast<uint32_t*>(c1dw_lds + 3 * 32 * C1DW_LDM * 2);          // NS * C1DW_CH bytes
    float (*sred)[32] = reinterpret_cast<float (*)[32]>(c1dw_lds + 3 * 32 * C1DW_LDM * 2 + NS * C1DW_CH);   // [16][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, g = lane >> 5;
    const int pn = tid & 31, pmg = tid >> 5;                      // prologue role: column n, quad group (16 groups)
    const int phase = wave & 3;

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;

    // per-lane A geometry: row i of a tile = (channel 2 pair + i / 16, kh = (i / 2) % 8, kw = phase + 4 (i % 2))
    int c_of[TPW], rowoff; bool live[TPW];
    rowoff = ((i >> 1) & 7) * C1DW_ROW + 4 * (i & 1) + g * (40 * C1DW_ROW);   // lane half g: output rows 10 g ... 10 g + 9
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int task = wave + 8 * t;
        live[t] = task < TASKS;
        const int c = 2 * ((live[t] ? task : 0) >> 2) + (i >> 4);
        c_of[t] = c < NS ? c : NS - 1;
    }

    for (int img = blockIdx.x; img < a.B; img += gridDim.x) {
        // the image -> LDS with coalesced 16-byte loads: the fragments of the k loop are strided gathers (2 channels x 8 rows per half
        // wave) - free in LDS, 4 x the cache lines per instruction of the old tiling when they came straight from memory
        constexpr int XPT = (NS * 441 + 511) / 512;
        uint4 xst[XPT];
        {
            const uint4* xg = reinterpret_cast<const uint4*>(a.x + (size_t)img * (NS * 7056));
#pragma unroll
            for (int k = 0; k < XPT; ++k) { const int t = tid + 512 * k; xst[k] = xg[t < NS * 441 ? t : NS * 441 - 1]; }
        }
        // ---- dY block of this image -> three k-major bf16 planes; bias partial sums
        {
            const float* dyb = a.dy + (size_t)img * 400 * 32;
            float s = 0.f;
            // a thread's quads in two batches (4 + 3): every load of a batch is in flight before its first value is split - two memory
            // round trips instead of seven, 16 live registers instead of 28
#pragma unroll
            for (int q0 = 0; q0 < 7; q0 += 4) {
                float y[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int quad = pmg + 16 * (q0 + q);
#pragma unroll
                    for (int j = 0; j < 4; ++j) y[q][j] = (q0 + q < 7 && quad < 100) ? dyb[(size_t)(4 * quad + j) * 32 + pn] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int quad = pmg + 16 * (q0 + q);
                    if (q0 + q < 7 && quad < 100) {
                        uint32_t hi[4], mid[4], lo[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            s += y[q][j];
                            hi[j] = __float_as_uint(y[q][j]) & 0xffff0000u;
                            const float r1 = y[q][j] - __uint_as_float(hi[j]);      // exact
                            mid[j] = __float_as_uint(r1) & 0xffff0000u;
                            const float r2 = r1 - __uint_as_float(mid[j]);          // exact, <= 8 significant bits
                            lo[j] = __float_as_uint(r2) & 0xffff0000u;
                        }
                        const int o = pn * C1DW_LDM + 4 * quad;
                        *reinterpret_cast<uint2*>(&planes[0 * 32 * C1DW_LDM + o]) = uint2{(hi[0] >> 16) | hi[1], (hi[2] >> 16) | hi[3]};
                        *reinterpret_cast<uint2*>(&planes[1 * 32 * C1DW_LDM + o]) = uint2{(mid[0] >> 16) | mid[1], (mid[2] >> 16) | mid[3]};
                        *reinterpret_cast<uint2*>(&planes[2 * 32 * C1DW_LDM + o]) = uint2{(lo[0] >> 16) | lo[1], (lo[2] >> 16) | lo[3]};
                    }
                }
            }
            sred[pmg][pn] = s;
        }
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int t = tid + 512 * k;
            if (t < NS * 441) {
                const uint32_t w[4] = {xst[k].x, xst[k].y, xst[k].z, xst[k].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int d = 4 * t + e, c = d / 1764, rem = d - c * 1764, r = rem / 21;   // dword d of the packed image -> (frame, row, column)
                    ximg_l[c * (C1DW_CH / 4) + r * (C1DW_ROW / 4) + (rem - r * 21)] = w[e];
                }
            }
        }
        __syncthreads();
        C1DW_TP(1);
        if (tid < 32) {
            float t = sred[0][tid];
#pragma unroll
            for (int k = 1; k < 16; ++k) t += sred[k][tid];
            bsum += t;
        }

        // ---- 25 k-steps of 16 positions, phase-specialised (wave-uniform branch)
        const uint8_t* src[TPW];
        U32x4A4 ring[TPW][C1DW_PF + 1][2];
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            src[t] = reinterpret_cast<const uint8_t*>(ximg_l) + c_of[t] * C1DW_CH + rowoff;
#pragma unroll
            for (int s0 = 0; s0 < C1DW_PF; ++s0) c1dw_load_step(src[t], s0, g, ring[t][s0]);
        }
        if (live[0]) {   // (a wave without a first task has none: tasks are dealt in order)
            if (phase == 0) c1dw_steps<0, TPW>(src, ring, planes, i, g, acc);
            else if (phase == 1) c1dw_steps<1, TPW>(src, ring, planes, i, g, acc);
            else if (phase == 2) c1dw_steps<2, TPW>(src, ring, planes, i, g, acc);
            else c1dw_steps<3, TPW>(src, ring, planes, i, g, acc);
        }
        C1DW_TP(2);
        __syncthreads();   // planes / sred are rewritten by the next image
    }

    float* part = a.part + (size_t)blockIdx.x * a.part_stride;
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        if (!live[t]) continue;   // (wave-uniform)
        const int pair = (wave + 8 * t) >> 2;
        // (sixteen dword stores per lane, each a full 128-byte row per half wave; the transposed accumulator - four 16-byte stores of
        // 32-byte row segments - was measured: 2.1 us instead of 1.6 for this epilogue)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * g;                  // row of the tile -> (channel, kh, kw) -> k = (c, kh, kw)
            const int c = 2 * pair + (row >> 4), kh = (row >> 1) & 7, kw = phase + 4 * (row & 1);
            if (c < NS) part[(size_t)(c * 64 + kh * 8 + kw) * 32 + i] = acc[t][r];
        }
    }
    if (tid < 32) part[64 * NS * 32 + tid] = bsum;
    C1DW_TP(3);
}

constexpr size_t c1dw_lds_bytes(int ns) { return (size_t)3 * 32 * C1DW_LDM * 2 + (size_t)ns * C1DW_CH + 16 * 32 * 4; }   // dY planes + image + bias partials

inline hipError_t launch_conv1_dw_bf16(int ns, dim3 grid, hipStream_t st, const Conv1DwArgs& d)
{
    switch (ns) {
#define BDR_C1DW_CASE(N) case N: { \
            static bool allowed[64] = {false};   /* per device: a process may drive several GPUs (the attribute belongs to the current one) */ \
            int dev = 0; \
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice; \
            if (!allowed[dev]) { \
                const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv1_dw_bf16<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c1dw_lds_bytes(N)); \
                if (e != hipSuccess) return e; \
                allowed[dev] = true; \
            } \
            hipLaunchKernelGGL(k_conv1_dw_bf16<N>, grid, dim3(512), c1dw_lds_bytes(N), st, d); break; }
        BDR_C1DW_CASE(1) BDR_C1DW_CASE(2) BDR_C1DW_CASE(3) BDR_C1DW_CASE(4) BDR_C1DW_CASE(5) BDR_C1DW_CASE(6) BDR_C1DW_CASE(7) BDR_C1DW_CASE(8)
#undef BDR_C1DW_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace bdr
