// A dense layer with a SHORT reduction (Kp = 64) over many rows on the bf16 matrix cores with split operands - IQN's cosine-embedding
// layer phi = relu(cos[B * N][64] x W[64][F] + b) (iqn/model/base.rs), 13 GFLOP per network at C4 whose result ([32 768][3 136] f32,
// 411 MB) is what the launch should cost.  The generic kernels pay a three-tile pipeline prologue for two k-tiles of work per
// workgroup (FP32 k_igemm: 172 us per network; k_igemm_b3: 225 us).  Here a workgroup keeps its 128 input rows resident - split ONCE
// into three bf16 planes in LDS - and walks all column tiles of the layer: per 128-column tile one 48 KB copy of the weight planes
// (double-buffered), 96 MFMAs per wave (six of the nine exact partial products, as igemm_b3.hpp), bias + ReLU + store.
// grid: ceil(M / 128) workgroups of 256 threads (C4: 256 = one per CU), 144 KB LDS.  gfx950 only.
#pragma once
#include "igemm_b3.hpp"

namespace bdr {

struct DenseK64Args {
    const float* x; int ldx;        // [M][64] f32 input rows (ldx >= 64)
    const uint16_t* wpl;            // [3][Np][64] bf16 planes of W^T (dense_split_planes' transposed planes)
    const float* bias;              // [Np]
    float* out; int ldo;            // [M][Np]
    int M, Np, relu;
};

// plane tiles [128 rows][64 k] bf16: 128-byte rows, the eight 16-byte chunks of row r stored at chunk ^ ((r >> 1) & 7): a fragment read
// (16 lanes = 16 rows, one chunk index) covers all 64 banks once, the staging writes are contiguous runs
__device__ __forceinline__ int k64_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 1) & 7)) << 3); }

static __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void k_dense_k64_b3(DenseK64Args a)
{
    constexpr int PLANE = 128 * 64;   // u16
    __shared__ __attribute__((aligned(16))) uint16_t As[3 * PLANE];       // 48 KB
    __shared__ __attribute__((aligned(16))) uint16_t Ws[2][3 * PLANE];    // 96 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, j = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * 128;
    const int NT = (a.Np + 127) / 128;

    u32x4_t rw[4][3];
    auto prefetch_w = [&](int t) {   // column tile t (clamped): 128 rows x 8 chunks x 3 planes, 12 x 16 B per thread
        const int n0 = min(t, NT - 1) * 128;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int e = tid + p * 256, n = min(n0 + (e >> 3), a.Np - 1);
            const uint16_t* src = a.wpl + (size_t)n * 64 + (e & 7) * 8;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) rw[p][pl] = *reinterpret_cast<const u32x4_t*>(src + (size_t)pl * a.Np * 64);
        }
    };
    auto commit_w = [&](int stage, int p) {
        const int e = tid + p * 256, o = k64_off(e >> 3, e & 7);
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4_t*>(&Ws[stage][pl * PLANE + o]) = rw[p][pl];
    };

    prefetch_w(0);
    // the workgroup's 128 input rows -> three bf16 planes in LDS (rows beyond M: zeros)
    f32x4 xin[8];   // (all eight loads first: behind a row test each was a round trip of its own)
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int e = tid + p * 256, row = e >> 4, kq = e & 15;
        xin[p] = *reinterpret_cast<const f32x4*>(a.x + (size_t)min(m0 + row, a.M - 1) * a.ldx + kq * 4);
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) {
        const int e = tid + p * 256, row = e >> 4, kq = e & 15;
        const f32x4 v = m0 + row < a.M ? xin[p] : f32x4{0.f, 0.f, 0.f, 0.f};
        u32x2_t sp[3];
        split3_f32x4(v, sp);
        const int o = k64_off(row, kq >> 1) + (kq & 1) * 4;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x2_t*>(&As[pl * PLANE + o]) = sp[pl];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) commit_w(0, p);
    prefetch_w(1);
    __syncthreads();

    // partial products, smallest first: (a plane, w plane); the three below 2^-24 |a||w| are dropped
    constexpr int ORD6[6][2] = {{2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
    int cur = 0;
    for (int t = 0; t < NT; ++t) {
        const int n0 = t * 128;
        f32x16 acc[2][2];
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;
        float bias[2];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) bias[tn] = a.bias[min(n0 + (wn * 2 + tn) * 32 + j, a.Np - 1)];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            bf16x8_t fa[2][3], fb[2][3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int tm = 0; tm < 2; ++tm) fa[tm][pl] = *reinterpret_cast<const bf16x8_t*>(&As[pl * PLANE + k64_off((wm * 2 + tm) * 32 + j, s * 2 + h)]);
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) fb[tn][pl] = *reinterpret_cast<const bf16x8_t*>(&Ws[cur][pl * PLANE + k64_off((wn * 2 + tn) * 32 + j, s * 2 + h)]);
            }
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][ORD6[q][0]], fb[tn][ORD6[q][1]], acc[tm][tn], 0, 0, 0);
            // the next column tile's weight planes ride on this tile's MFMAs: registers -> idle stage (every wave left that stage at the
            // barrier behind tile t - 1) during the first two k-steps, then the loads of tile t + 2 into the freed registers
            if (s == 0) { commit_w(cur ^ 1, 0); commit_w(cur ^ 1, 1); }
            else if (s == 1) { commit_w(cur ^ 1, 2); commit_w(cur ^ 1, 3); }
            else if (s == 2) prefetch_w(t + 2);
        }
        // values first, then the 64 stores back to back (a bias used behind a store costs a wait on that store: stores share vmcnt)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[tm][tn][r] + bias[tn];
                    if (a.relu) v = v > 0.f ? v : 0.f;
                    acc[tm][tn][r] = v;
                }
        if (m0 + 128 <= a.M && n0 + 128 <= a.Np) {   // full tile (workgroup-uniform): no per-element predicates, one base pointer per block
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    float* o = a.out + (size_t)(m0 + (wm * 2 + tm) * 32 + 4 * h) * a.ldo + n0 + (wn * 2 + tn) * 32 + j;
#pragma unroll
                    for (int r = 0; r < 16; ++r) __builtin_nontemporal_store(acc[tm][tn][r], o + (size_t)((r & 3) + 8 * (r >> 2)) * a.ldo);   // (a 411 MB result at C4: streamed)
                }
        } else {
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn) {
                    const int n = n0 + (wn * 2 + tn) * 32 + j;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + (wm * 2 + tm) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (m < a.M && n < a.Np) __builtin_nontemporal_store(acc[tm][tn][r], &a.out[(size_t)m * a.ldo + n]);
                    }
                }
        }
        __syncthreads();
        cur ^= 1;
    }
}

}  // namespace bdr
