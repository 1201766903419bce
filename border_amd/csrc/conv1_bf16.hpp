// conv1 forward on the bf16 matrix cores with EXACT operands.
//
// conv1's input is u8 pixels 0..255: exactly representable in bf16 (8-bit significand).  Each f32
// weight is split into three bf16 terms w = hi + mid + lo (truncation split, exact: 3 x 8 bits cover
// the 24-bit significand), so every product x*hi, x*mid, x*lo is exact in f32 and the MFMA only
// rounds in its f32 accumulation -- the same error class as the FP32 MFMA / an fmaf chain, at
// 3 bf16 MFMAs (32 cycles each, K=16) instead of 8 f32 MFMAs (64 cycles each, K=2) per 16 k:
// 5.3x fewer matrix-pipe cycles.  cnn/base.rs:26-28 (x/255 -> conv(4->32,k8,s4) -> relu); the 1/255
// is applied to the f32 accumulator in the epilogue.
//
// v_mfma_f32_32x32x16_bf16 operand maps: A lane l: A[i=l&31][k=8*(l>>5)+0..7], B lane l:
// B[k=8*(l>>5)+0..7][j=l&31], D as the f32 form.  k-step s covers channel c=s/4, patch rows
// kh=(s%4)*2+h: a lane's 8 k-values are the 8 contiguous pixels of one patch row = ONE 8-byte
// load straight from the NCHW u8 batch, no LDS staging for A.  The three weight planes live in
// LDS for the lifetime of a (persistent) workgroup in exactly the B-fragment order, so a lane
// fetches a fragment with one conflict-free ds_read_b128.  No barrier inside the item loop.
#pragma once
#include <hip/hip_runtime.h>

#include "igemm.hpp"

namespace bdr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int C1_MAX_STACK = 8;   // AtariCnnConfig::n_stack values the conv1 kernels are instantiated for: 1 ... 8 (cnn/config.rs:14-24; 4 in every example)
constexpr int c1_plane_vecs(int ns) { return 4 * ns * 2 * 32; }   // uint4 per plane: [s][h][n] x 8 bf16, s = 4 * n_stack k-steps of 16

struct Conv1Args {
    const uint8_t* x[3];     // [B][n_stack][84][84] u8
    const float* w1[3];      // [64 * n_stack][32] f32, k=(c,kh,kw)
    const float* bias[3];
    float* out[3];           // [M][32] f32 (NHWC)
    int M;                   // B*400
    int nz;
};

__device__ __forceinline__ bf16x8 u8x8_to_bf16(uint32_t lo, uint32_t hi)
{
    // integers 0..255 are exact in bf16: the bf16 pattern is the upper half of the f32 pattern
    const uint32_t f0 = __float_as_uint((float)(lo & 255u)), f1 = __float_as_uint((float)((lo >> 8) & 255u));
    const uint32_t f2 = __float_as_uint((float)((lo >> 16) & 255u)), f3 = __float_as_uint((float)(lo >> 24));
    const uint32_t f4 = __float_as_uint((float)(hi & 255u)), f5 = __float_as_uint((float)((hi >> 8) & 255u));
    const uint32_t f6 = __float_as_uint((float)((hi >> 16) & 255u)), f7 = __float_as_uint((float)(hi >> 24));
    uint4 v;
    v.x = __builtin_amdgcn_perm(f1, f0, 0x07060302u);
    v.y = __builtin_amdgcn_perm(f3, f2, 0x07060302u);
    v.z = __builtin_amdgcn_perm(f5, f4, 0x07060302u);
    v.w = __builtin_amdgcn_perm(f7, f6, 0x07060302u);
    return __builtin_bit_cast(bf16x8, v);
}

// grid: nz * G workgroups of 512 threads (8 waves, 2 workgroups per CU); workgroup b serves
// instance b % nz.  Prologue: split the instance's f32 weights into the three bf16 planes directly
// into LDS (each thread 2 fragments of 8 k).  Then every wave walks 32-pixel items with stride G*8;
// the next item's pixels are prefetched into registers, no barrier inside the item loop.
// NS = n_stack (input channels): K = 64 * NS, 4 * NS k-steps of 16.
template <int NS>
static __global__ __launch_bounds__(512, NS <= 4 ? 4 : 2) void k_conv1_bf16(Conv1Args a)
{
    constexpr int C1_PLANE_VECS = c1_plane_vecs(NS), KS = 4 * NS;
    __shared__ uint4 wl[3 * C1_PLANE_VECS];   // 12 KiB x n_stack: three bf16 weight planes in B-fragment order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.x % a.nz, wg = blockIdx.x / a.nz, nwg = gridDim.x / a.nz;
    const int i = lane & 31, h = lane >> 5;
    const int items = (a.M + 31) / 32, stride = nwg * 8;
    const uint8_t* x = a.x[z];

    // gather of one item: 16 patch rows of 8 pixels per lane (lane = pixel i, row parity h)
    auto load_item = [&](int item, uint2 (&r)[KS]) {
        int m = item * 32 + i;
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

    // first item's pixels are in flight while the weights are split
    uint2 nxt[KS];
    int item = wg * 8 + wave;
    if (item < items) load_item(item, nxt);
    {
        const float* w1 = a.w1[z];
        for (int t = tid; t < C1_PLANE_VECS; t += 512) {
            const int n = t & 31, hh = (t >> 5) & 1, s = t >> 6;
            uint32_t hi[8], mid[8], lo[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float w = w1[(size_t)(16 * s + 8 * hh + j) * 32 + n];
                hi[j] = __float_as_uint(w) & 0xffff0000u;
                const float r1 = w - __uint_as_float(hi[j]);            // exact
                mid[j] = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(mid[j]);          // exact, <= 8 significant bits
                lo[j] = __float_as_uint(r2) & 0xffff0000u;
            }
            wl[0 * C1_PLANE_VECS + t] = uint4{(hi[0] >> 16) | hi[1], (hi[2] >> 16) | hi[3], (hi[4] >> 16) | hi[5], (hi[6] >> 16) | hi[7]};
            wl[1 * C1_PLANE_VECS + t] = uint4{(mid[0] >> 16) | mid[1], (mid[2] >> 16) | mid[3], (mid[4] >> 16) | mid[5], (mid[6] >> 16) | mid[7]};
            wl[2 * C1_PLANE_VECS + t] = uint4{(lo[0] >> 16) | lo[1], (lo[2] >> 16) | lo[3], (lo[4] >> 16) | lo[5], (lo[6] >> 16) | lo[7]};
        }
    }
    __syncthreads();

    const float bias = a.bias[z][i];
    float* out = a.out[z];
    asm volatile("" ::"v"(bias));   // land the bias load here: otherwise the epilogue's first store waits vmcnt(0), i.e. for the NEXT item's pixel loads

    for (; item < items; item += stride) {
        uint2 cur[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) cur[s] = nxt[s];
        if (item + stride < items) load_item(item + stride, nxt);   // next item in flight during the MFMAs
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // B fragments one k-step ahead of the MFMAs that use them; sched_barrier keeps hipcc from
        // hoisting all 48 fragment reads (192 VGPRs) to the top of the item
        uint4 bq[3], bn[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) bq[pl] = wl[pl * C1_PLANE_VECS + h * 32 + i];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bn[pl] = wl[pl * C1_PLANE_VECS + ((s + 1) * 2 + h) * 32 + i];
            }
            const bf16x8 av = u8x8_to_bf16(cur[s].x, cur[s].y);
#pragma unroll
            for (int pl = 2; pl >= 0; --pl)   // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, bq[pl]), acc, 0, 0, 0);
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) bq[pl] = bn[pl];
            __builtin_amdgcn_sched_barrier(0);
        }
        const int m0 = item * 32;
        // Full items (all of them when M % 32 == 0) store without per-row predicates.  With 16 predicated stores the compiler
        // puts s_waitcnt vmcnt(0) into every predicate - a wait for the next item's pixel loads AND for the previous store
        // (vmcnt counts stores): 16 memory round trips in a row per item.
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
    }
}

// the instantiation for a run-time n_stack
inline hipError_t launch_conv1_bf16(int ns, dim3 grid, hipStream_t st, const Conv1Args& c)
{
    switch (ns) {
#define BDR_C1_CASE(N) case N: hipLaunchKernelGGL(k_conv1_bf16<N>, grid, dim3(512), 0, st, c); break;
        BDR_C1_CASE(1) BDR_C1_CASE(2) BDR_C1_CASE(3) BDR_C1_CASE(4) BDR_C1_CASE(5) BDR_C1_CASE(6) BDR_C1_CASE(7) BDR_C1_CASE(8)
#undef BDR_C1_CASE
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace bdr
