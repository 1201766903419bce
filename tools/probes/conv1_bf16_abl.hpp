// Copy of the product's k_conv1_bf16 (border_amd/csrc/conv1_bf16.hpp) with the timing ablations of tools/probes/conv1_probe.hip:
//   -DC1_NOLOAD no pixel loads, -DC1_PRESPLIT weight planes copied from a pre-split buffer, -DC1_NOMFMA no matrix instructions,
//   -DC1_NOSTORE no output stores  (results are WRONG with any of them).  Namespace bdr_abl; the product header carries none of this.
// Snapshot of the round-3/4 kernel body - re-copy when the product kernel changes.
#pragma once
#include "../../border_amd/csrc/conv1_bf16.hpp"

namespace bdr_abl {
using namespace bdr;

// grid: nz * G workgroups of 512 threads (8 waves, 2 workgroups per CU); workgroup b serves
// instance b % nz.  Prologue: split the instance's f32 weights into the three bf16 planes directly
// into LDS (each thread 2 fragments of 8 k).  Then every wave walks 32-pixel items with stride G*8;
// the next item's pixels are prefetched into registers, no barrier inside the item loop.
static __global__ __launch_bounds__(512, 4) void k_conv1_bf16(Conv1Args a)
{
    __shared__ uint4 wl[3 * C1_PLANE_VECS];   // 48 KiB: three bf16 weight planes in B-fragment order
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int z = blockIdx.x % a.nz, wg = blockIdx.x / a.nz, nwg = gridDim.x / a.nz;
    const int i = lane & 31, h = lane >> 5;
    const int items = (a.M + 31) / 32, stride = nwg * 8;
    const uint8_t* x = a.x[z];

    // gather of one item: 16 patch rows of 8 pixels per lane (lane = pixel i, row parity h)
    auto load_item = [&](int item, uint2 (&r)[16]) {
        int m = item * 32 + i;
        m = m < a.M ? m : a.M - 1;
        const int b = m / 400, rem = m - b * 400;
        const int oh = rem / 20, ow = rem - oh * 20;
        const uint8_t* p = x + (size_t)b * 28224 + (oh * 4 + h) * 84 + ow * 4;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(p + (s >> 2) * 7056 + ((s & 3) * 2) * 84);
#ifdef C1_NOLOAD
            r[s] = uint2{(uint32_t)(size_t)q, (uint32_t)lane};
#else
            r[s] = uint2{q[0], q[1]};
#endif
        }
    };

    // first item's pixels are in flight while the weights are split
    uint2 nxt[16];
    int item = wg * 8 + wave;
    if (item < items) load_item(item, nxt);
#ifdef C1_PRESPLIT   // tools/probes only (timing): the three planes copied as they are from a pre-split buffer
    {
        const uint4* wp = reinterpret_cast<const uint4*>(a.w1[z]);
        for (int t = tid; t < 3 * C1_PLANE_VECS; t += 512) wl[t] = wp[t % 2048];
    }
#else
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
#endif
    __syncthreads();

    const float bias = a.bias[z][i];
    float* out = a.out[z];
    asm volatile("" ::"v"(bias));   // land the bias load here: otherwise the epilogue's first store waits vmcnt(0), i.e. for the NEXT item's pixel loads

    for (; item < items; item += stride) {
        uint2 cur[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) cur[s] = nxt[s];
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
        for (int s = 0; s < 16; ++s) {
            if (s + 1 < 16) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) bn[pl] = wl[pl * C1_PLANE_VECS + ((s + 1) * 2 + h) * 32 + i];
            }
            const bf16x8 av = u8x8_to_bf16(cur[s].x, cur[s].y);
#pragma unroll
            for (int pl = 2; pl >= 0; --pl)   // small terms first
#ifdef C1_NOMFMA
                acc[pl] += (float)av[0] * (float)__builtin_bit_cast(bf16x8, bq[pl])[1];
#else
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, __builtin_bit_cast(bf16x8, bq[pl]), acc, 0, 0, 0);
#endif
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
#ifdef C1_NOSTORE
                if (v == 123.456f) out[(size_t)mo * 32 + i] = v;
#else
                out[(size_t)mo * 32 + i] = v > 0.f ? v : 0.f;
#endif
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

}  // namespace bdr_abl
