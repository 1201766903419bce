// Device-side restatement of border-atari-env's frame preprocessing:
//   border-atari-env/src/env.rs  skip_and_max :126-157 (max of the last two RGB frames), warp_and_grayscale :171-195
//   (image::imageops::resize(.., 84, 84, Triangle) + the reference's luma), stack_frame :197-209, reset :263-296.
// The resize belongs to the third-party crate image 0.23.14 (Cargo.toml:50), absent from /root/reference; the algorithm
// restated here (vertical_sample then horizontal_sample over a u8 intermediate, f32 weights and sums in source order,
// round-half-away) is documented in oracle/atari_prep.py, which this kernel matches bit for bit.  PARITY UNPINNED against
// the real crate: no vector exists offline.
//
// One workgroup per (environment, step).  HBM-bound byte work: 2 x 100.8 KB of RGB in, 7 KB of luma out, the three older
// frames of the stack shifted in place.  Pass 1 (rows 210 -> 84, 4 bytes per thread and item,
// all loads of an item in flight together) writes a [84][160][3] u8 intermediate to LDS (40 KB), pass 2
// (columns 160 -> 84) and the luma read it back.  FMA contraction is switched off for this file: the reference multiplies and adds
// separately (Rust never contracts).
#include <chrono>
#include "common.hpp"

// hipcc contracts a * b + c into an FMA by default (also through HIP's __fmul_rn / __fadd_rn, which are plain operators
// compiled with that default): switched off in every function of this file that does arithmetic, which therefore uses
// plain operators - the reference rounds the product and the sum separately
#define NO_FMA _Pragma("clang fp contract(off)")

using namespace bdr;

struct bdr_atari_prep {
    int32_t device = 0;
    uint32_t n_envs = 0, width = 0, height = 0;
    hipStream_t stream = nullptr;
    uint8_t* stacks = nullptr;      // [n_envs][4][84][84]
    uint8_t* prev = nullptr;        // [n_envs][4][84][84]: every environment's stack as it was BEFORE its last step (obs_t of the
                                    // transition whose next_obs is `stacks`) - what a device-side push needs (bdr_replay_push_device)
    uint8_t* d_frames = nullptr;    // staging: [cap][2][H][W][3]
    uint8_t* h_stage = nullptr;     // pinned
    uint32_t cap = 0;
};

namespace {
constexpr int OUT = 84;
constexpr int MAX_W = 160, MAX_H = 256;

struct Taps { int left, n; float w[8]; float sum; };

// IEEE-correctly-rounded f32 division whatever the compiler's fast-division setting: a double quotient rounded once more
// to f32 equals the correctly rounded f32 quotient (53 >= 2 * 24 + 2)
__device__ inline float div_rn(float a, float b) { return (float)((double)a / (double)b); }

// sample.rs: the taps of output coordinate o when resampling n_in -> OUT with the Triangle filter
__device__ inline Taps taps_of(int o, int n_in)
{
    NO_FMA
    Taps t;
    const float ratio = div_rn((float)n_in, (float)OUT);
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float support = (1.0f * sratio);
    float centre = ((float)o + 0.5f) * ratio;
    int left = (int)floorf((centre - support));
    left = min(max(left, 0), n_in - 1);
    int right = (int)ceilf((centre + support));
    right = min(max(right, left + 1), n_in);
    centre = (centre - 0.5f);
    t.left = left; t.n = min(right - left, 8); t.sum = 0.0f;
    for (int k = 0; k < 8; ++k) {
        float w = 0.0f;
        if (k < t.n) {
            const float x = div_rn(((float)(left + k) - centre), sratio);
            const float ax = fabsf(x);
            w = ax < 1.0f ? (1.0f - ax) : 0.0f;
            t.sum = (t.sum + w);
        }
        t.w[k] = w;
    }
    return t;
}

__device__ inline uint8_t finish(float t, float sum)
{
    NO_FMA
    t = div_rn(t, sum);
    t = fminf(fmaxf(t, 0.0f), 255.0f);
    return (uint8_t)roundf(t);   // FloatNearest: half away from zero
}

struct PrepArgs {
    const uint8_t* frames;   // [n][2][H][W][3] (reset: both copies equal)
    const uint32_t* env_ixs; // [n]
    uint8_t* stacks;
    uint8_t* prev;
    int W, H;
    int reset;               // 1: all four slots <- the new frame
};

__device__ inline uint32_t max_u8x4(uint32_t x, uint32_t y)
{
    uint32_t r = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) r |= max((x >> (8 * b)) & 255u, (y >> (8 * b)) & 255u) << (8 * b);
    return r;
}

// PREP_BANDS workgroups of 512 threads per environment: a workgroup owns a band of 84 / PREP_BANDS output rows through both passes (the
// vertical pass of a band needs nothing of the others), so one environment - the one-env online loop - is spread over six CUs instead of
// being one workgroup's 57 us latency chain.  The tap tables of both passes are built once per workgroup in LDS; pass 1 works on
// 4-byte groups of a row (W * 3 is a multiple of 4 for even W; odd widths take the byte path) with all of an item's loads
// issued before the arithmetic.
constexpr int PREP_THREADS = 512, PREP_BANDS = 6, BAND_ROWS = OUT / PREP_BANDS;
static_assert(OUT % PREP_BANDS == 0, "bands of whole output rows");
__global__ __launch_bounds__(PREP_THREADS) void k_atari_prep(PrepArgs a)
{
    NO_FMA
    extern __shared__ uint8_t tmp[];   // [BAND_ROWS][W][3] (rounded up to 16 B), then the two tap tables
    const int e = blockIdx.x / PREP_BANDS, y0 = (blockIdx.x % PREP_BANDS) * BAND_ROWS, tid = threadIdx.x;
    const int rowb = a.W * 3;
    Taps* vt = reinterpret_cast<Taps*>(tmp + ((BAND_ROWS * rowb + 15) & ~15));
    Taps* ht = vt + OUT;
    const size_t fsz = (size_t)a.H * rowb;
    const uint8_t* fa = a.frames + (size_t)e * 2 * fsz;
    const uint8_t* fb = fa + fsz;
    uint8_t* stack = a.stacks + (size_t)a.env_ixs[e] * 4 * OUT * OUT;
    uint8_t* prev = a.prev + (size_t)a.env_ixs[e] * 4 * OUT * OUT;

    if (tid < OUT) vt[tid] = taps_of(tid, a.H);
    else if (tid < 2 * OUT) ht[tid - OUT] = taps_of(tid - OUT, a.W);
    __syncthreads();

    // pass 1: vertical_sample of max(frame_a, frame_b) (env.rs:148-152)
    if ((rowb & 3) == 0 && (fsz & 3) == 0) {
        const int roww = rowb >> 2;
        const uint32_t* wa = reinterpret_cast<const uint32_t*>(fa);
        const uint32_t* wb = reinterpret_cast<const uint32_t*>(fb);
        uint32_t* tw = reinterpret_cast<uint32_t*>(tmp);
        for (int idx = tid; idx < BAND_ROWS * roww; idx += PREP_THREADS) {
            const int oy = idx / roww, xw = idx - oy * roww;
            const Taps& t = vt[y0 + oy];
            uint32_t px[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int row = min(t.left + k, a.H - 1);                 // k >= n: weight 0, row clamped
                px[k] = max_u8x4(wa[(size_t)row * roww + xw], wb[(size_t)row * roww + xw]);
            }
            float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int k = 0; k < t.n; ++k)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[b] = acc[b] + (float)((px[k] >> (8 * b)) & 255u) * t.w[k];
            uint32_t o = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) o |= (uint32_t)finish(acc[b], t.sum) << (8 * b);
            tw[idx] = o;
        }
    } else {
        for (int idx = tid; idx < BAND_ROWS * rowb; idx += PREP_THREADS) {
            const int oy = idx / rowb, xb = idx - oy * rowb;
            const Taps& t = vt[y0 + oy];
            float acc = 0.0f;
            for (int k = 0; k < t.n; ++k) {
                const size_t off = (size_t)(t.left + k) * rowb + xb;
                acc = acc + (float)max(fa[off], fb[off]) * t.w[k];
            }
            tmp[idx] = finish(acc, t.sum);
        }
    }
    __syncthreads();

    // stack_frame (env.rs:197-209): slots 1..3 <- slots 0..2, every thread moves its own pixels (oldest first)
    if (!a.reset) {
        for (int p = y0 * OUT + tid; p < (y0 + BAND_ROWS) * OUT; p += PREP_THREADS) {
            const uint8_t s0 = stack[p], s1 = stack[1 * OUT * OUT + p], s2 = stack[2 * OUT * OUT + p], s3 = stack[3 * OUT * OUT + p];
            prev[p] = s0; prev[1 * OUT * OUT + p] = s1; prev[2 * OUT * OUT + p] = s2; prev[3 * OUT * OUT + p] = s3;   // obs_t, kept for the push
            stack[3 * OUT * OUT + p] = s2;
            stack[2 * OUT * OUT + p] = s1;
            stack[1 * OUT * OUT + p] = s0;
        }
    }
    // pass 2: horizontal_sample + luma; one thread per output pixel (the same pixels it just moved)
    for (int p = y0 * OUT + tid; p < (y0 + BAND_ROWS) * OUT; p += PREP_THREADS) {
        const int oy = p / OUT, ox = p - oy * OUT;
        const Taps& t = ht[ox];
        float c[3] = {0.0f, 0.0f, 0.0f};
        for (int k = 0; k < t.n; ++k) {
            const uint8_t* px = tmp + (size_t)(oy - y0) * rowb + (t.left + k) * 3;
            for (int ch = 0; ch < 3; ++ch) c[ch] = c[ch] + (float)px[ch] * t.w[k];
        }
        const float c0 = (float)finish(c[0], t.sum), c1 = (float)finish(c[1], t.sum), c2 = (float)finish(c[2], t.sum);
        // env.rs:183-185: ((0.299 * r) + (0.587 * g) + (0.114 * b)) as u8 with (b, g, r) = channels (0, 1, 2)
        const float g = ((0.299f * c2) + (0.587f * c1)) + (0.114f * c0);
        const uint8_t v = (uint8_t)fminf(floorf(g), 255.0f);
        stack[p] = v;
        if (a.reset) { stack[OUT * OUT + p] = v; stack[2 * OUT * OUT + p] = v; stack[3 * OUT * OUT + p] = v; }   // env.rs:286-292
    }
}

int32_t ensure_cap(bdr_atari_prep* h, uint32_t n)
{
    if (n <= h->cap) return BDR_OK;
    BDR_HIP(hipStreamSynchronize(h->stream));
    (void)hipFree(h->d_frames); (void)hipHostFree(h->h_stage);
    h->d_frames = nullptr; h->h_stage = nullptr; h->cap = 0;
    const size_t fsz = (size_t)h->width * h->height * 3;
    // frames and environment indices travel in ONE pinned staging buffer and one copy: [n][2][H][W][3] u8, padded to 16 bytes, then [n] u32
    const size_t ixs_off = ((size_t)n * 2 * fsz + 15) & ~(size_t)15;
    BDR_HIP(hipMalloc((void**)&h->d_frames, ixs_off + (size_t)n * sizeof(uint32_t)));
    BDR_HIP(hipHostMalloc((void**)&h->h_stage, ixs_off + (size_t)n * sizeof(uint32_t), hipHostMallocDefault));
    h->cap = n;
    return BDR_OK;
}

int32_t run(bdr_atari_prep* h, uint32_t n, const uint32_t* env_ixs, const uint8_t* fa, const uint8_t* fb, int reset)
{
    BDR_REQUIRE(h && env_ixs && fa && fb, "null argument");
    if (n == 0) return BDR_OK;
    for (uint32_t k = 0; k < n; ++k) {
        BDR_REQUIRE(env_ixs[k] < h->n_envs, "environment index out of range");
        for (uint32_t j = 0; j < k; ++j) BDR_REQUIRE(env_ixs[j] != env_ixs[k], "an environment appears twice in one call");
    }
    BDR_HIP(hipSetDevice(h->device));
    BDR_TRY(ensure_cap(h, n));
    // (the staging buffer is free: the previous call returned after its kernel had finished)
    const size_t fsz = (size_t)h->width * h->height * 3;
    for (uint32_t k = 0; k < n; ++k) {
        memcpy(h->h_stage + (size_t)k * 2 * fsz, fa + (size_t)k * fsz, fsz);
        memcpy(h->h_stage + (size_t)k * 2 * fsz + fsz, fb + (size_t)k * fsz, fsz);
    }
    const size_t ixs_off = ((size_t)h->cap * 2 * fsz + 15) & ~(size_t)15;   // (the layout of ensure_cap: indices behind the capacity's frames)
    memcpy(h->h_stage + ixs_off, env_ixs, (size_t)n * sizeof(uint32_t));
    if (n == h->cap) BDR_HIP(hipMemcpyAsync(h->d_frames, h->h_stage, ixs_off + (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    else {
        BDR_HIP(hipMemcpyAsync(h->d_frames, h->h_stage, (size_t)n * 2 * fsz, hipMemcpyHostToDevice, h->stream));
        BDR_HIP(hipMemcpyAsync(h->d_frames + ixs_off, h->h_stage + ixs_off, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    }
    PrepArgs a{h->d_frames, reinterpret_cast<const uint32_t*>(h->d_frames + ixs_off), h->stacks, h->prev, (int)h->width, (int)h->height, reset};
    hipLaunchKernelGGL(k_atari_prep, dim3(n * PREP_BANDS), dim3(PREP_THREADS), (((size_t)BAND_ROWS * h->width * 3 + 15) & ~(size_t)15) + 2 * OUT * sizeof(Taps), h->stream, a);
    BDR_HIP(hipGetLastError());
    // The kernel is COMPLETE when the call returns (its end-of-kernel release included): the agent and the replay buffer read the stacks
    // on their own streams without an event.  (A completion word in pinned memory, as the acting calls use for their results, would return
    // ~10 us earlier - but before that release; the stacks' readers are kernels, not the host.)
    BDR_HIP(hipStreamSynchronize(h->stream));
    return BDR_OK;
}
}  // namespace

extern "C" {

int32_t bdr_atari_prep_create(int32_t device, uint32_t n_envs, uint32_t width, uint32_t height, bdr_atari_prep** out)
{
    BDR_REQUIRE(out, "null argument");
    BDR_REQUIRE(n_envs >= 1, "n_envs must be >= 1");
    BDR_REQUIRE(width >= 1 && width <= (uint32_t)MAX_W && height >= 1 && height <= (uint32_t)MAX_H, "frame size out of range (<= 160 x 256)");
    BDR_TRY(ensure_device(device));
    bdr_atari_prep* h = new bdr_atari_prep();
    h->device = device; h->n_envs = n_envs; h->width = width; h->height = height;
    hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc((void**)&h->stacks, (size_t)n_envs * 4 * OUT * OUT);
    if (e == hipSuccess) e = hipMemsetAsync(h->stacks, 0, (size_t)n_envs * 4 * OUT * OUT, h->stream);   // frames: vec![0; 4*84*84]
    if (e == hipSuccess) e = hipMalloc((void**)&h->prev, (size_t)n_envs * 4 * OUT * OUT);
    if (e == hipSuccess) e = hipMemsetAsync(h->prev, 0, (size_t)n_envs * 4 * OUT * OUT, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        if (h->stacks) (void)hipFree(h->stacks);
        if (h->prev) (void)hipFree(h->prev);
        if (h->stream) (void)hipStreamDestroy(h->stream);
        delete h;
        return fail(BDR_ERR_HIP, "atari_prep allocation failed: %s", hipGetErrorString(e));
    }
    *out = h;
    return BDR_OK;
}

int32_t bdr_atari_prep_destroy(bdr_atari_prep* h)
{
    if (!h) return BDR_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    (void)hipFree(h->stacks); (void)hipFree(h->prev); (void)hipFree(h->d_frames); (void)hipHostFree(h->h_stage);
    (void)hipStreamDestroy(h->stream);
    delete h;
    return BDR_OK;
}

int32_t bdr_atari_prep_reset(bdr_atari_prep* h, uint32_t n, const uint32_t* env_ixs, const uint8_t* frames)
{
    return run(h, n, env_ixs, frames, frames, 1);
}

int32_t bdr_atari_prep_step(bdr_atari_prep* h, uint32_t n, const uint32_t* env_ixs, const uint8_t* frames_a, const uint8_t* frames_b)
{
    return run(h, n, env_ixs, frames_a, frames_b, 0);
}

int32_t bdr_atari_prep_obs(bdr_atari_prep* h, uint32_t n, const uint32_t* env_ixs, uint8_t* obs_out)
{
    BDR_REQUIRE(h && env_ixs && obs_out, "null argument");
    BDR_HIP(hipSetDevice(h->device));
    for (uint32_t k = 0; k < n; ++k) {
        BDR_REQUIRE(env_ixs[k] < h->n_envs, "environment index out of range");
        BDR_HIP(hipMemcpyAsync(obs_out + (size_t)k * 4 * OUT * OUT, h->stacks + (size_t)env_ixs[k] * 4 * OUT * OUT, 4 * OUT * OUT,
                               hipMemcpyDeviceToHost, h->stream));
    }
    BDR_HIP(hipStreamSynchronize(h->stream));
    return BDR_OK;
}

int32_t bdr_atari_prep_device_stacks(bdr_atari_prep* h, const uint8_t** stacks)
{
    BDR_REQUIRE(h && stacks, "null argument");
    *stacks = h->stacks;
    return BDR_OK;
}

int32_t bdr_atari_prep_copy_stack(bdr_atari_prep* h, uint32_t env_ix, void* dst_dev)
{
    BDR_REQUIRE(h && dst_dev, "null argument");
    BDR_REQUIRE(env_ix < h->n_envs, "environment index out of range");
    BDR_HIP(hipSetDevice(h->device));
    BDR_HIP(hipMemcpyAsync(dst_dev, h->stacks + (size_t)env_ix * 4 * OUT * OUT, 4 * OUT * OUT, hipMemcpyDeviceToDevice, h->stream));
    BDR_HIP(hipStreamSynchronize(h->stream));
    return BDR_OK;
}

int32_t bdr_atari_prep_device_prev_stacks(bdr_atari_prep* h, const uint8_t** prev)
{
    BDR_REQUIRE(h && prev, "null argument");
    *prev = h->prev;
    return BDR_OK;
}

float bdr_atari_clip_reward(float r, int32_t train)   // env.rs:159-169
{
    if (!train) return r;
    return r == 0.0f ? 0.0f : (r > 0.0f ? 1.0f : (r < 0.0f ? -1.0f : r));
}

}  // extern "C"
