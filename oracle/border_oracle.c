/*
 * border_oracle.c -- CPU restatement of border's opt-step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under border_amd/ (the product) may link,
 * import or call this file.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  The arithmetic lives in third-party crates that
 * are not vendored under /root/reference:
 *   - rand = "=0.8.5" (Cargo.toml:53): StdRng = rand_chacha 0.3 ChaCha12Rng,
 *     rand_core 0.6 SeedableRng::seed_from_u64 (PCG32 key expansion)
 *   - tch = "0.16.0" (Cargo.toml:31) -> libtorch 2.3.0 ATen ops
 *     (conv2d, linear, relu, gather, argmax, smooth_l1_loss, mse_loss, Adam)
 * so this file restates their published algorithms.
 *
 * PARITY PINNING
 *   RNG: pinned.  ChaCha12 core + word order + next_u64 + from_rng reproduce
 *     rand 0.8.5's own known-answer test `test_stdrng_construction`
 *     (rand/src/rngs/std.rs: seed [1,0,0,0,23,0,0,0,200,1,0,0,210,30,0,0,0..],
 *     targets 10719222850664546238, 14064965282130556830) and the ChaCha20
 *     zero-key vector (76b8e0ad...).  seed_from_u64's PCG32 expansion is restated
 *     from rand_core 0.6 and has no upstream vector available offline.
 *   Float math: the reference holds no golden vector for DQN/IQN/SAC
 *      => "parity unpinned" at the libtorch boundary; the
 *     stand-in pin is PyTorch 2.10 CPU (same ATen op set tch binds) via
 *     oracle/torch_ref.py -> tests/golden/ (npz fixtures).
 *
 * Numerics: storage is f32 like the reference; dot products accumulate in
 * double and round once to f32, so the oracle sits at the centre of the
 * f32-roundoff cloud that ATen (any summation order) and the MFMA kernels
 * (k-ordered fmaf chains) both live in.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* StdRng (ChaCha12) -- generic_replay_buffer/base.rs:353 (seed_from_u64),     */
/* :386 (next_u32)                                                             */
/* ------------------------------------------------------------------------- */

typedef struct {
    uint32_t key[8];
    uint64_t word_pos; /* index of the next u32 word of the key stream */
} orc_rng;

static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }

#define QR(a, b, c, d)                                                                             \
    do {                                                                                           \
        a += b; d ^= a; d = rotl32(d, 16);                                                         \
        c += d; b ^= c; b = rotl32(b, 12);                                                         \
        a += b; d ^= a; d = rotl32(d, 8);                                                          \
        c += d; b ^= c; b = rotl32(b, 7);                                                          \
    } while (0)

/* One ChaCha block (rand_chacha layout: 64-bit block counter in words 12,13,
 * 64-bit stream id (=0) in words 14,15). */
ORC_API void orc_chacha_block(const uint32_t key[8], uint64_t counter, int rounds, uint32_t out[16])
{
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                      key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), 0u, 0u};
    uint32_t w[16];
    memcpy(w, s, sizeof w);
    for (int r = 0; r < rounds; r += 2) {
        QR(w[0], w[4], w[8], w[12]);  QR(w[1], w[5], w[9], w[13]);
        QR(w[2], w[6], w[10], w[14]); QR(w[3], w[7], w[11], w[15]);
        QR(w[0], w[5], w[10], w[15]); QR(w[1], w[6], w[11], w[12]);
        QR(w[2], w[7], w[8], w[13]);  QR(w[3], w[4], w[9], w[14]);
    }
    for (int i = 0; i < 16; ++i) out[i] = w[i] + s[i];
}

/* StdRng::from_seed: 32 little-endian key bytes, counter 0, stream 0. */
ORC_API void orc_rng_from_seed(orc_rng* g, const uint8_t seed[32])
{
    for (int i = 0; i < 8; ++i)
        g->key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) |
                    ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
    g->word_pos = 0;
}

/* rand_core 0.6 SeedableRng::seed_from_u64: PCG32 expands the u64 into the seed. */
ORC_API void orc_seed_bytes_from_u64(uint64_t state, uint8_t seed[32])
{
    for (int i = 0; i < 8; ++i) {
        state = state * 6364136223846793005ULL + 11634580027462260723ULL;
        uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
        seed[4 * i] = (uint8_t)x; seed[4 * i + 1] = (uint8_t)(x >> 8);
        seed[4 * i + 2] = (uint8_t)(x >> 16); seed[4 * i + 3] = (uint8_t)(x >> 24);
    }
}

ORC_API void orc_rng_seed_from_u64(orc_rng* g, uint64_t seed)
{
    uint8_t s[32];
    orc_seed_bytes_from_u64(seed, s);
    orc_rng_from_seed(g, s);
}

/* BlockRng::next_u32: words of block 0, then block 1, ... */
ORC_API uint32_t orc_rng_next_u32(orc_rng* g)
{
    uint32_t blk[16];
    orc_chacha_block(g->key, g->word_pos >> 4, 12, blk);
    return blk[g->word_pos++ & 15];
}

/* BlockRng::next_u64: low word first. */
ORC_API uint64_t orc_rng_next_u64(orc_rng* g)
{
    uint64_t lo = orc_rng_next_u32(g);
    uint64_t hi = orc_rng_next_u32(g);
    return lo | (hi << 32);
}

/* generic_replay_buffer/base.rs:384-390: ixs[k] = (next_u32() as usize) % size */
ORC_API void orc_sample_indices(orc_rng* g, uint64_t size, int n, uint64_t* out)
{
    for (int k = 0; k < n; ++k) out[k] = (uint64_t)orc_rng_next_u32(g) % size;
}

/* ------------------------------------------------------------------------- */
/* SimpleReplayBuffer -- generic_replay_buffer/base.rs:86-123 (state),         */
/* :295-316 (push), :376-402 (batch), :137-219 (reward/flag push + sample);    */
/* storage rows follow TensorBatch (border-tch-agent/src/tensor_batch.rs:85-   */
/* 120): push writes rows (i+k)%capacity, sample = index_select rows.          */
/* Rows are opaque bytes here (u8 Atari frames, f32 vectors, i64 actions).     */
/* ------------------------------------------------------------------------- */

typedef struct {
    uint64_t capacity, i, size;
    uint64_t obs_bytes, act_bytes;
    uint8_t *obs, *act, *next_obs;
    float* reward;
    int8_t *is_terminated, *is_truncated;
    orc_rng rng;
} orc_replay;

ORC_API orc_replay* orc_replay_build(uint64_t capacity, uint64_t seed, uint64_t obs_bytes,
                                     uint64_t act_bytes)
{
    orc_replay* r = (orc_replay*)calloc(1, sizeof *r);
    r->capacity = capacity; r->obs_bytes = obs_bytes; r->act_bytes = act_bytes;
    r->obs = (uint8_t*)calloc(capacity, obs_bytes);
    r->next_obs = (uint8_t*)calloc(capacity, obs_bytes);
    r->act = (uint8_t*)calloc(capacity, act_bytes);
    r->reward = (float*)calloc(capacity, sizeof(float));
    r->is_terminated = (int8_t*)calloc(capacity, 1);
    r->is_truncated = (int8_t*)calloc(capacity, 1);
    orc_rng_seed_from_u64(&r->rng, seed);
    return r;
}

ORC_API void orc_replay_free(orc_replay* r)
{
    if (!r) return;
    free(r->obs); free(r->next_obs); free(r->act); free(r->reward);
    free(r->is_terminated); free(r->is_truncated); free(r);
}

ORC_API uint64_t orc_replay_len(const orc_replay* r) { return r->size; }
ORC_API uint64_t orc_replay_head(const orc_replay* r) { return r->i; }

/* base.rs:295-316 */
ORC_API void orc_replay_push(orc_replay* r, uint64_t len, const uint8_t* obs, const uint8_t* act,
                             const uint8_t* next_obs, const float* reward, const int8_t* term,
                             const int8_t* trunc)
{
    for (uint64_t k = 0; k < len; ++k) {
        uint64_t j = (r->i + k) % r->capacity;
        memcpy(r->obs + j * r->obs_bytes, obs + k * r->obs_bytes, r->obs_bytes);
        memcpy(r->act + j * r->act_bytes, act + k * r->act_bytes, r->act_bytes);
        memcpy(r->next_obs + j * r->obs_bytes, next_obs + k * r->obs_bytes, r->obs_bytes);
        r->reward[j] = reward[k];
        r->is_terminated[j] = term[k];
        r->is_truncated[j] = trunc[k];
    }
    r->i = (r->i + len) % r->capacity;
    r->size += len;
    if (r->size >= r->capacity) r->size = r->capacity;
}

/* base.rs:376-402, uniform branch.  Returns -1 on an empty buffer (the
 * reference panics there: `% self.size` with size == 0). */
ORC_API int orc_replay_batch(orc_replay* r, int n, uint64_t* ixs, uint8_t* obs, uint8_t* act,
                             uint8_t* next_obs, float* reward, int8_t* term, int8_t* trunc)
{
    if (r->size == 0) return -1;
    orc_sample_indices(&r->rng, r->size, n, ixs);
    for (int k = 0; k < n; ++k) {
        uint64_t j = ixs[k];
        memcpy(obs + (uint64_t)k * r->obs_bytes, r->obs + j * r->obs_bytes, r->obs_bytes);
        memcpy(act + (uint64_t)k * r->act_bytes, r->act + j * r->act_bytes, r->act_bytes);
        memcpy(next_obs + (uint64_t)k * r->obs_bytes, r->next_obs + j * r->obs_bytes, r->obs_bytes);
        reward[k] = r->reward[j];
        term[k] = r->is_terminated[j];
        trunc[k] = r->is_truncated[j];
    }
    return 0;
}

/* ------------------------------------------------------------------------- */
/* Layers (libtorch semantics): conv2d = cross-correlation, no padding, OIHW;  */
/* linear = x W^T + b with W [out,in]; relu.                                   */
/* ------------------------------------------------------------------------- */

static void conv2d_fwd(const float* x, const float* w, const float* b, float* y, int B, int C,
                       int H, int W, int O, int K, int S, int relu)
{
    const int OH = (H - K) / S + 1, OW = (W - K) / S + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < B; ++n)
        for (int o = 0; o < O; ++o)
            for (int oh = 0; oh < OH; ++oh)
                for (int ow = 0; ow < OW; ++ow) {
                    double acc = 0.0;
                    for (int c = 0; c < C; ++c)
                        for (int kh = 0; kh < K; ++kh) {
                            const float* xr = x + (((size_t)n * C + c) * H + oh * S + kh) * W + ow * S;
                            const float* wr = w + (((size_t)o * C + c) * K + kh) * K;
                            for (int kw = 0; kw < K; ++kw) acc += (double)xr[kw] * (double)wr[kw];
                        }
                    float v = (float)acc + b[o];
                    if (relu && v < 0.f) v = 0.f;
                    y[(((size_t)n * O + o) * OH + oh) * OW + ow] = v;
                }
}

/* dy is the gradient w.r.t. the conv output (pre-activation). dx may be NULL. */
static void conv2d_bwd(const float* x, const float* w, const float* dy, float* dw, float* db,
                       float* dx, int B, int C, int H, int W, int O, int K, int S)
{
    const int OH = (H - K) / S + 1, OW = (W - K) / S + 1;
#pragma omp parallel for schedule(static)
    for (int o = 0; o < O; ++o) {
        double bacc = 0.0;
        for (int n = 0; n < B; ++n)
            for (int p = 0; p < OH * OW; ++p) bacc += dy[((size_t)n * O + o) * OH * OW + p];
        db[o] = (float)bacc;
        for (int c = 0; c < C; ++c)
            for (int kh = 0; kh < K; ++kh)
                for (int kw = 0; kw < K; ++kw) {
                    double acc = 0.0;
                    for (int n = 0; n < B; ++n)
                        for (int oh = 0; oh < OH; ++oh) {
                            const float* dyr = dy + (((size_t)n * O + o) * OH + oh) * OW;
                            const float* xr = x + (((size_t)n * C + c) * H + oh * S + kh) * W + kw;
                            for (int ow = 0; ow < OW; ++ow) acc += (double)dyr[ow] * (double)xr[ow * S];
                        }
                    dw[(((size_t)o * C + c) * K + kh) * K + kw] = (float)acc;
                }
    }
    if (!dx) return;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < B; ++n)
        for (int c = 0; c < C; ++c)
            for (int ih = 0; ih < H; ++ih)
                for (int iw = 0; iw < W; ++iw) {
                    double acc = 0.0;
                    for (int kh = 0; kh < K; ++kh) {
                        int t = ih - kh;
                        if (t < 0 || t % S) continue;
                        int oh = t / S;
                        if (oh >= OH) continue;
                        for (int kw = 0; kw < K; ++kw) {
                            int u = iw - kw;
                            if (u < 0 || u % S) continue;
                            int ow = u / S;
                            if (ow >= OW) continue;
                            for (int o = 0; o < O; ++o)
                                acc += (double)dy[(((size_t)n * O + o) * OH + oh) * OW + ow] *
                                       (double)w[(((size_t)o * C + c) * K + kh) * K + kw];
                        }
                    }
                    dx[(((size_t)n * C + c) * H + ih) * W + iw] = (float)acc;
                }
}

/* Loop order of the two linear functions: the contraction index is the contiguous one and several outputs share every
 * operand row that is streamed (IQN at batch 512 x 64 quantiles is a [32768][3136] x [3136][512] layer), `omp simd`
 * lets the compiler keep independent partial sums.  Accumulation stays in double: the order of the additions moves the
 * result by ~1e-16 relative, far below the single rounding to f32 that follows. */
static void linear_fwd(const float* x, const float* w, const float* b, float* y, int B, int I,
                       int O, int relu)
{
#pragma omp parallel for schedule(static)
    for (int n = 0; n < B; ++n) {
        const float* xr = x + (size_t)n * I;
        for (int o = 0; o < O; ++o) {
            double acc = 0.0;
            const float* wr = w + (size_t)o * I;
#pragma omp simd reduction(+ : acc)
            for (int i = 0; i < I; ++i) acc += (double)xr[i] * (double)wr[i];
            float v = (float)acc + b[o];
            if (relu && v < 0.f) v = 0.f;
            y[(size_t)n * O + o] = v;
        }
    }
}

static void linear_bwd(const float* x, const float* w, const float* dy, float* dw, float* db,
                       float* dx, int B, int I, int O)
{
    /* dW[o][i] = sum_n dy[n][o] x[n][i]: one thread owns a block of OB outputs and streams the rows of x once for
     * all of them (rows whose dy entries are all zero - most rows of IQN's one-hot dz - are skipped) */
    enum { OB = 4 };
#pragma omp parallel
    {
        double* acc = (double*)malloc(sizeof(double) * (size_t)OB * I);
#pragma omp for schedule(dynamic, 1)
        for (int o0 = 0; o0 < O; o0 += OB) {
            const int ob = O - o0 < OB ? O - o0 : OB;
            double bacc[OB] = {0.0, 0.0, 0.0, 0.0};
            for (size_t q = 0; q < (size_t)ob * I; ++q) acc[q] = 0.0;
            for (int n = 0; n < B; ++n) {
                const float* xr = x + (size_t)n * I;
                for (int k = 0; k < ob; ++k) {
                    const double d = (double)dy[(size_t)n * O + o0 + k];
                    if (d == 0.0) continue;
                    bacc[k] += d;
                    double* ak = acc + (size_t)k * I;
#pragma omp simd
                    for (int i = 0; i < I; ++i) ak[i] += d * (double)xr[i];
                }
            }
            for (int k = 0; k < ob; ++k) {
                db[o0 + k] = (float)bacc[k];
                for (int i = 0; i < I; ++i) dw[(size_t)(o0 + k) * I + i] = (float)acc[(size_t)k * I + i];
            }
        }
        free(acc);
    }
    if (!dx) return;
    /* dX[n][i] = sum_o dy[n][o] w[o][i]: rows of w are contiguous in i */
#pragma omp parallel
    {
        double* acc = (double*)malloc(sizeof(double) * (size_t)I);
#pragma omp for schedule(static)
        for (int n = 0; n < B; ++n) {
            for (int i = 0; i < I; ++i) acc[i] = 0.0;
            for (int o = 0; o < O; ++o) {
                const double d = (double)dy[(size_t)n * O + o];
                if (d == 0.0) continue;
                const float* wr = w + (size_t)o * I;
#pragma omp simd
                for (int i = 0; i < I; ++i) acc[i] += d * (double)wr[i];
            }
            for (int i = 0; i < I; ++i) dx[(size_t)n * I + i] = (float)acc[i];
        }
        free(acc);
    }
}

/* multiply gradient by relu'(post-activation value) */
static void relu_bwd(const float* y_post, float* dy, size_t n)
{
    for (size_t i = 0; i < n; ++i)
        if (!(y_post[i] > 0.f)) dy[i] = 0.f;
}

/* ------------------------------------------------------------------------- */
/* Networks.                                                                   */
/*  kind 0: AtariCnn  (border-tch-agent/src/cnn/base.rs:23-36): input u8       */
/*          [B,n_stack,1,84,84]; x.squeeze(2).float()/255; c1(k8,s4)+relu;     */
/*          c2(k4,s2)+relu; c3(k3,s1)+relu; flatten(C,H,W); l1+relu; l2.       */
/*          Parameter order in the flat vector: c1.weight c1.bias c2.weight    */
/*          c2.bias c3.weight c3.bias l1.weight l1.bias l2.weight l2.bias      */
/*          (OIHW / [out,in], the libtorch layouts).                           */
/*  kind 1: Mlp (border-tch-agent/src/mlp/base.rs:13-41): in->units..->out,    */
/*          relu between, optional relu on the output; order mlp.ln{i}.weight, */
/*          mlp.ln{i}.bias.  Input f32 [B,in_dim].                             */
/* ------------------------------------------------------------------------- */

#define ORC_MAX_UNITS 8
typedef struct {
    int32_t kind;      /* 0 AtariCnn, 1 Mlp */
    int32_t n_stack;   /* cnn */
    int32_t in_dim;    /* mlp */
    int32_t n_units;   /* mlp */
    int32_t units[ORC_MAX_UNITS];
    int32_t out_dim;
    int32_t activation_out; /* mlp */
} orc_net_cfg;

ORC_API int64_t orc_net_param_count(const orc_net_cfg* c)
{
    if (c->kind == 0) {
        int64_t n = 32LL * c->n_stack * 64 + 32;
        n += 64LL * 32 * 16 + 64;
        n += 64LL * 64 * 9 + 64;
        n += 512LL * 3136 + 512;
        n += (int64_t)c->out_dim * 512 + c->out_dim;
        return n;
    }
    int64_t n = 0; int in = c->in_dim;
    for (int i = 0; i < c->n_units; ++i) { n += (int64_t)c->units[i] * in + c->units[i]; in = c->units[i]; }
    n += (int64_t)c->out_dim * in + c->out_dim;
    return n;
}

/* activation cache; sizes for the largest case are computed at alloc time */
typedef struct {
    int B;
    float* x0;  /* cnn: normalised input [B,4,84,84]; mlp: copy of input */
    float* a[ORC_MAX_UNITS + 2]; /* post-activation outputs of each layer; last = net output */
    int n_layers;
} orc_cache;

static orc_cache* cache_alloc(const orc_net_cfg* c, int B)
{
    orc_cache* k = (orc_cache*)calloc(1, sizeof *k);
    k->B = B;
    if (c->kind == 0) {
        k->n_layers = 5;
        k->x0 = (float*)malloc(sizeof(float) * (size_t)B * c->n_stack * 84 * 84);
        k->a[0] = (float*)malloc(sizeof(float) * (size_t)B * 32 * 20 * 20);
        k->a[1] = (float*)malloc(sizeof(float) * (size_t)B * 64 * 9 * 9);
        k->a[2] = (float*)malloc(sizeof(float) * (size_t)B * 64 * 7 * 7);
        k->a[3] = (float*)malloc(sizeof(float) * (size_t)B * 512);
        k->a[4] = (float*)malloc(sizeof(float) * (size_t)B * c->out_dim);
    } else {
        k->n_layers = c->n_units + 1;
        k->x0 = (float*)malloc(sizeof(float) * (size_t)B * c->in_dim);
        for (int i = 0; i < c->n_units; ++i) k->a[i] = (float*)malloc(sizeof(float) * (size_t)B * c->units[i]);
        k->a[c->n_units] = (float*)malloc(sizeof(float) * (size_t)B * c->out_dim);
    }
    return k;
}

static void cache_free(orc_cache* k)
{
    if (!k) return;
    free(k->x0);
    for (int i = 0; i < k->n_layers; ++i) free(k->a[i]);
    free(k);
}

static const float* net_out(const orc_cache* k) { return k->a[k->n_layers - 1]; }

static void net_forward(const orc_net_cfg* c, const float* p, const void* input, orc_cache* k)
{
    const int B = k->B;
    if (c->kind == 0) {
        const uint8_t* u = (const uint8_t*)input;
        const size_t n = (size_t)B * c->n_stack * 84 * 84;
        /* cnn/base.rs:26  xs.squeeze_dim(2).internal_cast_float(true) / 255 */
        for (size_t i = 0; i < n; ++i) k->x0[i] = (float)u[i] / 255.0f;
        const float* w1 = p; const float* b1 = w1 + 32 * c->n_stack * 64;
        const float* w2 = b1 + 32; const float* b2 = w2 + 64 * 32 * 16;
        const float* w3 = b2 + 64; const float* b3 = w3 + 64 * 64 * 9;
        const float* w4 = b3 + 64; const float* b4 = w4 + 512 * 3136;
        const float* w5 = b4 + 512; const float* b5 = w5 + (size_t)c->out_dim * 512;
        conv2d_fwd(k->x0, w1, b1, k->a[0], B, c->n_stack, 84, 84, 32, 8, 4, 1);
        conv2d_fwd(k->a[0], w2, b2, k->a[1], B, 32, 20, 20, 64, 4, 2, 1);
        conv2d_fwd(k->a[1], w3, b3, k->a[2], B, 64, 9, 9, 64, 3, 1, 1);
        linear_fwd(k->a[2], w4, b4, k->a[3], B, 3136, 512, 1);
        linear_fwd(k->a[3], w5, b5, k->a[4], B, 512, c->out_dim, 0);
    } else {
        memcpy(k->x0, input, sizeof(float) * (size_t)B * c->in_dim);
        const float* x = k->x0; int in = c->in_dim; const float* q = p;
        for (int i = 0; i < c->n_units; ++i) {
            linear_fwd(x, q, q + (size_t)c->units[i] * in, k->a[i], B, in, c->units[i], 1);
            q += (size_t)c->units[i] * in + c->units[i]; x = k->a[i]; in = c->units[i];
        }
        linear_fwd(x, q, q + (size_t)c->out_dim * in, k->a[c->n_units], B, in, c->out_dim, c->activation_out);
    }
}

/* dout: gradient w.r.t. the network output (post activation_out).  grads gets
 * the gradient of every parameter, same layout as p. */
static void net_backward(const orc_net_cfg* c, const float* p, const orc_cache* k, const float* dout,
                         float* g)
{
    const int B = k->B;
    if (c->kind == 0) {
        const size_t o1 = 0, ob1 = o1 + 32 * c->n_stack * 64, o2 = ob1 + 32, ob2 = o2 + 64 * 32 * 16,
                     o3 = ob2 + 64, ob3 = o3 + 64 * 64 * 9, o4 = ob3 + 64, ob4 = o4 + 512 * 3136,
                     o5 = ob4 + 512, ob5 = o5 + (size_t)c->out_dim * 512;
        float* d4 = (float*)malloc(sizeof(float) * (size_t)B * 512);
        float* d3 = (float*)malloc(sizeof(float) * (size_t)B * 3136);
        float* d2 = (float*)malloc(sizeof(float) * (size_t)B * 64 * 81);
        float* d1 = (float*)malloc(sizeof(float) * (size_t)B * 32 * 400);
        linear_bwd(k->a[3], p + o5, dout, g + o5, g + ob5, d4, B, 512, c->out_dim);
        relu_bwd(k->a[3], d4, (size_t)B * 512);
        linear_bwd(k->a[2], p + o4, d4, g + o4, g + ob4, d3, B, 3136, 512);
        relu_bwd(k->a[2], d3, (size_t)B * 3136);
        conv2d_bwd(k->a[1], p + o3, d3, g + o3, g + ob3, d2, B, 64, 9, 9, 64, 3, 1);
        relu_bwd(k->a[1], d2, (size_t)B * 64 * 81);
        conv2d_bwd(k->a[0], p + o2, d2, g + o2, g + ob2, d1, B, 32, 20, 20, 64, 4, 2);
        relu_bwd(k->a[0], d1, (size_t)B * 32 * 400);
        conv2d_bwd(k->x0, p + o1, d1, g + o1, g + ob1, NULL, B, c->n_stack, 84, 84, 32, 8, 4);
        free(d4); free(d3); free(d2); free(d1);
    } else {
        const int L = c->n_units + 1;
        size_t off[ORC_MAX_UNITS + 2]; int ins[ORC_MAX_UNITS + 2] = {0}, outs[ORC_MAX_UNITS + 2] = {0};
        size_t o = 0; int in = c->in_dim;
        for (int i = 0; i < L; ++i) {
            int out = i < c->n_units ? c->units[i] : c->out_dim;
            off[i] = o; ins[i] = in; outs[i] = out; o += (size_t)out * in + out; in = out;
        }
        float* d = (float*)malloc(sizeof(float) * (size_t)B * outs[L - 1]);
        memcpy(d, dout, sizeof(float) * (size_t)B * outs[L - 1]);
        if (c->activation_out) relu_bwd(k->a[L - 1], d, (size_t)B * outs[L - 1]);
        for (int i = L - 1; i >= 0; --i) {
            const float* x = i == 0 ? k->x0 : k->a[i - 1];
            float* dx = i == 0 ? NULL : (float*)malloc(sizeof(float) * (size_t)B * ins[i]);
            linear_bwd(x, p + off[i], d, g + off[i], g + off[i] + (size_t)outs[i] * ins[i], dx, B, ins[i], outs[i]);
            if (dx) relu_bwd(k->a[i - 1], dx, (size_t)B * ins[i]);
            free(d); d = dx;
        }
    }
}

/* Q-values only (Policy::sample / tests).  out: [B,out_dim]. */
ORC_API void orc_net_forward(const orc_net_cfg* c, const float* params, const void* input, int B,
                             float* out)
{
    orc_cache* k = cache_alloc(c, B);
    net_forward(c, params, input, k);
    memcpy(out, net_out(k), sizeof(float) * (size_t)B * c->out_dim);
    cache_free(k);
}

/* ------------------------------------------------------------------------- */
/* Adam -- opt.rs:35 (tch nn::Adam::default(): beta1 .9, beta2 .999, wd 0,     */
/* eps 1e-8, amsgrad false) + libtorch torch/csrc/api/src/optim/adam.cpp:      */
/*   exp_avg.mul_(b1).add_(g, 1-b1); exp_avg_sq.mul_(b2).addcmul_(g,g,1-b2);   */
/*   denom = (exp_avg_sq.sqrt() / sqrt(1-b2^t)).add_(eps);                     */
/*   p.addcdiv_(exp_avg, denom, -(lr/(1-b1^t)))                                */
/* Scalars are doubles on the host and enter the f32 element kernels as f32.   */
/* ------------------------------------------------------------------------- */
typedef struct {
    double lr, beta1, beta2, eps;
    int64_t step;
} orc_adam_cfg;

ORC_API void orc_adam_step(orc_adam_cfg* a, float* p, const float* g, float* m, float* v, int64_t n)
{
    a->step += 1;
    const double bc1 = 1.0 - pow(a->beta1, (double)a->step);
    const double bc2 = 1.0 - pow(a->beta2, (double)a->step);
    const float b1 = (float)a->beta1, omb1 = (float)(1.0 - a->beta1);
    const float b2 = (float)a->beta2, omb2 = (float)(1.0 - a->beta2);
    const float sbc2 = (float)sqrt(bc2), eps = (float)a->eps;
    const float neg_step = (float)(-(a->lr / bc1));
    for (int64_t i = 0; i < n; ++i) {
        m[i] = m[i] * b1 + g[i] * omb1;
        v[i] = v[i] * b2 + omb2 * g[i] * g[i];
        float denom = sqrtf(v[i]) / sbc2 + eps;
        p[i] = p[i] + neg_step * m[i] / denom;
    }
}

/* util.rs:31-45  dest = tau*src + (1-tau)*dest, per variable (elementwise) */
ORC_API void orc_track(float* dest, const float* src, double tau, int64_t n)
{
    const float t = (float)tau, omt = (float)(1.0 - tau);
    for (int64_t i = 0; i < n; ++i) dest[i] = t * src[i] + omt * dest[i];
}

/* ------------------------------------------------------------------------- */
/* Dqn::update_critic -- border-tch-agent/src/dqn/base.rs:60-160               */
/* ------------------------------------------------------------------------- */
typedef struct {
    double discount_factor;
    int32_t double_dqn;
    int32_t critic_loss;  /* 0 = Mse, 1 = SmoothL1 (util.rs:17-23) */
    int32_t has_clip_td_err;
    double clip_min, clip_max;
} orc_dqn_cfg;

typedef struct {
    float* q_pred_all; /* [B,A] online Q(obs)            or NULL */
    float* q_next_all; /* [B,A] target-net Q(next_obs)   or NULL */
    float* pred;       /* [B]                             or NULL */
    float* tgt;        /* [B]                             or NULL */
    float* grads;      /* [n_params] dLoss/dparams        or NULL */
    float* td_abs;     /* [B] |pred-tgt| (PER priorities) or NULL */
} orc_dqn_probe;

/* argmax(-1): first maximal index, like at::argmax on CPU */
static int argmax_row(const float* r, int A)
{
    int best = 0;
    for (int a = 1; a < A; ++a)
        if (r[a] > r[best]) best = a;
    return best;
}

/* One critic update on a given minibatch.  act: int64 [B]; term: int8 [B];
 * weight: PER importance weights [B] or NULL (base.rs:123-145 branch).
 * Updates qnet params + Adam state in place; returns the loss. */
ORC_API float orc_dqn_update(const orc_net_cfg* net, float* qnet, const float* qnet_tgt,
                             orc_adam_cfg* adam, float* adam_m, float* adam_v,
                             const orc_dqn_cfg* cfg, int B, const void* obs, const int64_t* act,
                             const void* next_obs, const float* reward, const int8_t* term,
                             const float* weight, orc_dqn_probe* probe)
{
    const int A = net->out_dim;
    const int64_t np = orc_net_param_count(net);
    orc_cache* k = cache_alloc(net, B);
    orc_cache* kt = cache_alloc(net, B);
    float* pred = (float*)malloc(sizeof(float) * B);
    float* tgt = (float*)malloc(sizeof(float) * B);
    float* dq = (float*)calloc((size_t)B * A, sizeof(float));
    float* g = (float*)calloc((size_t)np, sizeof(float));

    /* :71-74  pred = qnet(obs).gather(-1, act).squeeze() */
    net_forward(net, qnet, obs, k);
    const float* q = net_out(k);
    for (int b = 0; b < B; ++b) pred[b] = q[(size_t)b * A + act[b]];
    if (probe && probe->q_pred_all) memcpy(probe->q_pred_all, q, sizeof(float) * (size_t)B * A);

    /* :91-105 target (no_grad) */
    const float gamma = (float)cfg->discount_factor;
    if (cfg->double_dqn) {
        orc_cache* kn = cache_alloc(net, B);
        net_forward(net, qnet, next_obs, kn);
        net_forward(net, qnet_tgt, next_obs, kt);
        for (int b = 0; b < B; ++b) {
            int y = argmax_row(net_out(kn) + (size_t)b * A, A);
            float qq = net_out(kt)[(size_t)b * A + y];
            tgt[b] = reward[b] + ((float)(1 - term[b]) * gamma) * qq;
        }
        cache_free(kn);
    } else {
        net_forward(net, qnet_tgt, next_obs, kt);
        for (int b = 0; b < B; ++b) {
            const float* r = net_out(kt) + (size_t)b * A;
            float qq = r[argmax_row(r, A)];
            tgt[b] = reward[b] + ((float)(1 - term[b]) * gamma) * qq;
        }
    }
    if (probe && probe->q_next_all) memcpy(probe->q_next_all, net_out(kt), sizeof(float) * (size_t)B * A);

    /* :123-152 loss and dLoss/dpred */
    double lsum = 0.0;
    for (int b = 0; b < B; ++b) {
        float d = pred[b] - tgt[b];
        float dl; /* d loss_b / d pred_b before the 1/B of Reduction::Mean */
        if (weight) {
            /* loss_b = criterion(w * clip(|pred-tgt|), 0) */
            float td = fabsf(d), s = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            float dtd = 1.f;
            if (cfg->has_clip_td_err) {
                if (td < (float)cfg->clip_min) { td = (float)cfg->clip_min; dtd = 0.f; }
                else if (td > (float)cfg->clip_max) { td = (float)cfg->clip_max; dtd = 0.f; }
            }
            if (probe && probe->td_abs) probe->td_abs[b] = td;
            float l = weight[b] * td;
            if (cfg->critic_loss == 1) {
                float z = fabsf(l);
                lsum += z < 1.f ? 0.5 * (double)z * z : (double)z - 0.5;
                dl = (z < 1.f ? l : (l > 0.f ? 1.f : -1.f)) * weight[b] * dtd * s;
            } else {
                lsum += (double)l * l;
                dl = 2.f * l * weight[b] * dtd * s;
            }
        } else {
            if (probe && probe->td_abs) probe->td_abs[b] = fabsf(d);
            if (cfg->critic_loss == 1) { /* smooth_l1_loss(beta = 1.0, Mean) */
                float z = fabsf(d);
                lsum += z < 1.f ? 0.5 * (double)z * z : (double)z - 0.5;
                dl = z < 1.f ? d : (d > 0.f ? 1.f : -1.f);
            } else { /* mse_loss(Mean) */
                lsum += (double)d * d;
                dl = 2.f * d;
            }
        }
        dq[(size_t)b * A + act[b]] = dl / (float)B;
    }
    const float loss = (float)(lsum / B);

    /* :150 qnet.backward_step(loss) = zero_grad; backward; Adam step (opt.rs:74-83) */
    net_backward(net, qnet, k, dq, g);
    if (probe && probe->grads) memcpy(probe->grads, g, sizeof(float) * (size_t)np);
    if (probe && probe->pred) memcpy(probe->pred, pred, sizeof(float) * B);
    if (probe && probe->tgt) memcpy(probe->tgt, tgt, sizeof(float) * B);
    orc_adam_step(adam, qnet, g, adam_m, adam_v, np);

    cache_free(k); cache_free(kt);
    free(pred); free(tgt); free(dq); free(g);
    return loss;
}

ORC_API int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
ORC_API void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}

/* ========================================================================= */
/* SAC -- border-tch-agent/src/sac/base.rs:73-198 (action_logp, qvals_min,     */
/* update_actor, update_critic, soft_update, opt_), mlp/mlp2.rs:23-50 (Mlp2:   */
/* trunk "al{i}" + heads "ml"/"sl", forward returns exp(head2)), mlp/base.rs:  */
/* 83-107 (critic = Mlp on cat(obs,act)), sac/ent_coef.rs:27-75.               */
/* The N(0,1) draws of action_logp (sac/base.rs:76, torch global CPU RNG) are  */
/* inputs here so that a fixed minibatch is reproducible.                      */
/* Quirks kept: sigma = exp(clip(exp(head2), min_lstd, max_lstd)); log-prob    */
/* omits -sum(ln sigma); is_truncated ignored; actor is updated BEFORE the     */
/* critics and the critic target uses the updated actor; track every update.   */
/* ========================================================================= */
typedef struct {
    int32_t obs_dim, act_dim;
    int32_t n_pi_units, pi_units[ORC_MAX_UNITS];
    int32_t n_q_units, q_units[ORC_MAX_UNITS];
    int32_t n_critics;
    double gamma, tau, epsilon, min_lstd, max_lstd, reward_scale;
    int32_t critic_loss;       /* 0 Mse, 1 SmoothL1 */
    int32_t auto_alpha;        /* EntCoefMode::Auto */
    double target_entropy;
} orc_sac_cfg;

typedef struct { int n; int in[ORC_MAX_UNITS + 2], out[ORC_MAX_UNITS + 2], relu[ORC_MAX_UNITS + 2]; size_t off[ORC_MAX_UNITS + 2]; size_t total; } sac_mlp;

static sac_mlp sac_mlp_make(int in_dim, const int32_t* units, int n_units, int out_dim, int with_out)
{
    sac_mlp m; memset(&m, 0, sizeof m);
    int in = in_dim; size_t o = 0;
    for (int i = 0; i < n_units + (with_out ? 1 : 0); ++i) {
        int out = i < n_units ? units[i] : out_dim;
        m.in[i] = in; m.out[i] = out; m.relu[i] = i < n_units; m.off[i] = o;
        o += (size_t)out * in + out; in = out; m.n++;
    }
    m.total = o;
    return m;
}

/* acts[i] = output of layer i (post-relu where applicable); x = input */
static void sac_mlp_fwd(const sac_mlp* m, const float* p, const float* x, int B, float** acts)
{
    const float* in = x;
    for (int i = 0; i < m->n; ++i) {
        linear_fwd(in, p + m->off[i], p + m->off[i] + (size_t)m->out[i] * m->in[i], acts[i], B, m->in[i], m->out[i], m->relu[i]);
        in = acts[i];
    }
}

/* dout: grad wrt the last layer's output (post-activation if relu).  g may be NULL (no param
 * grads wanted); dx_in may be NULL. */
static void sac_mlp_bwd(const sac_mlp* m, const float* p, const float* x, int B, float** acts, const float* dout,
                        float* g, float* dx_in)
{
    float* d = (float*)malloc(sizeof(float) * (size_t)B * m->out[m->n - 1]);
    memcpy(d, dout, sizeof(float) * (size_t)B * m->out[m->n - 1]);
    for (int i = m->n - 1; i >= 0; --i) {
        if (m->relu[i]) relu_bwd(acts[i], d, (size_t)B * m->out[i]);
        const float* in = i == 0 ? x : acts[i - 1];
        float* dx = (i == 0 && !dx_in) ? NULL : (float*)malloc(sizeof(float) * (size_t)B * m->in[i]);
        float* gw = g ? g + m->off[i] : (float*)malloc(sizeof(float) * ((size_t)m->out[i] * m->in[i] + m->out[i]));
        linear_bwd(in, p + m->off[i], d, gw, gw + (size_t)m->out[i] * m->in[i], dx, B, m->in[i], m->out[i]);
        if (!g) free(gw);
        free(d);
        d = dx;
    }
    if (dx_in && d) { memcpy(dx_in, d, sizeof(float) * (size_t)B * m->in[0]); }
    free(d);
}

static float** acts_alloc(const sac_mlp* m, int B)
{
    float** a = (float**)calloc(m->n, sizeof(float*));
    for (int i = 0; i < m->n; ++i) a[i] = (float*)malloc(sizeof(float) * (size_t)B * m->out[i]);
    return a;
}
static void acts_free(const sac_mlp* m, float** a) { for (int i = 0; i < m->n; ++i) free(a[i]); free(a); }

ORC_API int64_t orc_sac_pi_param_count(const orc_sac_cfg* c)
{
    sac_mlp t = sac_mlp_make(c->obs_dim, c->pi_units, c->n_pi_units, 0, 0);
    int h = c->n_pi_units ? c->pi_units[c->n_pi_units - 1] : c->obs_dim;
    return (int64_t)t.total + 2 * ((int64_t)c->act_dim * h + c->act_dim);
}
ORC_API int64_t orc_sac_q_param_count(const orc_sac_cfg* c)
{
    return (int64_t)sac_mlp_make(c->obs_dim + c->act_dim, c->q_units, c->n_q_units, 1, 1).total;
}

/* action_logp (sac/base.rs:73-87).  Outputs a[B][A], logp[B]; optional intermediates for backward. */
typedef struct { float** acts; float *mean, *e, *s, *sd, *a; sac_mlp trunk; int h; } sac_pi_cache;

static void sac_action_logp(const orc_sac_cfg* c, const float* pi, const float* o, const float* z, int B, float* a_out,
                            float* logp, sac_pi_cache* k)
{
    const int A = c->act_dim;
    sac_mlp trunk = sac_mlp_make(c->obs_dim, c->pi_units, c->n_pi_units, 0, 0);
    const int h = c->n_pi_units ? c->pi_units[c->n_pi_units - 1] : c->obs_dim;
    float** acts = acts_alloc(&trunk, B);
    sac_mlp_fwd(&trunk, pi, o, B, acts);
    const float* hid = trunk.n ? acts[trunk.n - 1] : o;
    const float* wm = pi + trunk.total; const float* bm = wm + (size_t)A * h;
    const float* ws = bm + A; const float* bs = ws + (size_t)A * h;
    float* mean = (float*)malloc(sizeof(float) * (size_t)B * A);
    float* e = (float*)malloc(sizeof(float) * (size_t)B * A);
    float* s = (float*)malloc(sizeof(float) * (size_t)B * A);
    float* sd = (float*)malloc(sizeof(float) * (size_t)B * A);
    linear_fwd(hid, wm, bm, mean, B, h, A, 0);
    linear_fwd(hid, ws, bs, e, B, h, A, 0);
    const float lo = (float)c->min_lstd, hi = (float)c->max_lstd, eps = (float)c->epsilon;
    const float cst = (float)(-0.5 * log(2.0 * 3.14159265358979323846));   /* f32 pi in the reference: same value after rounding */
    for (int b = 0; b < B; ++b) {
        float nl = 0.f, sl = 0.f;
        for (int j = 0; j < A; ++j) {
            const size_t q = (size_t)b * A + j;
            s[q] = expf(e[q]);                                   /* Mlp2::forward: .exp() */
            float cl = s[q] < lo ? lo : (s[q] > hi ? hi : s[q]); /* lstd.clip(min_lstd, max_lstd) */
            sd[q] = expf(cl);                                    /* .exp() */
            const float u = sd[q] * z[q] + mean[q];
            const float av = tanhf(u);
            a_out[q] = av;
            nl += cst - 0.5f * (z[q] * z[q]);
            sl += logf((1.0f - av * av) + eps);
        }
        logp[b] = nl - sl;
    }
    if (k) { k->acts = acts; k->mean = mean; k->e = e; k->s = s; k->sd = sd; k->trunk = trunk; k->h = h; k->a = NULL; }
    else { acts_free(&trunk, acts); free(mean); free(e); free(s); free(sd); }
}

typedef struct {
    float loss_critic, loss_actor, ent_coef;
} orc_sac_record;

typedef struct {
    float* pi_grads;     /* [pi params]            or NULL */
    float* q_grads;      /* [n_critics][q params]  or NULL */
    float* a;            /* [B][A] actor action    or NULL */
    float* log_p;        /* [B]                    or NULL */
    float* tgt;          /* [B]                    or NULL */
    float* q_pi;         /* [n_critics][B] Q_i(obs, a_pi), update_actor (sac/base.rs:157-158)          or NULL */
    float* q_pred;       /* [n_critics][B] Q_i(obs, act), update_critic (:128-131)                     or NULL */
    float* q_next;       /* [n_critics][B] target critics on (next_obs, a') (:112-118)                 or NULL */
    float* next_log_p;   /* [B] log p(a' | next_obs) under the updated actor (:113)                    or NULL */
    float* next_a;       /* [B][A] a'                                                                 or NULL */
} orc_sac_probe;

/* One Sac::opt_ update (sac/base.rs:175-198 body of the loop) on a given minibatch. */
ORC_API void orc_sac_update(const orc_sac_cfg* c, float* pi, float** qs, float** qs_tgt, float* log_alpha,
                            orc_adam_cfg* adam_pi, float* pi_m, float* pi_v,
                            orc_adam_cfg* adam_q, float** q_m, float** q_v,          /* adam_q: array [n_critics] */
                            orc_adam_cfg* adam_alpha, float* alpha_m, float* alpha_v,
                            int B, const float* obs, const float* act, const float* next_obs, const float* reward,
                            const int8_t* term, const float* z_actor, const float* z_next, orc_sac_record* rec,
                            orc_sac_probe* probe)
{
    const int A = c->act_dim, O = c->obs_dim, NC = c->n_critics;
    const int64_t npi = orc_sac_pi_param_count(c), nq = orc_sac_q_param_count(c);
    sac_mlp qm = sac_mlp_make(O + A, c->q_units, c->n_q_units, 1, 1);
    const float lo = (float)c->min_lstd, hi = (float)c->max_lstd, eps = (float)c->epsilon;

    /* ---------------- update_actor (:151-167) ---------------- */
    float* a = (float*)malloc(sizeof(float) * (size_t)B * A);
    float* logp = (float*)malloc(sizeof(float) * B);
    sac_pi_cache k;
    sac_action_logp(c, pi, obs, z_actor, B, a, logp, &k);
    if (probe && probe->a) memcpy(probe->a, a, sizeof(float) * (size_t)B * A);
    if (probe && probe->log_p) memcpy(probe->log_p, logp, sizeof(float) * B);
    if (c->auto_alpha) {   /* ent_coef.rs:69-75: loss = -(log_alpha * (logp + H)).mean() */
        double s = 0.0;
        for (int b = 0; b < B; ++b) s += (double)(logp[b] + (float)c->target_entropy);
        float g = (float)(-(s / B));
        orc_adam_step(adam_alpha, log_alpha, &g, alpha_m, alpha_v, 1);
    }
    const float alpha = expf(*log_alpha);
    /* critics on (obs, a): min over critics and its gradient w.r.t. a */
    float* xin = (float*)malloc(sizeof(float) * (size_t)B * (O + A));
    for (int b = 0; b < B; ++b) { memcpy(xin + (size_t)b * (O + A), obs + (size_t)b * O, sizeof(float) * O); memcpy(xin + (size_t)b * (O + A) + O, a + (size_t)b * A, sizeof(float) * A); }
    float* qv = (float*)malloc(sizeof(float) * (size_t)NC * B);
    float*** qacts = (float***)calloc(NC, sizeof(float**));
    for (int i = 0; i < NC; ++i) { qacts[i] = acts_alloc(&qm, B); sac_mlp_fwd(&qm, qs[i], xin, B, qacts[i]); memcpy(qv + (size_t)i * B, qacts[i][qm.n - 1], sizeof(float) * B); }
    if (probe && probe->q_pi) memcpy(probe->q_pi, qv, sizeof(float) * (size_t)NC * B);
    double la = 0.0;
    int* imin = (int*)malloc(sizeof(int) * B);
    for (int b = 0; b < B; ++b) {
        int im = 0;
        for (int i = 1; i < NC; ++i) if (qv[(size_t)i * B + b] < qv[(size_t)im * B + b]) im = i;
        imin[b] = im;
        la += (double)(alpha * logp[b] - qv[(size_t)im * B + b]);
    }
    const float loss_actor = (float)(la / B);
    float* dqda = (float*)calloc((size_t)B * A, sizeof(float));   /* d(qmin)/d(a) */
    for (int i = 0; i < NC; ++i) {
        float* dout = (float*)calloc(B, sizeof(float));
        int any = 0;
        for (int b = 0; b < B; ++b) if (imin[b] == i) { dout[b] = 1.0f; any = 1; }
        if (any) {
            float* dx = (float*)malloc(sizeof(float) * (size_t)B * (O + A));
            sac_mlp_bwd(&qm, qs[i], xin, B, qacts[i], dout, NULL, dx);
            for (int b = 0; b < B; ++b) for (int j = 0; j < A; ++j) dqda[(size_t)b * A + j] += dx[(size_t)b * (O + A) + O + j];
            free(dx);
        }
        free(dout);
    }
    /* dL/dmean, dL/de */
    float* gmean = (float*)malloc(sizeof(float) * (size_t)B * A);
    float* ge = (float*)malloc(sizeof(float) * (size_t)B * A);
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < A; ++j) {
            const size_t q = (size_t)b * A + j;
            const float av = a[q];
            const float dlogp_da = (2.0f * av) / ((1.0f - av * av) + eps);   /* d(-ln(1-a^2+eps))/da */
            const float ga = (alpha * dlogp_da - dqda[q]) / (float)B;
            const float gu = ga * (1.0f - av * av);                           /* tanh' */
            gmean[q] = gu;
            const float inr = (k.s[q] >= lo && k.s[q] <= hi) ? 1.0f : 0.0f;   /* clamp backward */
            ge[q] = gu * z_actor[q] * k.sd[q] * inr * k.s[q];
        }
    /* heads + trunk backward */
    float* gpi = (float*)calloc((size_t)npi, sizeof(float));
    {
        const int h = k.h;
        const float* hid = k.trunk.n ? k.acts[k.trunk.n - 1] : obs;
        float* wm = pi + k.trunk.total; float* ws = wm + (size_t)A * h + A;
        float* gwm = gpi + k.trunk.total; float* gws = gwm + (size_t)A * h + A;
        float* dh1 = (float*)malloc(sizeof(float) * (size_t)B * h);
        float* dh2 = (float*)malloc(sizeof(float) * (size_t)B * h);
        linear_bwd(hid, wm, gmean, gwm, gwm + (size_t)A * h, dh1, B, h, A);
        linear_bwd(hid, ws, ge, gws, gws + (size_t)A * h, dh2, B, h, A);
        for (size_t t = 0; t < (size_t)B * h; ++t) dh1[t] += dh2[t];
        if (k.trunk.n) sac_mlp_bwd(&k.trunk, pi, obs, B, k.acts, dh1, gpi, NULL);
        free(dh1); free(dh2);
    }
    if (probe && probe->pi_grads) memcpy(probe->pi_grads, gpi, sizeof(float) * (size_t)npi);
    orc_adam_step(adam_pi, pi, gpi, pi_m, pi_v, npi);
    acts_free(&k.trunk, k.acts); free(k.mean); free(k.e); free(k.s); free(k.sd);
    for (int i = 0; i < NC; ++i) acts_free(&qm, qacts[i]);
    free(qacts); free(qv); free(imin); free(dqda); free(gmean); free(ge); free(gpi);

    /* ---------------- update_critic (:107-149) ---------------- */
    float* na = (float*)malloc(sizeof(float) * (size_t)B * A);
    float* nlogp = (float*)malloc(sizeof(float) * B);
    sac_action_logp(c, pi, next_obs, z_next, B, na, nlogp, NULL);   /* updated actor */
    if (probe && probe->next_log_p) memcpy(probe->next_log_p, nlogp, sizeof(float) * B);
    if (probe && probe->next_a) memcpy(probe->next_a, na, sizeof(float) * (size_t)B * A);
    for (int b = 0; b < B; ++b) { memcpy(xin + (size_t)b * (O + A), next_obs + (size_t)b * O, sizeof(float) * O); memcpy(xin + (size_t)b * (O + A) + O, na + (size_t)b * A, sizeof(float) * A); }
    float* tgt = (float*)malloc(sizeof(float) * B);
    {
        float** ta = acts_alloc(&qm, B);
        float* nq = (float*)malloc(sizeof(float) * B);
        for (int i = 0; i < NC; ++i) {
            sac_mlp_fwd(&qm, qs_tgt[i], xin, B, ta);
            if (probe && probe->q_next) memcpy(probe->q_next + (size_t)i * B, ta[qm.n - 1], sizeof(float) * B);
            for (int b = 0; b < B; ++b) nq[b] = (i == 0 || ta[qm.n - 1][b] < nq[b]) ? ta[qm.n - 1][b] : nq[b];
        }
        const float gm = (float)c->gamma, rs = (float)c->reward_scale;
        for (int b = 0; b < B; ++b) {
            const float nv = nq[b] - alpha * nlogp[b];
            tgt[b] = rs * reward[b] + ((1.0f - (float)term[b]) * gm) * nv;
        }
        acts_free(&qm, ta); free(nq);
    }
    if (probe && probe->tgt) memcpy(probe->tgt, tgt, sizeof(float) * B);
    for (int b = 0; b < B; ++b) { memcpy(xin + (size_t)b * (O + A), obs + (size_t)b * O, sizeof(float) * O); memcpy(xin + (size_t)b * (O + A) + O, act + (size_t)b * A, sizeof(float) * A); }
    double lc = 0.0;
    for (int i = 0; i < NC; ++i) {
        float** ca = acts_alloc(&qm, B);
        sac_mlp_fwd(&qm, qs[i], xin, B, ca);
        if (probe && probe->q_pred) memcpy(probe->q_pred + (size_t)i * B, ca[qm.n - 1], sizeof(float) * B);
        float* dout = (float*)malloc(sizeof(float) * B);
        double ls = 0.0;
        for (int b = 0; b < B; ++b) {
            const float d = ca[qm.n - 1][b] - tgt[b];
            if (c->critic_loss == 1) { const float zz = fabsf(d); ls += zz < 1.f ? 0.5 * (double)zz * zz : (double)zz - 0.5; dout[b] = (zz < 1.f ? d : (d > 0.f ? 1.f : -1.f)) / (float)B; }
            else { ls += (double)d * d; dout[b] = 2.f * d / (float)B; }
        }
        lc += (double)(float)(ls / B);
        float* gq = (float*)calloc((size_t)nq, sizeof(float));
        sac_mlp_bwd(&qm, qs[i], xin, B, ca, dout, gq, NULL);
        if (probe && probe->q_grads) memcpy(probe->q_grads + (size_t)i * nq, gq, sizeof(float) * (size_t)nq);
        orc_adam_step(&adam_q[i], qs[i], gq, q_m[i], q_v[i], nq);
        acts_free(&qm, ca); free(dout); free(gq);
    }
    /* ---------------- soft_update (:169-173) ---------------- */
    for (int i = 0; i < NC; ++i) orc_track(qs_tgt[i], qs[i], c->tau, nq);
    rec->loss_critic = (float)(lc / NC); rec->loss_actor = loss_actor; rec->ent_coef = expf(*log_alpha);
    free(a); free(logp); free(xin); free(na); free(nlogp); free(tgt);
}

/* ========================================================================= */
/* IQN -- border-tch-agent/src/iqn/base.rs:63-170 (update_critic),             */
/* iqn/model/base.rs:162-191 (cos embedding: cos(tau * (PI * i)), i = 1..embed */
/* INCLUSIVE, linear "iqn_cos_to_feature" + relu), :198-234 (forward: psi(x)   */
/* [B,1,F] * phi [B,N,F] -> f), util/quantile_loss.rs:7-13.                    */
/* Percent points tau are inputs (the reference draws them with Tensor::rand). */
/* psi: kind 0 AtariCnn{skip_linear} (c1..c3, flatten(C,H,W) = 3136) or kind 1 */
/* Mlp(in -> units -> feature_dim, activation_out).  f: Mlp(F -> units -> A).   */
/* Flat parameter order: psi vars, iqn_cos_to_feature.weight [F][E], .bias [F], */
/* f vars.                                                                     */
/* ========================================================================= */
typedef struct {
    int32_t psi_kind;            /* 0 cnn, 1 mlp */
    /* psi_in: the Mlp's input width; for the cnn the frame-stack depth n_stack (cnn/config.rs:14-24; 0 = the default 4) */
    int32_t psi_in, n_psi_units, psi_units[ORC_MAX_UNITS], psi_activation_out;
    int32_t feature_dim, embed_dim;
    int32_t n_f_units, f_units[ORC_MAX_UNITS];
    int32_t n_actions;
    double discount_factor;
} orc_iqn_cfg;

static int iqn_n_stack(const orc_iqn_cfg* c) { return c->psi_in > 0 ? c->psi_in : 4; }
static int64_t iqn_psi_count(const orc_iqn_cfg* c)
{
    if (c->psi_kind == 0) return 2048 * iqn_n_stack(c) + 32 + 32768 + 64 + 36864 + 64;
    return (int64_t)sac_mlp_make(c->psi_in, c->psi_units, c->n_psi_units, c->feature_dim, 1).total;
}
ORC_API int64_t orc_iqn_param_count(const orc_iqn_cfg* c)
{
    return iqn_psi_count(c) + (int64_t)c->feature_dim * c->embed_dim + c->feature_dim +
           (int64_t)sac_mlp_make(c->feature_dim, c->f_units, c->n_f_units, c->n_actions, 1).total;
}

typedef struct {
    int B, N;
    /* psi */
    float *x0, *a1, *a2, *psi;          /* cnn activations (psi = post-relu conv3, flattened C,H,W) */
    float** pacts; sac_mlp pm; const float* px;   /* mlp psi */
    float *cosv, *phi, *m;              /* [B*N][E], [B*N][F], [B*N][F] */
    float** facts; sac_mlp fm;
} iqn_cache;

static void iqn_forward(const orc_iqn_cfg* c, const float* p, const void* x, const float* tau, int B, int N, iqn_cache* k)
{
    const int F = c->feature_dim, E = c->embed_dim;
    memset(k, 0, sizeof *k);
    k->B = B; k->N = N;
    const float* q = p;
    if (c->psi_kind == 0) {
        const uint8_t* u = (const uint8_t*)x;
        const int ns = iqn_n_stack(c);
        const size_t n = (size_t)B * ns * 84 * 84, o2 = 2048 * (size_t)ns + 32, o3 = o2 + 32768 + 64;
        k->x0 = (float*)malloc(sizeof(float) * n);
        for (size_t i = 0; i < n; ++i) k->x0[i] = (float)u[i] / 255.0f;
        k->a1 = (float*)malloc(sizeof(float) * (size_t)B * 32 * 400);
        k->a2 = (float*)malloc(sizeof(float) * (size_t)B * 64 * 81);
        k->psi = (float*)malloc(sizeof(float) * (size_t)B * 3136);
        conv2d_fwd(k->x0, q, q + o2 - 32, k->a1, B, ns, 84, 84, 32, 8, 4, 1);
        conv2d_fwd(k->a1, q + o2, q + o2 + 32768, k->a2, B, 32, 20, 20, 64, 4, 2, 1);
        conv2d_fwd(k->a2, q + o3, q + o3 + 36864, k->psi, B, 64, 9, 9, 64, 3, 1, 1);
    } else {
        k->pm = sac_mlp_make(c->psi_in, c->psi_units, c->n_psi_units, c->feature_dim, 1);
        k->pm.relu[k->pm.n - 1] = c->psi_activation_out;
        k->pacts = acts_alloc(&k->pm, B);
        k->px = (const float*)x;
        sac_mlp_fwd(&k->pm, q, k->px, B, k->pacts);
        k->psi = (float*)malloc(sizeof(float) * (size_t)B * F);
        memcpy(k->psi, k->pacts[k->pm.n - 1], sizeof(float) * (size_t)B * F);
    }
    q += iqn_psi_count(c);
    const float* wc = q; const float* bc = q + (size_t)F * E;
    q = bc + F;
    const int M = B * N;
    k->cosv = (float*)malloc(sizeof(float) * (size_t)M * E);
    const float pif = (float)3.14159265358979323846;
    for (int r = 0; r < M; ++r)
        for (int i = 0; i < E; ++i) k->cosv[(size_t)r * E + i] = cosf(tau[r] * (pif * (float)(i + 1)));
    k->phi = (float*)malloc(sizeof(float) * (size_t)M * F);
    linear_fwd(k->cosv, wc, bc, k->phi, M, E, F, 1);
    k->m = (float*)malloc(sizeof(float) * (size_t)M * F);
    for (int r = 0; r < M; ++r)
        for (int j = 0; j < F; ++j) k->m[(size_t)r * F + j] = k->psi[(size_t)(r / N) * F + j] * k->phi[(size_t)r * F + j];
    k->fm = sac_mlp_make(F, c->f_units, c->n_f_units, c->n_actions, 1);
    k->facts = acts_alloc(&k->fm, M);
    sac_mlp_fwd(&k->fm, q, k->m, M, k->facts);
}

static void iqn_cache_free(iqn_cache* k)
{
    free(k->x0); free(k->a1); free(k->a2); free(k->psi); free(k->cosv); free(k->phi); free(k->m);
    if (k->pacts) acts_free(&k->pm, k->pacts);
    if (k->facts) acts_free(&k->fm, k->facts);
}

typedef struct { float* z_pred; float* z_tgt; float* tgt; float* grads; } orc_iqn_probe;

/* One Iqn::update_critic (iqn/base.rs:63-170) on a minibatch; returns the loss. */
ORC_API float orc_iqn_update(const orc_iqn_cfg* c, float* iqn, const float* iqn_tgt, orc_adam_cfg* adam, float* am, float* av,
                             int B, const void* obs, const int64_t* act, const void* next_obs, const float* reward,
                             const int8_t* term, const float* tau_pred, int n_pred, const float* tau_tgt, int n_tgt,
                             orc_iqn_probe* probe)
{
    const int A = c->n_actions, F = c->feature_dim, E = c->embed_dim;
    const int64_t np = orc_iqn_param_count(c);
    iqn_cache k, kt;
    iqn_forward(c, iqn, obs, tau_pred, B, n_pred, &k);
    iqn_forward(c, iqn_tgt, next_obs, tau_tgt, B, n_tgt, &kt);
    const float* z = k.facts[k.fm.n - 1];      /* [B*Np][A] */
    const float* zt = kt.facts[kt.fm.n - 1];   /* [B*Nt][A] */
    if (probe && probe->z_pred) memcpy(probe->z_pred, z, sizeof(float) * (size_t)B * n_pred * A);
    if (probe && probe->z_tgt) memcpy(probe->z_tgt, zt, sizeof(float) * (size_t)B * n_tgt * A);
    float* tgt = (float*)malloc(sizeof(float) * (size_t)B * n_tgt);
    const float gamma = (float)c->discount_factor;
    for (int b = 0; b < B; ++b) {
        /* a* = argmax_a mean_tau z_tgt */
        int best = 0; float bv = 0.f;
        for (int a = 0; a < A; ++a) {
            double s = 0.0;
            for (int n = 0; n < n_tgt; ++n) s += zt[((size_t)b * n_tgt + n) * A + a];
            float mv = (float)(s / n_tgt);
            if (a == 0 || mv > bv) { bv = mv; best = a; }
        }
        for (int n = 0; n < n_tgt; ++n)
            tgt[(size_t)b * n_tgt + n] = reward[b] + ((float)(1 - term[b]) * gamma) * zt[((size_t)b * n_tgt + n) * A + best];
    }
    if (probe && probe->tgt) memcpy(probe->tgt, tgt, sizeof(float) * (size_t)B * n_tgt);
    /* loss = mean_{b,n',n} |tau_p[b,n] - 1{d<0}| * huber_1(d),  d = tgt[b,n'] - pred[b,n] */
    double lsum = 0.0;
    float* dz = (float*)calloc((size_t)B * n_pred * A, sizeof(float));
    const float inv = 1.0f / ((float)B * (float)n_tgt * (float)n_pred);
    for (int b = 0; b < B; ++b)
        for (int n = 0; n < n_pred; ++n) {
            const float pred = z[((size_t)b * n_pred + n) * A + act[b]];
            const float tp = tau_pred[(size_t)b * n_pred + n];
            double g = 0.0;
            for (int m = 0; m < n_tgt; ++m) {
                const float d = tgt[(size_t)b * n_tgt + m] - pred;
                const float zabs = fabsf(d);
                const float hub = zabs < 1.f ? 0.5f * zabs * zabs : zabs - 0.5f;
                const float dh = zabs < 1.f ? d : (d > 0.f ? 1.f : -1.f);
                const float w = fabsf(tp - (d < 0.f ? 1.f : 0.f));
                lsum += (double)(w * hub);
                g += (double)(w * dh) * -1.0;   /* d(diff)/d(pred) = -1 */
            }
            dz[((size_t)b * n_pred + n) * A + act[b]] = (float)g * inv;
        }
    const float loss = (float)(lsum * (double)inv);
    /* backward */
    float* g = (float*)calloc((size_t)np, sizeof(float));
    const int64_t npsi = iqn_psi_count(c);
    float* gwc = g + npsi; float* gbc = gwc + (size_t)F * E; float* gf = gbc + F;
    const float* fparams = iqn + npsi + (size_t)F * E + F;
    const int M = B * n_pred;
    float* dm = (float*)malloc(sizeof(float) * (size_t)M * F);
    sac_mlp_bwd(&k.fm, fparams, k.m, M, k.facts, dz, gf, dm);
    float* dphi = (float*)malloc(sizeof(float) * (size_t)M * F);
    float* dpsi = (float*)calloc((size_t)B * F, sizeof(float));
    for (int r = 0; r < M; ++r)
        for (int j = 0; j < F; ++j) {
            const size_t q = (size_t)r * F + j;
            dphi[q] = k.phi[q] > 0.f ? dm[q] * k.psi[(size_t)(r / n_pred) * F + j] : 0.f;
        }
    for (int b = 0; b < B; ++b)
        for (int j = 0; j < F; ++j) {
            double s = 0.0;
            for (int n = 0; n < n_pred; ++n) { const size_t q = ((size_t)b * n_pred + n) * F + j; s += (double)dm[q] * k.phi[q]; }
            dpsi[(size_t)b * F + j] = (float)s;
        }
    linear_bwd(k.cosv, iqn + npsi, dphi, gwc, gbc, NULL, M, E, F);
    if (c->psi_kind == 0) {
        float* d3 = dpsi;   /* grad wrt post-relu conv3 output, (C,H,W) order */
        relu_bwd(k.psi, d3, (size_t)B * 3136);
        float* d2 = (float*)malloc(sizeof(float) * (size_t)B * 64 * 81);
        float* d1 = (float*)malloc(sizeof(float) * (size_t)B * 32 * 400);
        const int ns = iqn_n_stack(c);
        const size_t o2 = 2048 * (size_t)ns + 32, o3 = o2 + 32768 + 64;
        conv2d_bwd(k.a2, iqn + o3, d3, g + o3, g + o3 + 36864, d2, B, 64, 9, 9, 64, 3, 1);
        relu_bwd(k.a2, d2, (size_t)B * 64 * 81);
        conv2d_bwd(k.a1, iqn + o2, d2, g + o2, g + o2 + 32768, d1, B, 32, 20, 20, 64, 4, 2);
        relu_bwd(k.a1, d1, (size_t)B * 32 * 400);
        conv2d_bwd(k.x0, iqn, d1, g, g + o2 - 32, NULL, B, ns, 84, 84, 32, 8, 4);
        free(d2); free(d1);
    } else {
        sac_mlp_bwd(&k.pm, iqn, k.px, B, k.pacts, dpsi, g, NULL);
    }
    if (probe && probe->grads) memcpy(probe->grads, g, sizeof(float) * (size_t)np);
    orc_adam_step(adam, iqn, g, am, av, np);
    free(tgt); free(dz); free(g); free(dm); free(dphi); free(dpsi);
    iqn_cache_free(&k); iqn_cache_free(&kt);
    return loss;
}

/* ------------------------------------------------------------------------- */
/* Prioritized replay: SumTree -- generic_replay_buffer/base/sum_tree.rs       */
/* (pinned by the reference's own KATs, sum_tree.rs:180-217: see              */
/* tests/test_oracle_per.py).  f32 throughout, same operation order: the sums  */
/* in the tree are INCREMENTAL (`tree[parent] += change`, :46-52), so parity    */
/* of the sampled indices needs the same sequence of float additions.         */
/* min/max trees (segment-tree 2.0 SegmentPoint, MinIgnoreNaN / MaxIgnoreNaN)  */
/* are exact operations: plain leaf arrays + scans restate them bit for bit.   */
/* ------------------------------------------------------------------------- */
typedef struct {
    float eps, alpha;
    size_t capacity, n_samples;
    float* tree;    /* 2*capacity - 1, leaves at [capacity-1, 2*capacity-2]  (:39)  */
    float* minleaf; /* init f32::MAX  (:40) */
    float* maxleaf; /* init 1e-8      (:41) */
    int normalize;  /* 0 = All, 1 = Batch (:13-18) */
} orc_sumtree;

ORC_API orc_sumtree* orc_sumtree_new(uint64_t capacity, float alpha, int normalize)
{
    orc_sumtree* t = (orc_sumtree*)calloc(1, sizeof *t);
    t->eps = 1e-8f; t->alpha = alpha; t->capacity = capacity; t->n_samples = 0; t->normalize = normalize;
    t->tree = (float*)calloc(2 * capacity - 1, sizeof(float));
    t->minleaf = (float*)malloc(capacity * sizeof(float));
    t->maxleaf = (float*)malloc(capacity * sizeof(float));
    for (size_t i = 0; i < capacity; ++i) { t->minleaf[i] = 3.40282347e+38f; t->maxleaf[i] = 1e-8f; }
    return t;
}
ORC_API void orc_sumtree_free(orc_sumtree* t)
{
    if (!t) return;
    free(t->tree); free(t->minleaf); free(t->maxleaf); free(t);
}
ORC_API float orc_sumtree_total(const orc_sumtree* t) { return t->tree[0]; }                 /* :68-70 */
ORC_API uint64_t orc_sumtree_n_samples(const orc_sumtree* t) { return t->n_samples; }
ORC_API const float* orc_sumtree_tree(const orc_sumtree* t) { return t->tree; }
static float st_max_all(const orc_sumtree* t)
{
    float m = t->maxleaf[0];
    for (size_t i = 1; i < t->capacity; ++i) if (t->maxleaf[i] > m) m = t->maxleaf[i];
    return m;
}
static float st_min_prefix(const orc_sumtree* t, size_t n)
{
    float m = 3.40282347e+38f;   /* MinIgnoreNaN identity */
    for (size_t i = 0; i < n; ++i) if (t->minleaf[i] < m) m = t->minleaf[i];
    return m;
}
ORC_API float orc_sumtree_max(const orc_sumtree* t) { return powf(st_max_all(t), 1.0f / t->alpha); }   /* :72-76 */
ORC_API float orc_sumtree_min_p(const orc_sumtree* t) { return st_min_prefix(t, t->n_samples); }

ORC_API void orc_sumtree_update(orc_sumtree* t, uint64_t ix, float p)                           /* :92-107 */
{
    const float pa = powf(p + t->eps, t->alpha);
    t->minleaf[ix] = pa; t->maxleaf[ix] = pa;
    size_t i = ix + t->capacity - 1;
    const float change = pa - t->tree[i];
    t->tree[i] = pa;
    while (i != 0) {   /* propagate (:46-52): every ancestor += change, leaf to root */
        i = (i - 1) / 2;
        t->tree[i] += change;
    }
}
ORC_API void orc_sumtree_add(orc_sumtree* t, uint64_t ix, float p)                              /* :81-89 */
{
    orc_sumtree_update(t, ix, p);
    if (t->n_samples < t->capacity) t->n_samples += 1;
}
ORC_API uint64_t orc_sumtree_get(const orc_sumtree* t, float s)                                  /* :54-66, :110-114 */
{
    const size_t len = 2 * t->capacity - 1;
    size_t ix = 0;
    for (;;) {
        const size_t left = 2 * ix + 1, right = left + 1;
        if (left >= len) break;
        if (s <= t->tree[left] || t->tree[right] == 0.0f) ix = left;
        else { s -= t->tree[left]; ix = right; }
    }
    return ix + 1 - t->capacity;
}
/* sample (:120-157) with the batch's uniforms u[k] in [0,1) supplied by the caller (the reference calls
 * fastrand::f32(), an unseeded global generator).  Returns indices and normalised weights. */
ORC_API void orc_sumtree_sample(const orc_sumtree* t, int n, float beta, const float* u, int64_t* ixs, float* ws)
{
    const float p_sum = t->tree[0];
    for (int k = 0; k < n; ++k) ixs[k] = (int64_t)orc_sumtree_get(t, p_sum * u[k]);
    const float nn = (float)t->n_samples / p_sum;
    for (int k = 0; k < n; ++k) ws[k] = powf(nn * t->tree[ixs[k] + t->capacity - 1], -beta);
    float w_max_inv;
    if (t->normalize == 0) w_max_inv = powf(nn * st_min_prefix(t, t->n_samples), beta);
    else {
        float m = ws[0];   /* fold(NaN, |m, v| v.max(m)) == max of the batch */
        for (int k = 1; k < n; ++k) if (ws[k] > m) m = ws[k];
        w_max_inv = 1.0f / m;
    }
    for (int k = 0; k < n; ++k) ws[k] = ws[k] * w_max_inv;
}
/* IwScheduler::beta (base/iw_scheduler.rs:35-43) */
ORC_API float orc_iw_beta(float beta_0, float beta_final, uint64_t n_opts_final, uint64_t n_opts)
{
    if (n_opts >= n_opts_final) return beta_final;
    const float d = beta_final - beta_0;
    return beta_0 + d * ((float)n_opts / (float)n_opts_final);
}
