// The one-environment online loop of Trainer::train (trainer.rs:267-327) in compiled code, with the observation resident in HBM:
// emulator frames (a pre-rendered pool standing in for render_rgb24) -> bdr_atari_prep (skip_and_max, warp_and_grayscale,
// stack_frame on the device, border-atari-env/src/env.rs:126-209) -> bdr_env_vtable with obs_on_device = 1 -> bdr_trainer_train
// (sample_device, push_device, opt at batch_size).  It prints the loop rate between the first and the last opt step: what the
// loop costs when neither Python nor ctypes is in it.
// Build (python border_amd/build.py does it):
//     g++ -O2 -std=c++17 examples/online_loop_atari.cpp -Iinclude -Lborder_amd -lborder_amd -o examples/online_loop_atari
// Run:    examples/online_loop_atari [max_opts] [batch_size] [device] [obs: device|host]
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "border_amd.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        const int32_t rc_ = (call);                                                  \
        if (rc_ != BDR_OK) {                                                         \
            std::fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, bdr_last_error()); \
            return 1;                                                                \
        }                                                                            \
    } while (0)

static const uint32_t W = 160, H = 210, FRAME = W * H * 3, OBS = 4 * 84 * 84;

// Emulator stand-in: 64 pre-rendered RGB frames; a step hands the two last frames of its skip-4 window, a reward and the flags
struct Emulator {
    std::vector<uint8_t> pool;
    uint64_t s = 0x51ED;
    uint64_t next() { s += 0x9E3779B97F4A7C15ull; uint64_t x = s; x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; return x ^ (x >> 31); }
    Emulator() : pool(64ull * FRAME) { for (size_t i = 0; i < pool.size(); i += 8) { const uint64_t v = next(); std::memcpy(&pool[i], &v, 8); } }
    const uint8_t* frame() { return &pool[(next() & 63) * FRAME]; }
};

struct AtariEnv {
    Emulator emu;
    bdr_atari_prep* prep = nullptr;
    bool on_device = true;
    const uint32_t ix = 0;
    int32_t write_stack(void* out) { return on_device ? bdr_atari_prep_copy_stack(prep, 0, out) : bdr_atari_prep_obs(prep, 1, &ix, (uint8_t*)out); }
    static int32_t reset(void* ctx, void* obs_out)
    {
        AtariEnv* e = (AtariEnv*)ctx;
        const int32_t rc = bdr_atari_prep_reset(e->prep, 1, &e->ix, e->emu.frame());
        return rc != BDR_OK ? rc : e->write_stack(obs_out);
    }
    static int32_t step(void* ctx, const void* act, void* obs_out, float* reward, int8_t* term, int8_t* trunc, void* init_obs_out)
    {
        AtariEnv* e = (AtariEnv*)ctx;
        (void)act;
        const uint8_t* fa = e->emu.frame();
        const uint8_t* fb = e->emu.frame();
        int32_t rc = bdr_atari_prep_step(e->prep, 1, &e->ix, fa, fb);
        if (rc != BDR_OK) return rc;
        if ((rc = e->write_stack(obs_out)) != BDR_OK) return rc;
        const uint64_t r = e->emu.next();
        *reward = bdr_atari_clip_reward((r & 31) == 0 ? 7.0f : ((r & 31) == 1 ? -3.0f : 0.0f), 1);
        *term = ((r >> 8) & 511) == 0; *trunc = 0;
        if (*term) {
            if ((rc = bdr_atari_prep_reset(e->prep, 1, &e->ix, e->emu.frame())) != BDR_OK) return rc;
            rc = e->write_stack(init_obs_out);
        }
        return rc;
    }
};

struct Clock {
    std::chrono::steady_clock::time_point first, last;
    uint64_t env_first = 0, env_last = 0, opt_first = 0, opt_last = 0;
    bool started = false;
};

static void observe(void* ctx, uint64_t env_steps, uint64_t opt_steps, int32_t event, const float*, int32_t)
{
    if (event != BDR_TRAINER_EVENT_OPT && event != BDR_TRAINER_EVENT_OPT_RECORD) return;
    Clock* c = (Clock*)ctx;
    const auto now = std::chrono::steady_clock::now();
    if (!c->started && opt_steps >= 64) { c->started = true; c->first = now; c->env_first = env_steps; c->opt_first = opt_steps; }
    c->last = now; c->env_last = env_steps; c->opt_last = opt_steps;
}

int main(int argc, char** argv)
{
    const uint64_t max_opts = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 2000;
    const uint32_t batch = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 256;
    const int device = argc > 3 ? std::atoi(argv[3]) : 0;
    const bool on_device = !(argc > 4 && std::strcmp(argv[4], "host") == 0);

    bdr_replay_config rc{};
    rc.capacity = 65536; rc.seed = 42; rc.obs_row_bytes = OBS; rc.act_row_bytes = 8; rc.device = device;
    bdr_replay* rb = nullptr;
    CHECK(bdr_replay_create(&rc, &rb));

    bdr_dqn_config dc;
    bdr_dqn_config_default(&dc);
    dc.net.kind = BDR_NET_ATARI_CNN; dc.net.n_stack = 4; dc.net.out_dim = 6;
    dc.lr = 1e-4; dc.batch_size = batch; dc.critic_loss = BDR_LOSS_SMOOTH_L1; dc.tau = 1.0; dc.soft_update_interval = 1000; dc.device = device;
    bdr_agent* agent = nullptr;
    CHECK(bdr_dqn_create(&dc, &agent));
    bdr_explorer_config ec;
    bdr_explorer_config_default(&ec, BDR_EXPLORER_EPS_GREEDY);
    ec.final_step = 100000; ec.seed = 7;
    CHECK(bdr_agent_set_explorer(agent, &ec));

    AtariEnv env;
    env.on_device = on_device;
    CHECK(bdr_atari_prep_create(device, 1, W, H, &env.prep));
    bdr_env_vtable vt{};
    vt.ctx = &env; vt.reset = AtariEnv::reset; vt.step_with_reset = AtariEnv::step; vt.obs_on_device = on_device ? 1 : 0; vt.device = device;
    bdr_trainer_ops ops;
    bdr_trainer_ops_default(&ops, agent, rb);
    bdr_trainer_config tc;
    bdr_trainer_config_default(&tc);
    tc.max_opts = max_opts; tc.opt_interval = 1; tc.warmup_period = batch < 512 ? 512 : batch; tc.record_agent_info_interval = 0; tc.record_compute_cost_interval = 0;
    tc.obs_row_bytes = OBS; tc.act_row_bytes = 8;
    bdr_trainer_stats st{};
    Clock clk;
    CHECK(bdr_trainer_train(&tc, &ops, &vt, observe, &clk, &st));
    CHECK(bdr_agent_sync(agent));
    const auto end = std::chrono::steady_clock::now();
    uint64_t len = 0, n_opts = 0;
    CHECK(bdr_replay_len(rb, &len));
    CHECK(bdr_agent_n_opts(agent, &n_opts));
    const double secs = std::chrono::duration<double>(end - clk.first).count();
    const uint64_t its = clk.env_last - clk.env_first;
    std::printf("online loop, observations on the %s, batch %u: %llu iterations (env step + preprocessing + sample + push + opt) in %.3f s = %.1f it/s"
                " (%.1f us per iteration; opt %.1f us of it by the trainer's timer)\n",
                on_device ? "device" : "host", batch, (unsigned long long)its, secs, clk.started && secs > 0 ? its / secs : 0.0,
                its ? 1e6 * secs / its : 0.0, st.opt_steps ? 1e6 * st.opt_seconds / st.opt_steps : 0.0);
    std::printf("done: env_steps %llu opt_steps %llu episodes %llu buffer_len %llu n_opts %llu\n", (unsigned long long)st.env_steps,
                (unsigned long long)st.opt_steps, (unsigned long long)st.n_episodes, (unsigned long long)len, (unsigned long long)n_opts);
    CHECK(bdr_atari_prep_destroy(env.prep));
    CHECK(bdr_agent_destroy(agent));
    CHECK(bdr_replay_destroy(rb));
    return n_opts == max_opts ? 0 : 2;
}
