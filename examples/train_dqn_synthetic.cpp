// A complete training program above the C ABI, in compiled code only (what a border maintainer's Rust shim would do through
// FFI): synthetic Atari-shaped environment -> Trainer::train (bdr_trainer_train) -> DQN Nature-CNN agent
// and HBM replay buffer.
// Build (python border_amd/build.py does it):
//     g++ -O2 -std=c++17 examples/train_dqn_synthetic.cpp -Iinclude -Lborder_amd -lborder_amd -o examples/train_dqn_synthetic
// Run:    examples/train_dqn_synthetic [max_opts] [device]
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

// Env stand-in (border-core/src/base/env.rs): random 84x84x4 u8 frames, episode ends with probability 1/64
struct SyntheticEnv {
    uint64_t s = 0x1234;
    uint64_t next() { s += 0x9E3779B97F4A7C15ull; uint64_t x = s; x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; return x ^ (x >> 31); }
    void fill(uint8_t* obs) { for (int i = 0; i < 28224; i += 8) { const uint64_t v = next(); std::memcpy(obs + i, &v, 8); } }
    static int32_t reset(void* ctx, void* obs_out) { ((SyntheticEnv*)ctx)->fill((uint8_t*)obs_out); return BDR_OK; }
    static int32_t step(void* ctx, const void* act, void* obs_out, float* reward, int8_t* term, int8_t* trunc, void* init_obs_out)
    {
        SyntheticEnv* e = (SyntheticEnv*)ctx;
        (void)act;
        e->fill((uint8_t*)obs_out);
        const uint64_t r = e->next();
        *reward = (r & 31) == 0 ? 1.0f : ((r & 31) == 1 ? -1.0f : 0.0f);
        *term = ((r >> 8) & 63) == 0; *trunc = 0;
        if (*term) e->fill((uint8_t*)init_obs_out);
        return BDR_OK;
    }
};

static void observe(void*, uint64_t env_steps, uint64_t opt_steps, int32_t event, const float* scalars, int32_t n)
{
    if (event == BDR_TRAINER_EVENT_OPT_RECORD && n >= 1) std::printf("env_steps %llu opt_steps %llu loss %.6f\n", (unsigned long long)env_steps, (unsigned long long)opt_steps, scalars[0]);
    if (event == BDR_TRAINER_EVENT_COST) std::printf("opt_steps %llu average_opt_time %.3f ms average_sample_time %.3f ms\n", (unsigned long long)opt_steps, scalars[0], scalars[1]);
}

int main(int argc, char** argv)
{
    const uint64_t max_opts = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 200;
    const int device = argc > 2 ? std::atoi(argv[2]) : 0;

    bdr_replay_config rc{};
    rc.capacity = 10000; rc.seed = 42; rc.obs_row_bytes = 28224; rc.act_row_bytes = 8; rc.device = device;
    bdr_replay* rb = nullptr;
    CHECK(bdr_replay_create(&rc, &rb));

    bdr_dqn_config dc;
    bdr_dqn_config_default(&dc);
    dc.net.kind = BDR_NET_ATARI_CNN; dc.net.n_stack = 4; dc.net.out_dim = 6;
    dc.lr = 1e-4; dc.batch_size = 32; dc.critic_loss = BDR_LOSS_SMOOTH_L1; dc.tau = 1.0; dc.soft_update_interval = 100; dc.device = device;
    bdr_agent* agent = nullptr;
    CHECK(bdr_dqn_create(&dc, &agent));
    bdr_explorer_config ec;
    bdr_explorer_config_default(&ec, BDR_EXPLORER_EPS_GREEDY);
    ec.final_step = 1000; ec.seed = 7;
    CHECK(bdr_agent_set_explorer(agent, &ec));

    SyntheticEnv env;
    bdr_env_vtable vt{&env, SyntheticEnv::reset, SyntheticEnv::step};
    bdr_trainer_ops ops;
    bdr_trainer_ops_default(&ops, agent, rb);
    bdr_trainer_config tc;
    bdr_trainer_config_default(&tc);
    tc.max_opts = max_opts; tc.opt_interval = 1; tc.warmup_period = 64; tc.record_agent_info_interval = 50; tc.record_compute_cost_interval = 100;
    tc.obs_row_bytes = 28224; tc.act_row_bytes = 8;
    bdr_trainer_stats st{};
    CHECK(bdr_trainer_train(&tc, &ops, &vt, observe, nullptr, &st));
    CHECK(bdr_agent_sync(agent));
    uint64_t len = 0, n_opts = 0;
    CHECK(bdr_replay_len(rb, &len));
    CHECK(bdr_agent_n_opts(agent, &n_opts));
    std::printf("done: env_steps %llu opt_steps %llu episodes %llu buffer_len %llu n_opts %llu opt %.3f s sample %.3f s\n",
                (unsigned long long)st.env_steps, (unsigned long long)st.opt_steps, (unsigned long long)st.n_episodes, (unsigned long long)len,
                (unsigned long long)n_opts, st.opt_seconds, st.sample_seconds);
    CHECK(bdr_agent_destroy(agent));
    CHECK(bdr_replay_destroy(rb));
    return 0;
}
