/* BASELINE configs[1] — Breakout-shaped dqn, 256 vectorized envs x 80 steps + 20 updates of batch 512 per iteration — configs[2] — c51 rainbow-lite: NoisyNet,
 * dueling, double-Q, 3-step returns, prioritized replay — configs[3] — Asterix-shaped (9 actions) implicit quantile network — or configs[4]'s per-GPU workload — fqf on 9 actions — run by a host that is NOT Python: plain C against include/agent0_hip.h, the three handles
 * a0_actor / a0_rbuf / a0_learner (library-owned HBM) and the loop of trainer.py:74-119,171-184 written out.
 * usage: c_host_loop [iterations] [replay_size] [env_task 0|1] [config 1|2|3|4] [gradient exchange 0|1]     prints one JSON line (tests/test_gpu_trainer.py compiles and runs it on the GPU box). */
#include <hip/hip_runtime_api.h>      /* gcc -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "agent0_hip.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, a0_last_error()); return 1; } } while (0)
#define HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }
static unsigned lcg(unsigned* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }

/* position-weighted word sums (mod 2^64) of a device buffer: the fingerprint tests/test_gpu_trainer.py recomputes with numpy from the Python run of the same seed */
static int fingerprint(const void* dev, size_t bytes, unsigned long long* s1, unsigned long long* s2, unsigned long long* pos) {
    unsigned* h = (unsigned*)malloc(bytes);
    if (!h || hipMemcpy(h, dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) { free(h); return 1; }
    for (size_t i = 0; i < bytes / 4; ++i) { *s1 += h[i]; *s2 += (unsigned long long)(++*pos) * h[i]; }
    free(h);
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const long long size = argc > 2 ? atoll(argv[2]) : 100000;
    const int task = argc > 3 ? atoi(argv[3]) : A0_ENV_TASK_STREAM;
    const int config = argc > 4 ? atoi(argv[4]) : 1;
    const int exchange = argc > 5 ? atoi(argv[5]) : 0;      /* 1: data-parallel form with a ONE-rank RCCL communicator (a rehearsal: the sums are the identity) */
    const int E = 256, T = 80, B = 512, LSTEPS = 20, A = config >= 3 ? 9 : 4, OBS = 4 * 84 * 84;
    const long long start_steps = size < 100000 ? size / 2 : 100000, exploration = 1000000;
    const double min_eps = 0.01;
    const int rainbow = config == 2;
    a0_learner_desc ld = {A, rainbow, rainbow, B, rainbow ? 3 : 1, 0.99, 5e-4, 0.0, 500, rainbow ? A0_ALGO_C51 : (config == 3 ? A0_ALGO_IQN : (config == 4 ? A0_ALGO_FQF : A0_ALGO_DQN)), 51, -10.0, 10.0,
                          rainbow, 42 + 15485863, /* iqn K, N, N' */ 32, 64, 64, /* fqf F */ 32};
    a0_rbuf_desc rd = {size, OBS, B, rainbow, 0.5, 0.01, 0.4, 10000000, 42 + 104729};
    a0_actor_desc ad = {E, T, A, rainbow, rainbow ? 3 : 1, 0.99, 42, 0, task, 4};
    a0_learner* L = NULL; a0_rbuf* R = NULL; a0_actor* ac = NULL;
    CHECK(a0_learner_create(&ld, &L)); CHECK(a0_rbuf_create(&rd, &R)); CHECK(a0_actor_create(&ad, &ac));
    CHECK(a0_actor_bind(ac, L, 0));      /* the rollout's workspaces now: nothing below allocates or synchronises the device */
    long long comm = 0;
    if (exchange) {
        /* one process per GPU; rank 0 creates the 128-byte id and hands it to the other ranks over whatever transport the job has (MPI, a file, a socket), every rank
         * then joins with its rank and the world size.  From here on a0_learner_update sums the gradients over the ranks between backward and Adam. */
        char id[128];
        CHECK(a0_dp_unique_id(id));
        comm = a0_dp_init(id, 0, 1);
        if (!comm) { fprintf(stderr, "a0_dp_init failed: %s\n", a0_last_error()); return 1; }
        CHECK(a0_learner_set_exchange(L, comm));
    }
    /* small random initial weights (a real host would load a packed checkpoint: agent0_amd/deepq/layout.py) */
    const long long n = a0_learner_param_floats(L);
    float* hp = (float*)malloc((size_t)n * sizeof(float));
    unsigned seed = 7u;
    for (long long i = 0; i < n; ++i) hp[i] = ((float)(lcg(&seed) % 2001) - 1000.0f) * 2e-5f;
    float* dp = NULL; float* dloss = NULL;
    HIP(hipMalloc((void**)&dp, (size_t)n * 4));
    CHECK(a0_learner_loss_buffer(L, &dloss));
    HIP(hipMemcpy(dp, hp, (size_t)n * 4, hipMemcpyHostToDevice));
    CHECK(a0_learner_set_params(L, dp, NULL, NULL));
    uint8_t* ring = NULL;
    CHECK(a0_rbuf_buffers(R, &ring, NULL, NULL, NULL, NULL, NULL));
    float* qs = (float*)malloc(T * sizeof(float)); float* rets = (float*)malloc((size_t)T * E * sizeof(float));
    long long frames = 0, updates = 0, episodes = 0;
    double ret_sum = 0.0, t0 = 0.0, qmax = 0.0;
    int timed = 0;
    for (int it = 0; timed < iters; ++it) {
        const double eps = frames > exploration ? min_eps : (1.0 - (double)frames / (double)exploration) + min_eps;      /* trainer.py:46-50 */
        CHECK(a0_actor_rollout(ac, L, R, (float)eps, NULL));
        CHECK(a0_rbuf_commit(R, (long long)T * E, NULL));
        frames += (long long)T * E;
        const int training = a0_rbuf_len(R) > start_steps;
        if (training) {
            if (timed == 0 && t0 == 0.0) { HIP(hipDeviceSynchronize()); t0 = now(); }
            a0_batch blk[32];
            if (!rainbow) CHECK(a0_rbuf_sample_block(R, LSTEPS, blk, NULL));      /* uniform replay: the block's 20 batches in one launch */
            for (int u = 0; u < LSTEPS; ++u) {
                a0_batch b;
                if (rainbow) CHECK(a0_rbuf_sample(R, &b, NULL)); else b = blk[u];
                CHECK(a0_learner_update(L, ring, b.slot, 2LL * OBS, b.act, b.rew, b.done, b.weights, NULL, NULL));
                CHECK(a0_rbuf_update_priority(R, dloss, NULL, NULL));      /* the learner's own loss buffer: no copy */
                ++updates;
            }
        }
        int nret = 0;
        CHECK(a0_actor_collect(ac, qs, rets, T * E, &nret, NULL));      /* the one device -> host read of the iteration */
        for (int k = 0; k < nret; ++k) { ret_sum += rets[k]; ++episodes; }
        qmax = qs[T - 1];
        if (training) ++timed;
    }
    HIP(hipDeviceSynchronize());
    const double dt = now() - t0;
    float hl[512];
    HIP(hipMemcpy(hl, dloss, B * 4, hipMemcpyDeviceToHost));
    double mean = 0.0; int finite = 1;
    for (int b = 0; b < B; ++b) { mean += hl[b] / B; if (!isfinite(hl[b])) finite = 0; }
    printf("{\"host\": \"plain C (tests/c_host_loop.c)\", \"iterations_timed\": %d, \"ms_per_iteration\": %.3f, \"env_frames_per_sec\": %.1f, \"frames\": %lld, \"updates\": %lld, "
           "\"episodes\": %lld, \"mean_return\": %.4f, \"last_mean_loss\": %.6g, \"last_qmax\": %.5g, \"finite\": %d, \"replay_size\": %lld, \"env_task\": %d, "
           "\"gradient_exchange\": \"%s\", \"config\": \"%s\"}\n",
           timed, 1e3 * dt / timed, (double)timed * T * E / dt, frames, updates, episodes, episodes ? ret_sum / (double)episodes : 0.0, mean, qmax, finite, size, task,
           exchange ? "a0_learner_set_exchange, one-rank RCCL communicator" : "none",
           rainbow ? "BASELINE configs[2]: c51 + NoisyNet + dueling + double-Q + 3-step + prioritized replay"
                   : (config == 3 ? "BASELINE configs[3]: iqn, 9 actions, K = 32, N = N' = 64, uniform replay"
                                  : (config == 4 ? "BASELINE configs[4] (one GPU's share): fqf, 9 actions, F = 32, uniform replay" : "BASELINE configs[1]: dqn, uniform replay")));
    {   /* what the run left behind: parameters, target, Adam moments; the ring's actions / n-step rewards / dones and every 997th row's st||st_next bytes */
        unsigned long long p1 = 0, p2 = 0, pp = 0, r1 = 0, r2 = 0, rp = 0;
        float* tmp = NULL;
        HIP(hipMalloc((void**)&tmp, (size_t)n * 4));
        for (int k = 0; k < 4; ++k) {
            CHECK(a0_learner_get(L, k == 0 ? tmp : NULL, k == 1 ? tmp : NULL, k == 2 ? tmp : NULL, k == 3 ? tmp : NULL, NULL, NULL));
            HIP(hipDeviceSynchronize());
            if (fingerprint(tmp, (size_t)n * 4, &p1, &p2, &pp)) return 1;
        }
        HIP(hipFree(tmp));
        int* r_act = NULL; float *r_rew = NULL, *r_done = NULL;
        CHECK(a0_rbuf_buffers(R, NULL, &r_act, &r_rew, &r_done, NULL, NULL));
        if (fingerprint(r_act, (size_t)size * 4, &r1, &r2, &rp) || fingerprint(r_rew, (size_t)size * 4, &r1, &r2, &rp) || fingerprint(r_done, (size_t)size * 4, &r1, &r2, &rp)) return 1;
        for (long long row = 0; row < size; row += 997)
            if (fingerprint(ring + row * 2LL * OBS, (size_t)2 * OBS, &r1, &r2, &rp)) return 1;
        printf("{\"fingerprint\": {\"learner\": [%llu, %llu], \"ring\": [%llu, %llu]}}\n", p1, p2, r1, r2);
    }
    if (comm) { CHECK(a0_learner_set_exchange(L, 0)); CHECK(a0_dp_destroy(comm)); }
    CHECK(a0_actor_destroy(ac)); CHECK(a0_rbuf_destroy(R)); CHECK(a0_learner_destroy(L));
    return 0;
}
