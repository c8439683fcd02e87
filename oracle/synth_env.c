/* CPU oracle: synthetic Atari-shaped vector environment — TEST INFRASTRUCTURE.
 *
 * Stands in for make_atari (reference agent0/common/atari_wrappers.py:59-69, gymnasium 0.28.1 + ale-py, neither
 * present on the GPU box; SURVEY.md §8(d) "Synthetic inputs").  It honours the tuple/info contract that
 * Actor.sample consumes (agent0/deepq/agent.py:55-62,85-88): obs (E,4,84,84) u8 frame stack, reward in {-1,0,1},
 * terminal, truncated (never), info["life_loss"], and an episode-return record on terminal with the observation
 * already auto-reset.  It is NOT Atari; it exists so throughput can be measured with inputs of the right shape.
 * The device twin is agent0_amd/csrc/synth_env.hip and must match this file byte-for-byte.
 *
 *   frame(e,g)[y][x] = lit ? (h & 255) : 0,   h = mix(seed ^ mix(e*0x9E3779B1 + g) ^ ((y*84+x)*0x85EBCA77)),
 *                      lit = ((h >> 8) & 3) == 0;  an 8x8 block of 255 at (by,bx) = ((3g+11e)%77, (5g+7e)%77)
 *   draws (x0..x3) = philox4x32_10(ctr = (e, g, 0, 0x454E56), key = (seed_lo, seed_hi ^ rank))
 *   reward   = task 0 ("stream"): x0%1000 < 50 ? -1 : x0%1000 < 100 ? +1 : 0     P = (0.05, 0.05, 0.90), independent of the action (bench workload)
 *              task 1 ("block"):  the newest frame of the observation the action was chosen on is frame(e, g-1); its block sits at
 *                                 (by, bx) = ((3(g-1)+11e)%77, (5(g-1)+7e)%77); target = (2[by >= 39] + [bx >= 39]) % A (the block's quadrant);
 *                                 +1 for action == target, -1 for action == (target+1) % A, else 0: chance level 0, optimum +1 per step.
 *                                 Exists so that LEARNING can be tested (the reference's acceptance evidence is learning curves, README.md:62-112).
 *              task 2 ("chase"):  TEMPORAL credit.  The block lives on a 4 x 4 lattice (cell c = 4 cy + cx, top-left pixel (4 + 22 cy, 4 + 22 cx)) and the action moves
 *                                 it: a % 4 = 0 up, 1 down, 2 left, 3 right, clamped at the walls.  The current cell is read back from the newest frame of the
 *                                 observation (probe pixel (7 + 22 cy, 7 + 22 cx) == 255; the background is clamped to 254 under this task; start cell
 *                                 (7 e + 3) % 15).  +1 only on arrival at cell 15, then a respawn at the (x3 % 10)-th cell with cy + cx <= 3 (Manhattan distance
 *                                 >= 3); 0 otherwise.  frame(e, g) shows the block AFTER the move / respawn of step g.  Optimum: one reward per 4.0 steps.
 *   terminal = x1 % 500 == 0;  life_loss = !terminal && x2 % 200 == 0;  truncated = 0
 *   obs'     = terminal ? 4 x frame(e,g) : shift(obs) + frame(e,g)
 */
#include <stdint.h>
#include <string.h>

void a0o_philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]);

#define A0O_H 84
#define A0O_W 84
#define A0O_PIX (A0O_H * A0O_W)

static uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

/* chase_cell >= 0: the frame of the chase task — block at that lattice cell, background clamped below 255 */
static void env_frame_at(uint32_t seed, uint32_t e, uint32_t g, int chase_cell, uint8_t* out /* [84*84] */) {
    uint32_t base = seed ^ mix32(e * 0x9E3779B1u + g);
    uint32_t by = (3u * g + 11u * e) % 77u, bx = (5u * g + 7u * e) % 77u;
    if (chase_cell >= 0) { by = 4u + 22u * (uint32_t)(chase_cell >> 2); bx = 4u + 22u * (uint32_t)(chase_cell & 3); }
    for (uint32_t y = 0; y < A0O_H; ++y)
        for (uint32_t x = 0; x < A0O_W; ++x) {
            uint32_t h = mix32(base ^ ((y * A0O_W + x) * 0x85EBCA77u));
            uint8_t v = (((h >> 8) & 3u) == 0u) ? (uint8_t)(h & 255u) : 0;
            if (chase_cell >= 0 && v == 255) v = 254;
            if (y >= by && y < by + 8 && x >= bx && x < bx + 8) v = 255;
            out[y * A0O_W + x] = v;
        }
}

void a0o_env_frame(uint32_t seed, uint32_t e, uint32_t g, uint8_t* out /* [84*84] */) { env_frame_at(seed, e, g, -1, out); }

static int chase_start_cell(uint32_t e) { return (int)((7u * e + 3u) % 15u); }

/* the block's cell as the newest frame shows it */
int32_t a0o_env_chase_cell(const uint8_t* newest, uint32_t e) {
    for (int c = 0; c < 16; ++c)
        if (newest[(7 + 22 * (c >> 2)) * A0O_W + 7 + 22 * (c & 3)] == 255) return c;
    return chase_start_cell(e);
}

static int chase_step(int cell, int a, uint32_t x3, float* reward) {
    int cy = cell >> 2, cx = cell & 3, m = a & 3;
    if (m == 0) cy = cy > 0 ? cy - 1 : 0;
    else if (m == 1) cy = cy < 3 ? cy + 1 : 3;
    else if (m == 2) cx = cx > 0 ? cx - 1 : 0;
    else cx = cx < 3 ? cx + 1 : 3;
    int nc = 4 * cy + cx;
    *reward = 0.0f;
    if (nc == 15) {
        static const int spawn[10] = {0, 1, 2, 3, 4, 5, 6, 8, 9, 12};
        *reward = 1.0f;
        nc = spawn[x3 % 10u];
    }
    return nc;
}

/* envs e0 .. e0+E-1 of the vector env (a worker process of a host env pool owns such a slice); arrays are indexed from 0 */
void a0o_env_reset_task_at(uint64_t seed, uint32_t rank, int64_t e0, int64_t E, int32_t task, uint32_t* g, float* ep_ret, uint8_t* obs /* [E,4,84,84] */) {
    (void)rank;
    for (int64_t e = 0; e < E; ++e) {
        g[e] = 0; ep_ret[e] = 0.0f;
        uint8_t* o = obs + (size_t)e * 4 * A0O_PIX;
        env_frame_at((uint32_t)seed, (uint32_t)(e0 + e), 0, task == 2 ? chase_start_cell((uint32_t)(e0 + e)) : -1, o);
        for (int c = 1; c < 4; ++c) memcpy(o + c * A0O_PIX, o, A0O_PIX);
    }
}

void a0o_env_reset_at(uint64_t seed, uint32_t rank, int64_t e0, int64_t E, uint32_t* g, float* ep_ret, uint8_t* obs /* [E,4,84,84] */) {
    a0o_env_reset_task_at(seed, rank, e0, E, 0, g, ep_ret, obs);
}

void a0o_env_reset(uint64_t seed, uint32_t rank, int64_t E, uint32_t* g, float* ep_ret, uint8_t* obs) {
    a0o_env_reset_at(seed, rank, 0, E, g, ep_ret, obs);
}

float a0o_env_reward(uint32_t x0, int32_t task, int32_t A, uint32_t e, uint32_t g, int32_t a) {
    if (task == 1) {
        uint32_t gp = g - 1u;
        uint32_t by = (3u * gp + 11u * e) % 77u, bx = (5u * gp + 7u * e) % 77u;
        int32_t target = (int32_t)((2u * (by >= 39u ? 1u : 0u) + (bx >= 39u ? 1u : 0u)) % (uint32_t)A);
        int32_t wrong = (target + 1) % A;
        return a == target ? 1.0f : (a == wrong ? -1.0f : 0.0f);
    }
    uint32_t rw = x0 % 1000u;
    return rw < 50u ? -1.0f : (rw < 100u ? 1.0f : 0.0f);
}

void a0o_env_step_task_at(uint64_t seed, uint32_t rank, int64_t e0, int64_t E, const int32_t* action, int32_t A, int32_t task, uint32_t* g, float* ep_ret,
                     const uint8_t* obs_in, uint8_t* obs_out, float* reward, uint8_t* terminal, uint8_t* truncated,
                     uint8_t* life_loss, uint8_t* final_mask, float* final_ret) {
    for (int64_t e = 0; e < E; ++e) {
        uint32_t gg = g[e] + 1u;
        g[e] = gg;
        uint32_t ctr[4] = {(uint32_t)(e0 + e), gg, 0u, 0x454E56u};
        uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32) ^ rank};
        uint32_t x[4];
        a0o_philox4x32_10(ctr, key, x);
        const uint8_t* in = obs_in + (size_t)e * 4 * A0O_PIX;
        int chase_cell = -1;
        float r;
        if (task == 2) chase_cell = chase_step(a0o_env_chase_cell(in + 3 * A0O_PIX, (uint32_t)(e0 + e)), action[e], x[3], &r);
        else r = a0o_env_reward(x[0], task, A, (uint32_t)(e0 + e), gg, task == 1 ? action[e] : 0);
        uint8_t term = (x[1] % 500u) == 0u;
        uint8_t life = (!term) && ((x[2] % 200u) == 0u);
        reward[e] = r; terminal[e] = term; truncated[e] = 0; life_loss[e] = life;
        ep_ret[e] += r;
        final_mask[e] = term; final_ret[e] = term ? ep_ret[e] : 0.0f;
        if (term) ep_ret[e] = 0.0f;
        uint8_t* out = obs_out + (size_t)e * 4 * A0O_PIX;
        uint8_t fr[A0O_PIX];
        env_frame_at((uint32_t)seed, (uint32_t)(e0 + e), gg, chase_cell, fr);
        if (term) {
            for (int c = 0; c < 4; ++c) memcpy(out + c * A0O_PIX, fr, A0O_PIX);
        } else {
            memmove(out, in + A0O_PIX, 3 * A0O_PIX);
            memcpy(out + 3 * A0O_PIX, fr, A0O_PIX);
        }
    }
}

/* terminal[(g - 1) * E + e] = does env e0 + e terminate at step g, g = 1 .. steps (terminals do not depend on actions): lets a test turn the episode returns
 * a training run reports (in step-major, env-major order) into per-step rewards */
void a0o_env_terminals(uint64_t seed, uint32_t rank, int64_t e0, int64_t E, int64_t steps, uint8_t* terminal) {
    for (int64_t g = 1; g <= steps; ++g)
        for (int64_t e = 0; e < E; ++e) {
            uint32_t ctr[4] = {(uint32_t)(e0 + e), (uint32_t)g, 0u, 0x454E56u};
            uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32) ^ rank};
            uint32_t x[4];
            a0o_philox4x32_10(ctr, key, x);
            terminal[(g - 1) * E + e] = (x[1] % 500u) == 0u;
        }
}

void a0o_env_step_at(uint64_t seed, uint32_t rank, int64_t e0, int64_t E, const int32_t* action, uint32_t* g, float* ep_ret,
                     const uint8_t* obs_in, uint8_t* obs_out, float* reward, uint8_t* terminal, uint8_t* truncated,
                     uint8_t* life_loss, uint8_t* final_mask, float* final_ret) {
    a0o_env_step_task_at(seed, rank, e0, E, action, 1, 0, g, ep_ret, obs_in, obs_out, reward, terminal, truncated, life_loss, final_mask, final_ret);
}

void a0o_env_step(uint64_t seed, uint32_t rank, int64_t E, const int32_t* action, uint32_t* g, float* ep_ret,
                  const uint8_t* obs_in, uint8_t* obs_out, float* reward, uint8_t* terminal, uint8_t* truncated,
                  uint8_t* life_loss, uint8_t* final_mask, float* final_ret) {
    a0o_env_step_at(seed, rank, 0, E, action, g, ep_ret, obs_in, obs_out, reward, terminal, truncated, life_loss, final_mask, final_ret);
}
