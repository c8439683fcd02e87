/* agent0_hip.h — C-ABI of libagent0_hip.so: the MI355X (gfx950) hot path of agent0's deepq actor-learner loop.
 *
 * The reference (zhoubin-me/agent0) is pure Python and defines no FFI; each entry point below names the reference
 * code it replaces (paths relative to the reference repo).  INTEGRATION.md shows the ctypes binding a maintainer
 * of the reference would add.  Conventions:
 *   - every function returns int: 0 = A0_OK, < 0 = error (A0_E*), message via a0_last_error() (thread-local);
 *   - all pointers are DEVICE pointers unless a parameter is named host_*; buffers are owned by the caller and
 *     only borrowed for the duration of the call (the library keeps no reference, a0_net tables excepted);
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous w.r.t. the host, never synchronise,
 *     never allocate (except a0_net_create) and are safe to capture into a hipGraph;
 *   - fp32 everywhere the reference is fp32; u8 frames; int32 actions/slots; int64 replay indices.
 */
#ifndef AGENT0_HIP_H
#define AGENT0_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A0_ABI_VERSION 1

/* Control block (device int64[A0_CTRL_WORDS], optional): lets a sequence of launches be captured ONCE into a hipGraph and replayed
 * while counters keep advancing — each kernel adds ctrl[index] to the corresponding immediate argument.  NULL = no adjustment. */
#define A0_CTRL_ENV_STEP 0      /* a0_env_synth_step: g            */
#define A0_CTRL_ACTOR_STEPS 1   /* a0_actor_nstep: steps           */
#define A0_CTRL_RNG_ACTION 2    /* a0_actor_egreedy_rng: off_a     */
#define A0_CTRL_RNG_UNIFORM 3   /* a0_actor_egreedy_rng: off_u     */
#define A0_CTRL_REPLAY_SLOT 4   /* a0_replay_insert: start_slot    */
#define A0_CTRL_RNG_TAUS 5      /* a0_rng_uniform_ctrl (IQN taus)  */
#define A0_CTRL_RNG_NOISE 6     /* a0_rng_normal_ctrl (NoisyNet)   */
#define A0_CTRL_WORDS 8

const char* a0_last_error(void);
int a0_abi_version(void);
/* "default" for the product build; a tuning build (tools/build_variant.sh) reports its name and -D flags, and the product loader refuses it. */
const char* a0_build_info(void);
int a0_device_info(int* cu_count, long long* hbm_bytes, char* arch_name64);

/* ---------------------------------------------------------------- network geometry (agent0/deepq/model.py:90-105) */
typedef struct a0_net_desc {
    int C, H, W;          /* observation shape, cfg.obs_shape (agent0/deepq/main.py:31) */
} a0_net_desc;

typedef struct a0_net a0_net;

/* where a batch of u8 observations lives: frames[(slot ? slot[b] : b) * sample_stride + chan_off + c*H*W + y*W + x] */
typedef struct a0_frames_arg {
    const uint8_t* frames;
    const int* slot;            /* optional gather indices (replay sample), NULL = dense batch */
    long long sample_stride;    /* bytes between samples: 8*H*W for replay rows, 4*H*W for actor observations */
    int chan_off;               /* 0 = st, 4*H*W = st_next half of a replay row (agent0/deepq/agent.py:132-135) */
} a0_frames_arg;

/* one pending slab reduction of a weight gradient (loss.backward(), agent.py:153-155): out[i] = sum_z slabs[z * slab_stride + i], i < count (z ascending: deterministic) */
typedef struct a0_reduce_seg { const float* slabs; long long slab_stride; int nslab; float* out; long long count; } a0_reduce_seg;
typedef struct a0_pending_reduce { a0_reduce_seg seg[4]; int n; } a0_pending_reduce;

/* packed conv weights: w1 [32][C*8*8] in (c,kh,kw) order, w2 [64][4*4*32] and w3 [64][3*3*64] in (kh,kw,c) order */
typedef struct a0_encoder_weights {
    const float *w1, *b1, *w2, *b2, *w3, *b3;
} a0_encoder_weights;

/* one forward pass of a0_net_encoder_fwd_fused_multi: fragment-major weight copies (a0_net_conv_wt_refresh), biases (w->b1 .. b3), frames, observations, outputs */
typedef struct a0_encoder_pass {
    const float* wt; const a0_encoder_weights* w; const a0_frames_arg* f; int B;
    float *act1, *act2, *act3;      /* act1 / act2 optional (a pass that is differentiated keeps them) */
} a0_encoder_pass;

int a0_net_create(const a0_net_desc* desc, a0_net** out);
int a0_net_destroy(a0_net* net);
int a0_net_geometry(const a0_net* net, int* out8);  /* H1,W1,H2,W2,H3,W3,feat_dim,K1 */

/* ConvEncoder.forward (model.py:104-105) with the uint8->fp32 /255 of agent.py:27 / agent.py:129-134 fused into conv1.
 * Outputs are NHWC: act1 [B][H1][W1][32], act2 [B][H2][W2][64], act3 [B][H3][W3][64] (= features in (h,w,c) order). */
int a0_net_encoder_fwd(const a0_net* net, const a0_encoder_weights* w, const a0_frames_arg* frames, int B,
                       float* act1, float* act2, float* act3, void* stream);

/* Fused variant of a0_net_encoder_fwd: one workgroup per observation, activations resident in LDS, bit-identical outputs.
 * wt = k-major copies of the three conv weight matrices (a0_net_conv_wt_floats(C) floats, refreshed by a0_net_conv_wt_refresh
 * whenever the packed weights change); act1 / act2 may be NULL when no backward pass follows. */
int a0_net_encoder_fused_supported(int C, int H, int W);
long long a0_net_conv_wt_floats(int C);
int a0_net_conv_wt_refresh(const a0_encoder_weights* w, int C, float* wt, void* stream);
/* the online network's copies after an optimizer step, mirrored into the target network's copies when that step triggered a target sync
 * (state[4] of a0_adam_step_sync): one launch per update instead of one refresh per network */
int a0_net_conv_wt_refresh_sync(const a0_encoder_weights* w, int C, float* wt, float* wt_target, const int* state, void* stream);
int a0_net_encoder_fwd_fused(int C, int H, int W, const float* wt, const a0_encoder_weights* w, const a0_frames_arg* frames, int B,
                             float* act1, float* act2, float* act3, void* stream);
/* Up to three passes of a0_net_encoder_fwd_fused in ONE launch (the learner's target / online passes over one batch — reference agent.py:176-181 / 222-231 evaluate
 * model_target(next_obs), model(next_obs), model(obs) one after the other; they are independent): bit-identical outputs, 4 x 84 x 84 observations only. */
int a0_net_encoder_fwd_fused_multi(int C, int H, int W, int n, const a0_encoder_pass* pass, void* stream);

/* Fused conv3 + conv2 data gradients per observation (84x84 geometry), the backward-data half of ConvEncoder (model.py:93-105 under
 * autograd, agent.py:136-141): d3 [B][7][7][64] -> d2 [B][9][9][64] and d1 [B][20][20][32], masked by act2 / act1 > 0.  wt = the
 * buffer a0_net_conv_wt_refresh fills (it also holds the flipped / phase-split weight matrices this kernel streams). */
int a0_net_encoder_dgrad_fused_supported(int C, int H, int W);
int a0_net_encoder_dgrad_fused(int C, int H, int W, const float* wt, const float* d3, const float* act1, const float* act2, int B, float* d2,
                               float* d1, void* stream);

/* autograd backward of the encoder (agent.py:153-155).  d3 = dL/d(conv3 pre-activation), already ReLU-masked.
 * g1,g2,g3 receive [dW | db] of each conv in the packed layout.  slabs: a0_net_encoder_bwd_scratch() floats. */
long long a0_net_encoder_bwd_scratch(const a0_net* net, int B);
int a0_net_encoder_bwd(const a0_net* net, const a0_encoder_weights* w, const a0_frames_arg* frames, int B,
                       const float* act1, const float* act2, const float* d3, float* d2, float* d1,
                       float* g1, float* g2, float* g3, float* slabs, void* stream);

/* the three weight-gradient GEMMs of a0_net_encoder_bwd alone: d2 / d1 are inputs (from a0_net_encoder_dgrad_fused) */
int a0_net_encoder_wgrad(const a0_net* net, const a0_encoder_weights* w, const a0_frames_arg* frames, int B, const float* act1, const float* act2,
                         const float* d3, const float* d2, const float* d1, float* g1, float* g2, float* g3, float* slabs, const a0_pending_reduce* pend,
                         void* stream);

/* nn.Linear / NoisyLinear forward+backward (model.py:54-62,112-114): Y = act(X W^T + b), W [N][K] row-major.
 * N, K, ldx multiples of 4.  scratch sizes from the *_scratch functions (0 => may pass NULL). */
long long a0_dense_fwd_scratch(int R, int N, int K);
int a0_dense_fwd(const float* X, int ldx, const float* W, const float* b, float* Y, int R, int N, int K, int relu,
                 float* scratch, void* stream);
int a0_dense_dgrad(const float* dY, const float* W, const float* act_mask, float* dX, int R, int N, int K, void* stream);
/* (round 4) a0_dense_dgrad with the ReLU mask X and the unsplit a0_dense_wgrad of ONE layer (loss.backward() through first_dense, agent.py:153-155) as one launch: the two
 * GEMMs have the same number of 64 x 64 tiles (R == N) and together keep the chip's workgroup slots filled; bit-identical to the two calls.  Shapes: _ok (fc1 of a 512-row batch) */
int a0_dense_dgrad_wgrad_ok(int R, int N, int K);
int a0_dense_dgrad_wgrad_ok2(int R, int N, int K);      /* (round 5) _ok, or a SMALL layer whose two gradients together fit one round of 64 x 64 tiles (a distributional head at a
                                                         * 512-row batch): side by side in one launch; the weight gradient is then an unsplit sum (a0_dense_wgrad's up to the association order) */
int a0_dense_dgrad_wgrad(const float* dY, const float* W, const float* X, int ldx, float* dX, float* grad_w_b, int R, int N, int K, void* stream);
/* (round 5) the same launch also carries the NEXT layer's weight gradient: dW2 = dY2^T X2 (+ bias row sums) into grad2 [N2 x K2 | N2], an unsplit sum over the R rows
 * (what a0_dense_wgrad(dY2, X2, ldx2, grad2, R, N2, K2) computes, up to the association order of the fp32 additions).  fc1's two gradients and the head's weight gradient
 * of a learner's backward pass (agent.py:153-155) are independent once the loss kernel has written dY and dY2.  N2 x K2 must have no more 128 x 64 tiles than R x K. */
int a0_dense_dgrad_wgrad2_ok(int R, int N, int K, int N2, int K2);      /* where the library's own learners use it: heads of at most eight such tiles (scalar heads) */
int a0_dense_dgrad_wgrad2(const float* dY, const float* W, const float* X, int ldx, float* dX, float* grad, int R, int N, int K,
                          const float* dY2, const float* X2, int ldx2, float* grad2, int N2, int K2, void* stream);
long long a0_dense_wgrad_scratch(int R, int N, int K);
int a0_dense_wgrad(const float* dY, const float* X, int ldx, float* grad_w_b, int R, int N, int K, float* slabs, void* stream);
/* n <= 4 dense weight gradients (the head's and fc1's, + the cosine embedding's) whose slab reductions share one launch; layer i reduces in
 * slabs + slab_off[i] (a0_dense_wgrad_scratch floats each, disjoint).  The pointer / shape arrays are host arrays. */
int a0_dense_wgrad_multi(int n, const float* const* dY, const float* const* X, const int* ldx, float* const* grad_w_b, const int* R, const int* N, const int* K,
                         float* slabs, const long long* slab_off, a0_pending_reduce* pend, void* stream);
/* (round 4) pend != NULL: the slab reductions are NOT launched but appended to *pend (set pend->n = 0 before the first call); the next
 * a0_net_encoder_wgrad(..., pend, ...) adds them to its own reduction launch — one launch less per update.  Until then the gradients of the deferred layers are
 * not final and their slabs (slabs + slab_off[i]) must stay untouched: give the encoder a slab region of its own. */

/* Y[r] = act(X[r] W^T + b) * M[r / group]: the IQN / FQF embedding relu(cosine_emb(...)) times the state features (model.py:244-247) in the
 * GEMM's epilogue, for passes that are not differentiated; only when a0_dense_fwd_scratch(R, N, K) == 0 (unsplit GEMM) */
int a0_dense_fwd_mul(const float* X, int ldx, const float* W, const float* b, const float* M, int group, float* Y, int R, int N, int K, int relu, void* stream);

/* the same for a pass that IS differentiated: E[r] = act(X[r] W^T + b) is kept for a0_hadamard_bwd and Y[r] = E[r] * M[r / group] is written in
 * the same launch (replaces a0_dense_fwd + a0_hadamard_fwd on model.py:244-247).  Only for the shapes of the short-reduction kernel
 * (K = 64, R >= 256, N >= 64): a0_dense_fwd_mul_keep_ok returns 1 for them, a0_dense_fwd_mul_keep fails with A0_EINVAL otherwise */
int a0_dense_fwd_mul_keep_ok(int R, int N, int K, int ldx);
int a0_dense_fwd_mul_keep(const float* X, int ldx, const float* W, const float* b, const float* M, int group, float* E, float* Y, int R, int N, int K, int relu, void* stream);

/* a0_dense_fwd without its slab reduction: slab z of [R][N] at stride R*N holds X W^T over the z-th k range; the consumer kernel sums the
 * a0_dense_fwd_partial_slabs(R, N, K) slabs in order, adds the bias and applies the activation (a0_dqn_head_loss_slabs) */
int a0_dense_fwd_partial_slabs(int R, int N, int K);
int a0_dense_fwd_partial(const float* X, int ldx, const float* W, int R, int N, int K, float* slabs, void* stream);
/* (round 4) n = 2 or 3 passes of ONE layer shape with their own inputs, weights and slab buffers — the target / online fc1 passes of an update (agent.py:176-181) — in one
 * launch: fewer, deeper splits per pass (a0_dense_fwd_partial_multi_slabs of them; the partial sums associate differently from a0_dense_fwd_partial's) because the passes
 * fill the chip together.  Host arrays of device pointers; shapes: a0_dense_fwd_partial_multi_ok */
int a0_dense_fwd_partial_multi_ok(int n, int R, int N, int K);
int a0_dense_fwd_partial_multi_slabs(int n, int R, int N, int K);
int a0_dense_fwd_partial_multi(int n, const float* const* X, int ldx, const float* const* W, int R, int N, int K, float* const* slabs, const long long* slab_stride,
                               void* stream);      /* slab_stride (optional, per pass): floats between consecutive slabs, default R * N — passes may share one [splits][2R][N] buffer */
/* The reduction a0_dense_fwd performs behind its split-K GEMM, for up to four layers of the same width N in ONE launch: out[i] = act(sum_z slabs[i][z] + bias[i])
 * (slabs added in slab order: bit-identical to a0_dense_fwd).  The three fc1 passes of a distributional update (agent.py:219-231: online on s, online on s',
 * target on s') finish in one launch instead of three.  Host arrays of n entries; every buffer 16-byte aligned, slab strides multiples of 4 floats. */
int a0_reduce_bias_act_multi(int n, const float* const* slabs, const long long* slab_stride, const int* nslab, const float* const* bias, float* const* out,
                             const int* rows, int N, int relu, void* stream);

/* measurement hook for bench.py: HIP events around every launch of the GEMM tagged `tag` (1 conv1 fwd, 2 conv2 fwd, 3 conv3 fwd,
 * 4 dense fwd, 5 dense dgrad, 6 dense wgrad, 7/8 conv3 wgrad/dgrad, 9/10 conv2 wgrad/dgrad, 11 conv1 wgrad, 12 fused encoder, 13 fused encoder dgrad,
 * 14 the actor step's tail + env step + encoder kernel of a0_actor_qhead_env_step_enc), recorded on the
 * launch stream.  a0_probe_end writes host_out3 = {launches, total ms, total algorithmic FLOP (2*M*N*K)}. */
int a0_probe_begin(int tag, int max_launches);
int a0_probe_end(double* host_out3);

/* Named host-side ranges for rocprofv3 --marker-trace (roctx; SURVEY.md §5 — the reference has wall-clock timing only, trainer.py:176-180).  Active only when the
 * process was started with A0_ROCTX=1 and librocprofiler-sdk-roctx.so can be loaded (a0_trace_enabled); otherwise every call is a branch.  The library's own
 * handles open ranges `rollout` (a0_actor_rollout), `update` (a0_learner_update), `exchange` (the gradient all-reduce's enqueue), `sample` (a0_rbuf_sample*);
 * a host adds its own around them (the Python Trainer: `iteration`, `update_block`).  a0_trace_rank(r) prefixes every name with "r<r>:" (r < 0: no prefix).
 * A range brackets the ENQUEUE on the host; the kernels it launched are in the kernel trace of the same run. */
int a0_trace_enabled(void);
int a0_trace_rank(int rank);
int a0_trace_push(const char* name);
int a0_trace_pop(void);

/* Weights as term planes (round 6): a0_split_planes writes the three exact bf16 terms of W [N][K] (K % 4 == 0) into `planes` (a0_weight_planes_words(N, K) 32-bit words) once
 * per change of W; a0_dense_fwd_wplanes is a0_dense_fwd (model.py:112-114) for the shapes a0_dense_fwd_wplanes_ok accepts (unsplit, K % 32 == 0, R >= 2048, N >= 128) reading them —
 * the same products, the same result bit for bit, without the weight operand's split instructions in the GEMM. */
long long a0_weight_planes_words(int N, int K);
int a0_split_planes(const float* W, unsigned int* planes, int N, int K, void* stream);
int a0_dense_fwd_wplanes_ok(int R, int N, int K);
int a0_dense_fwd_wplanes(const float* X, int ldx, const unsigned int* Wplanes, const float* b, float* Y, int R, int N, int K, int relu, void* stream);

/* Matrix pipe used by every fp32-operand GEMM above (dense layers, conv2/conv3 weight gradients, unfused conv layers):
 * 1 (default) = bf16 MFMA with both operands split exactly into three bf16 terms, nine products, fp32 accumulation (igemm_x9.h);
 * 0 = fp32 MFMA fmaf chain (igemm.h).  Same results up to the association order of the fp32 additions.  Returns the previous mode;
 * mode < 0 only queries.  Process-wide, not a per-stream setting: change it between launches, never during graph capture. */
int a0_gemm_mode(int mode);

/* Cross products the split-operand kernels form per multiply (the x9 GEMM family, the fused encoder's conv2 / conv3 and their data gradients, the fused conv2 / conv3
 * weight gradient): with a = a0 + a1 + a2, b = b0 + b1 + b2 (exact bf16 terms, |a1| <= 2^-8 |a|, |a2| <= 2^-16 |a|)
 *   9 = all of them: every partial product of the fp32 fmaf chain, exactly (the strict mode);
 *   6 = those with i + j <= 2 (default): a1*b2, a2*b1, a2*b2 — each below 2^-24 of a*b, below the rounding the fp32 chain applies to every partial sum — are not
 *       formed; against fp64 the result is at least as close as the fp32 fmaf chain's at every reduction length of the path (profiles/r06_x6_accuracy.txt).
 * n = 6 or 9 sets it, anything else only queries; returns the previous value.  Environment: A0_X9_PRODUCTS=9.  Process-wide: change it between launches, never
 * during graph capture (captured graphs keep the kernels they were captured with).  conv1 (bytes x three weight terms) always forms all three. */
int a0_x9_products(int n);

/* ---------------------------------------------------------------- heads and losses */
/* dueling combine (model.py:127-130,168-172,228-231): raw [R][ld] = [A*T advantages | T values | pad] -> q [R][A][T] */
int a0_dueling_fwd(const float* raw, int ld, float* q, int R, int A, int T, int dueling, void* stream);
int a0_dueling_bwd(const float* dq, float* draw, int ld, int R, int A, int T, int dueling, void* stream);

/* head.qval + argmax (model.py:133,176-177,190-192,253-257,280-284; agent.py:32,177-180,224-227,277-280).
 * x(b,a,t) = x[b*sb + a*sa + t*st]; mode 0 identity, 1 mean over t, 2 C51 expectation (aux = atoms[T]),
 * 3 FQF sum (tau[t+1]-tau[t]) x (aux = taus [B][T+1]).  Outputs optional. */
int a0_select_action(const float* x, long long sb, long long sa, long long st, int B, int A, int T, int mode,
                     const float* aux, int* a_star, float* qsel, float* qmax, void* stream);
/* Actor.act's tail for distributional heads (c51: mode 2 with atoms [T]; qr: mode 1) in one launch: head slabs (a0_dense_fwd_partial) ->
 * slab sum + bias -> dueling combine -> expectation -> first-max argmax -> epsilon-greedy draw from the actor's Philox streams.  Same
 * arithmetic as a0_dense_fwd's reduction + a0_dueling_fwd + a0_select_action + a0_actor_egreedy_rng. */
int a0_actor_dist_tail(const float* slabs, long long slab_stride, int nslab, const float* bias, int ld, int A, int T, int dueling, int mode,
                       const float* atoms, int E, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                       unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax, void* stream);
/* a0_actor_dist_tail + a0_env_synth_step_commit in ONE launch (a workgroup per env; the distributional counterpart of a0_actor_qhead_env_step):
 * arguments of a0_actor_dist_tail, then those of a0_env_synth_step_commit (the action is taken from, and written to, `action`). */
int a0_actor_dist_tail_env_step(const float* slabs, long long slab_stride, int nslab, const float* bias, int ld, int A, int T, int dueling, int mode,
                                const float* atoms, int E, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                                unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax,
                                unsigned long long env_seed, unsigned int rank, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                                float* final_mask, float* final_ret, int n, long long steps, double gamma, int* ring_act, float* ring_rew, float* ring_done,
                                const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act, float* r_rew, float* r_done, int task, void* stream);

/* the quantile networks' counterpart (iqn: mode 1, mean over the T = K sampled quantiles; fqf: mode 3, sum over the T = F fractions weighted by their widths,
 * taus [E][T + 1]): `slabs` are the split-K slabs [nslab][E * T][ld] of the head GEMM over (env, quantile) rows (a0_dense_fwd_partial), columns = actions
 * (+ the dueling value); replaces a0_reduce_bias_act + a0_dueling_fwd + a0_select_action + a0_actor_egreedy_rng + a0_env_synth_step_commit of one actor
 * step (reference agent.py:25-39 with model.py:253-257 / 280-284 behind it, and agent.py:44-90's per-step work), same arithmetic statement for statement */
int a0_actor_quantile_tail_env_step(const float* slabs, long long slab_stride, int nslab, const float* bias, int ld, int A, int T, int dueling, int mode,
                                    const float* taus, int E, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                                    unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax,
                                    unsigned long long env_seed, unsigned int rank, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                                    float* final_mask, float* final_ret, int n, long long steps, double gamma, int* ring_act, float* ring_rew, float* ring_done,
                                    const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act, float* r_rew, float* r_done, int task, void* stream);

/* DQNLearner.train_step (agent.py:173-190): loss [B], dq [B][A] = d(sum_b w_b loss_b)/dq */
int a0_loss_dqn(const float* q, const float* q_next, int A, const int* act, const int* a_star, const float* rew,
                const float* done, const float* wgt, float gamma_n, int B, float* loss, float* dq, int* nan_flag, void* stream);
/* DQNLearner.train_step (agent.py:173-190) from the fc1 activations on, one launch: q heads (online on h(s), target on h'(s'), online on
 * h(s') when h_sel != NULL = double-Q), dueling combine (model.py:123-131), argmax, smooth-L1 loss, and the gradient w.r.t. the raw head
 * outputs draw [B][ld] (dueling backward applied).  W_* [A(+1)][512] + b_*; q_on_out [B][A] (q_tg_out optional) for inspection. */
int a0_dqn_head_loss(const float* h_on, const float* h_tg, const float* h_sel, const float* W_on, const float* b_on, const float* W_tg,
                     const float* b_tg, int A, int dueling, int ld, const int* act, const float* rew, const float* done, const float* wgt,
                     float gamma_n, int B, float* loss, float* q_on_out, float* q_tg_out, float* draw, int* nan_flag, void* stream);
/* a0_dqn_head_loss that also finishes the three fc1 layers from their split-K slabs (a0_dense_fwd_partial): slab sum in order + bias + ReLU,
 * bit-identical to a0_dense_fwd; writes the online activations h(s) [B][512] for the backward pass.  slabs_sel = NULL without double-Q.
 * dh_out (optional, [B][512]): the gradient w.r.t. h(s), i.e. draw times the online head's rows masked by h > 0 — what a0_dense_dgrad would
 * compute from draw (the head's backward-data pass, agent.py:153-155), produced here where all of its operands already are. */
int a0_dqn_head_loss_slabs(const float* slabs_on, const float* slabs_tg, const float* slabs_sel, long long slab_stride, int nslab, const float* b1_on,
                           const float* b1_tg, float* h_on_out, const float* W_on, const float* b_on, const float* W_tg, const float* b_tg, int A,
                           int dueling, int ld, const int* act, const float* rew, const float* done, const float* wgt, float gamma_n, int B, float* loss,
                           float* q_on_out, float* q_tg_out, float* draw, int* nan_flag, float* dh_out, void* stream);

/* MDQNLearner.train_step (agent.py:194-215): q_next = target(next_obs), q_cur_tgt = target(obs), both [B][A] */
int a0_loss_mdqn(const float* q, const float* q_next, const float* q_cur_tgt, int A, const int* act, const float* rew, const float* done,
                 const float* wgt, float gamma_n, float tau, float lo, int B, float* loss, float* dq, int* nan_flag, void* stream);
/* (round 5) MDQNLearner.train_step (agent.py:193-215) from the fc1 GEMMs' split-K slabs on, one launch — a0_dqn_head_loss_slabs with the Munchausen target: slabs_cur are
 * the TARGET network's fc1 slabs on the CURRENT observation (agent.py:202-204; finished with the target's fc1 bias, evaluated with the target's head), tau / lo as in
 * a0_loss_mdqn.  q_cur_out (optional, [B][A]) receives target(obs).  Given the q values it writes, loss and draw equal a0_loss_mdqn + a0_dueling_bwd bit for bit. */
int a0_mdqn_head_loss_slabs(const float* slabs_on, const float* slabs_tg, const float* slabs_cur, long long slab_stride, int nslab, const float* b1_on,
                            const float* b1_tg, float* h_on_out, const float* W_on, const float* b_on, const float* W_tg, const float* b_tg, int A,
                            int dueling, int ld, const int* act, const float* rew, const float* done, const float* wgt, float gamma_n, float tau, float lo,
                            int B, float* loss, float* q_on_out, float* q_tg_out, float* q_cur_out, float* draw, int* nan_flag, float* dh_out, void* stream);
/* C51Learner.train_step (agent.py:219-269) from the head GEMMs' split-K slabs on, one launch: slab sums + bias (a0_dense_fwd's reduction), dueling combine per
 * atom (model.py:163-177), greedy next action from the expectation under softmax (agent.py:225-231), projection of the target distribution and cross entropy
 * (a0_loss_c51), d loss / d logits carried back through the dueling combine into `draw` [B][ld] — the gradient w.r.t. the online head GEMM's output on s.
 * slabs_on [nslab_on][rows_on][ld]: the ONLINE head over rows [0, B) = s and, under double-Q, rows [sel_off, sel_off + B) = s' (one GEMM over both, they share
 * the weights); sel_off < 0: no double-Q, the target's own expectation selects.  slabs_tg [nslab_tg][B][ld]: the target head on s'.  Optional outputs:
 * q_on_out / q_tg_out [B][A][T] (the combined logits), m_out [B][T] (projected target), a_star_out [B].  Same arithmetic, statement for statement, as
 * a0_dense_fwd's reduction + a0_dueling_fwd + a0_select_action + a0_loss_c51 + a0_dueling_bwd. */
int a0_c51_head_loss_slabs(const float* slabs_on, long long stride_on, int nslab_on, int rows_on, const float* slabs_tg, long long stride_tg, int nslab_tg,
                           int sel_off, const float* bias_on, const float* bias_tg, int ld, int A, int T, int dueling, const int* act, const float* rew,
                           const float* done, const float* wgt, const float* atoms, float gamma_n, float vmin, float vmax, int B, float* loss, float* draw,
                           float* q_on_out, float* q_tg_out, float* m_out, int* a_star_out, int* nan_flag, void* stream);
/* (round 5) QRLearner.train_step (agent.py:272-293 with huber_qr_loss 110-114) from the head GEMMs' split-K slabs on, one launch — the quantile-regression counterpart
 * of a0_c51_head_loss_slabs, same slab layout and the same meaning of rows_on / sel_off: slab sums + bias, dueling combine per quantile (model.py:163-177 through QRHead
 * 180-192), greedy next action from the mean over the T quantiles (agent.py:277-280), target quantiles r + gamma^n (1 - d) q'(a*), the T x T pairwise quantile Huber loss
 * with fractions taus [T] (the fixed midpoints, agent.py:274), d loss / d q carried back through the dueling combine into `draw` [B][ld].  One workgroup per sample, the
 * staged head outputs and the targets in LDS, the B x T x T pair tensor never materialised.  Optional outputs: q_on_out / q_tg_out [B][A][T], a_star_out [B].  Same
 * arithmetic, statement for statement, as a0_dense_fwd's reduction + a0_dueling_fwd + a0_select_action(mode 1) + a0_quantile_target + a0_loss_quantile_huber +
 * a0_dueling_bwd. */
int a0_qr_head_loss_slabs(const float* slabs_on, long long stride_on, int nslab_on, int rows_on, const float* slabs_tg, long long stride_tg, int nslab_tg,
                          int sel_off, const float* bias_on, const float* bias_tg, int ld, int A, int T, int dueling, const int* act, const float* rew,
                          const float* done, const float* wgt, const float* taus, float gamma_n, int B, float* loss, float* draw, float* q_on_out,
                          float* q_tg_out, int* a_star_out, int* nan_flag, void* stream);
/* C51Learner.train_step (agent.py:219-269): logits / tgt_logits [B][A][T]; m_out (optional) = projected target [B][T] */
int a0_loss_c51(const float* logits, const float* tgt_logits, int A, int T, const int* act, const int* a_star,
                const float* rew, const float* done, const float* wgt, const float* atoms, float gamma_n,
                float vmin, float vmax, int B, float* loss, float* dlogits, float* m_out, int* nan_flag, void* stream);
/* quantile target r + gamma^n (1-d) q_next[a*] (agent.py:281-286,313-318,358-364) and BaseLearner.huber_qr_loss
 * (agent.py:110-114) with its gradient; strides let QR ([B][A][N]) and IQN/FQF ([B][N][A]) share the kernels. */
int a0_quantile_target(const float* q_next, long long sb, long long sj, long long sa, const int* a_star, const float* rew,
                       const float* done, float gamma_n, int B, int Nd, float* y, void* stream);
int a0_loss_quantile_huber(const float* q, long long sb, long long si, long long sa, const float* y, const float* taus,
                           long long tb, const int* act, const float* wgt, int B, int N, int Nd, float* loss, float* dq,
                           int* nan_flag, void* stream);

/* IQN / FQF head pieces (model.py:235-251, 268-278; agent.py:371-387) */
int a0_cos_features(const float* taus, float* out, long long R, int D, void* stream);
/* (round 6) the actor's IQN step: taus[r] = element offset + r of the Philox uniform stream (== a0_rng_uniform / a0_rng_uniform_ctrl: ctrl, when given, adds
 * ctrl[ctrl_idx] to the offset) and out = a0_cos_features(taus) in ONE launch; the same bits as the two calls (model.py:238-241) */
int a0_tau_cos_features(unsigned long long seed, unsigned int stream, unsigned long long offset, const long long* ctrl, int ctrl_idx, float* taus, float* out,
                        long long R, int D, void* stream_h);
int a0_hadamard_fwd(const float* emb, const float* feat, float* x, int B, int n, int D, void* stream);
int a0_hadamard_bwd(const float* dx, const float* emb, const float* feat, float* demb, float* d3, int B, int n, int D, void* stream);
/* (round 6) a0_dense_dgrad(dY [R][N], W [N][K], no mask) + a0_hadamard_bwd in one launch for R = B * n rows, n = 32 or 64: dx = dY W is consumed in the GEMM's epilogue and never
 * reaches HBM.  demb [R][K] is bit-identical to the two calls'; d3 [R / n][K] sums a sample's n rows in another order (fp32 rounding).  Shapes: a0_dense_dgrad_hadamard_ok. */
int a0_dense_dgrad_hadamard_ok(int R, int N, int K, int n);
int a0_dense_dgrad_hadamard(const float* dY, const float* W, const float* emb, const float* feat, float* demb, float* d3, int R, int N, int K, int n, void* stream);
int a0_fqf_taus(const float* logits, int ld, float* taus, float* tau_hat, int B, int F, void* stream);
/* (round 6) a0_fqf_taus + a0_cos_features(tau_hat) -> cos_out [B * F][D] in one launch (the actor's FQF step); the same bits as the two calls */
int a0_fqf_taus_cos(const float* logits, int ld, float* taus, float* tau_hat, float* cos_out, int D, int B, int F, void* stream);
int a0_fqf_inner_taus(const float* taus, float* out, int B, int F, void* stream);
int a0_fqf_fraction_loss(const float* q, const float* qh, const float* taus, const int* act, const float* wgt, int B, int F, int A,
                         int ldl, float* loss, float* dlogits, const float* logits, void* stream);

/* ---------------------------------------------------------------- optimizer / target sync (agent.py:102-106,152-161,333-338) */
/* state: int[8] device block: [0] nan flag (set by losses) [1] update_steps [2] skipped [3] skip_now [4] sync_now [5] scratch of a0_adam_step_sync_wt
 * [6] calls of a0_adam_step_sync_wt with a loss ring (free-running: the ring slot of the next call is state[6] % ring_cap) */
int a0_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int* state,
                 float* scalars2, double lr, double beta1, double beta2, double eps, int target_update_freq, void* stream);
/* a0_adam_step with the target copy of agent.py:160-161 folded into the same pass: when update_steps % target_update_freq == 0 after this
 * update, target[0, n_total) receives the new parameters (n_total >= n also covers blocks Adam does not own).  == a0_adam_step +
 * a0_target_sync(force = 0), two launches instead of three. */
int a0_adam_step_sync(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int* state, float* scalars2, double lr,
                      double beta1, double beta2, double eps, int target_update_freq, float* target, long long n_total, const float* extra_nan_flag,
                      void* stream);
/* The optimizer tail of a network whose convolutions run in the fused kernels, two launches instead of three: a0_adam_step_sync with the step's
 * bookkeeping (NaN skip, step count, bias corrections, "sync now": agent.py:152-161) derived inside the Adam kernel, then
 * a0_net_conv_wt_refresh_sync of the online copies `wt` (mirrored into `wt_target` on a sync step), which also commits the step count.
 * state[5] is scratch.  Same results as the two calls it replaces.  loss (optional, [loss_n]): the update's per-sample losses; their batch mean — the Trainer's
 * `loss` statistic, trainer.py:99,111-113, == a0_mean_rows — is written to loss_ring[state[6] % ring_cap] by the Adam launch itself (one launch per update less). */
int a0_adam_step_sync_wt(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int* state, float* scalars2, double lr,
                         double beta1, double beta2, double eps, int target_update_freq, float* target, long long n_total, const float* extra_nan_flag,
                         const a0_encoder_weights* w, int C, float* wt, float* wt_target, const float* loss, int loss_n, float* loss_ring, int ring_cap, void* stream);
/* ---------------------------------------------------------------- a whole learner behind one handle (opaque handles, library-owned HBM)
 * BaseLearner (agent.py:97-169) with DQNLearner.train_step (173-190) for scalar heads on 4 x 84 x 84 observations — BASELINE configs[1]: online + target parameters
 * in the packed layout (agent0_amd/deepq/layout.py: conv1 | conv2 | conv3 | fc1 | head, each [W (N x K) | b (N)], head rows padded to a multiple of 32), gradients,
 * Adam moments, status words and every workspace live in HBM that a0_learner_create allocates and a0_learner_destroy frees; a0_learner_update is BaseLearner.train —
 * forward passes, loss, backward, Adam(lr, eps = adam_eps or 1e-2 / B), NaN guard, update counter, target copy every target_update_freq updates — as ONE call that
 * enqueues the same launches, in the same order, as the per-kernel entry points above (bit-identical results), never allocates and never synchronises.
 * algo = A0_ALGO_C51 (round 4): C51Learner.train_step (agent.py:218-268) — BASELINE configs[2], with NoisyLinear layers (model.py:28-87; packed as fc1.mu | fc1.sigma |
 * head.mu | head.sigma, composed weights and the noise vectors in HBM of the handle, both networks' noise redrawn per update from Philox stream 4 of `seed` exactly
 * as BaseLearner.train does, agent.py:125-127), dueling, double-Q and n-step.  algo = A0_ALGO_IQN / A0_ALGO_FQF: BASELINE configs[3] / [4]; A0_ALGO_QR / A0_ALGO_MDQN: the reference's remaining learners
 * (num_atoms quantiles per action for qr; NoisyNet allowed for both). */
#define A0_ALGO_DQN 0
#define A0_ALGO_C51 1
#define A0_ALGO_QR 4    /* QRLearner.train_step (agent.py:272-293): num_atoms fixed quantiles per action, quantile Huber loss at the midpoints (2 i + 1) / (2 T) */
#define A0_ALGO_MDQN 5  /* MDQNLearner.train_step (agent.py:194-215): Munchausen DQN, the target network also evaluated on the current observation */
#define A0_ALGO_FQF 3   /* FQFLearner.train_step (agent.py:334-388) with the fraction proposal network (model.py:260-284; its own RMSprop step, agent.py:139-148): packed layout
                         * conv1 | conv2 | conv3 | fc1 | head | cos | frac, Adam over everything before frac; BASELINE configs[4] */
#define A0_ALGO_IQN 2   /* IQNLearner.train_step (agent.py:296-331) with the cosine-embedding head (model.py:203-257): packed layout conv1 | conv2 | conv3 | fc1 | head | cos;
                         * the three tau draws of an update (K, N', N per sample, in that order) come from Philox stream 3 of `seed` as BaseLearner's do */
typedef struct a0_learner a0_learner;
typedef struct a0_learner_desc {
    int A, dueling, double_q;         /* cfg.action_dim, learner.dueling_head, learner.double_q (config.py:72-95) */
    int B, n_step;                    /* learner.batch_size, learner.n_step_q */
    double discount, lr, adam_eps;    /* learner.discount, learner.learning_rate; adam_eps <= 0: the reference's 1e-2 / B (agent.py:105) */
    int target_update_freq;           /* learner.target_update_freq (agent.py:160-161) */
    int algo;                         /* A0_ALGO_* */
    int num_atoms; double vmin, vmax; /* c51: learner.c51.num_atoms / vmin / vmax (config.py:97-101); the support is torch.linspace(vmin, vmax, num_atoms) */
    int noisy;                        /* learner.noisy_net */
    unsigned long long seed;          /* the learner's Philox seed (the Python classes use cfg.seed + 15485863): noise and tau draws */
    int iqn_K, iqn_N, iqn_N_dash;     /* iqn: learner.iqn.K / N / N_dash (config.py:103-109); 64 cosines */
    int fqf_F;                        /* fqf: learner.iqn.F fractions (<= 32) */
    double mdqn_tau, mdqn_lo;         /* mdqn: learner.mdqn.tau / lo (config.py:88-92) */
    double max_grad_norm;             /* (round 5) learner.max_grad_norm: > 0 clips the fqf fraction net's gradient to this L2 norm before its RMSprop step, the only
                                       * place the reference clips (agent.py:143-147); <= 0 (and a zero-initialised trailing field): no clipping */
} a0_learner_desc;
int a0_learner_create(const a0_learner_desc* desc, a0_learner** out);
/* BaseLearner.__init__ (agent.py:97-110) over HBM the CALLER already holds (each pointer may be NULL: the library allocates that buffer): what lets a host that keeps its own views of the
 * parameters — the Python classes' flat tensors, a checkpointing layer — hand the update loop to the library without copies (agent0_amd/deepq/native_loop.py).  Sizes:
 * online / target / adam_m / adam_v a0_learner_param_floats floats (known from a throw-away handle or deepq/layout.py), grads 4 more, state 8 ints, scalars 4,
 * loss_ring loss_ring_cap floats (the per-update batch-mean loss lands in slot state[6] % cap), wt_* a0_net_conv_wt_floats(4), eff_* the composed NoisyNet weights
 * [fc1 | head] and noise [online | target] the noise vectors (layout: a0_learner_desc above).  Borrowed buffers are used as they are: no initial noise is drawn into a
 * borrowed `noise` (a0_learner_set_rng tells the handle where the caller's stream stands).  The library cannot measure a borrowed buffer: `grads` MUST hold
 * a0_learner_param_floats + 4 floats (the NaN flag of a data-parallel update rides behind the parameters), every other size is exactly as listed, all 16-byte aligned. */
typedef struct a0_learner_buffers {
    float *online, *target, *grads, *adam_m, *adam_v;
    int* state;
    float* scalars;
    float* loss_ring; int loss_ring_cap;
    float *wt_online, *wt_target;
    float *eff_online, *eff_target, *noise;
    float* rms_sq;                    /* fqf: the fraction net's RMSprop state, 32 * feat + 32 floats */
} a0_learner_buffers;
int a0_learner_create_on(const a0_learner_desc* desc, const a0_learner_buffers* buffers, a0_learner** out);
/* the next draw offset of one of the learner's Philox streams (3 = the quantile fractions of agent.py:300-310, 4 = NoisyNet's reset_noise of agent.py:125-127) */
int a0_learner_set_rng(a0_learner* learner, int stream_id, unsigned long long offset);
int a0_learner_destroy(a0_learner* learner);
long long a0_learner_param_floats(const a0_learner* learner);
/* Data parallelism through the handle (one process per GPU, the reference itself has no multi-GPU learner — launch.py:30-61's actor processes
 * are its only parallelism): with `comm` from a0_dp_init every a0_learner_update SUMs its gradients over the ranks between backward and Adam — the dense blocks
 * (plus the NaN flag behind them) on a side stream while the encoder backward runs, then the convolution blocks, then the join; Adam skips the step on every rank
 * if any rank saw a NaN.  Same buckets and order as agent0_amd/deepq/dist.py::RcclGradAllReduce, issued eagerly.  Collective: every rank sets it before its first
 * update and calls a0_learner_update the same number of times.  comm = 0 switches the exchange off; the caller keeps ownership of the communicator. */
int a0_learner_set_exchange(a0_learner* learner, long long comm);
/* parameters in the packed layout (device pointers, a0_learner_param_floats floats each); target_packed = NULL: target = copy of online (agent.py:100) */
int a0_learner_set_params(a0_learner* learner, const float* online_packed, const float* target_packed, void* stream);
/* copies of what the handle holds (any pointer may be NULL): parameters, target parameters, Adam moments (param_floats each), the eight status words */
int a0_learner_get(const a0_learner* learner, float* online_out, float* target_out, float* adam_m_out, float* adam_v_out, int* state_out8, void* stream);
/* the handle's own buffer of per-sample losses [B] of the last update — BaseLearner.train's return value (agent.py:163-169) — as a device pointer valid for the
 * handle's lifetime: what a0_rbuf_update_priority (trainer.py:103-104) takes without the copy a non-NULL loss_out of a0_learner_update costs */
int a0_learner_loss_buffer(const a0_learner* learner, float** loss_dev);
/* (round 5) borrowed views of the handle's HBM for inspection — the flat gradient buffer (layout of the parameters) and the differentiated pass's activations
 * (NHWC; their signs are the ReLU decisions of the last update), which a0_learner_get does not copy out: *dev_ptr / *count (floats).  Valid until the next call on
 * the handle.  tests/test_gpu_trace.py walks the handle path against the CPU oracle with them. */
enum { A0_PEEK_GRADS = 0, A0_PEEK_ACT1 = 1, A0_PEEK_ACT2 = 2, A0_PEEK_ACT3 = 3, A0_PEEK_FC1 = 4, A0_PEEK_LOSS = 5 };
int a0_learner_peek(const a0_learner* learner, int what, float** dev_ptr, long long* count);
/* fqf: a copy of the per-sample fraction losses [B] of the last update (the `fraction_loss` statistic, trainer.py:99-101) into out_dev */
int a0_learner_get_frac_loss(const a0_learner* learner, float* out_dev, void* stream);
/* c51: the support atoms [num_atoms] from HOST memory, for a caller that holds the exact values its reference run used (default: linspace in fp32, torch's formula) */
int a0_learner_set_support(a0_learner* learner, const float* atoms_host);
/* frames: u8 replay rows st || st_next of row_bytes bytes, read through slot [B] (ring slots of the sampled batch; NULL = rows 0 .. B-1); act int32, rew / done /
 * wgt fp32 [B]; loss_out (optional, [B]): the per-sample losses, i.e. what update_priority takes (trainer.py:103-104) */
int a0_learner_update(a0_learner* learner, const uint8_t* frames, const int* slot, long long row_bytes, const int* act, const float* rew, const float* done,
                      const float* wgt, float* loss_out, void* stream);

/* ---------------------------------------------------------------- the rest of the loop behind handles: replay ring and actor (csrc/runtime.hip)
 * a0_rbuf = ReplayDataset (replay.py:14-59) + the sampling of trainer.py:63-72,91-96: the HBM ring of st || st_next rows (2 * obs_bytes each) with its metadata,
 * uniform sampling (the DataLoader's shuffled epochs as a Feistel permutation, last batch of an epoch never returned: utils.py:51-56) or, prioritize != 0,
 * prioritize == 1, proportional sampling from the sum-tree with importance weights and the beta schedule, or, prioritize == 2 (round 6), the reference's prioritized
 * mode TO THE LETTER (replay.sumtree=false; replay.py:45-59, trainer.py:91-104, quirks Q1 / Q2 / Q7): uniform permutation batches, a flat priority vector [size]
 * whose TAIL takes max_p^alpha on every extend, importance weights from priority[idx] / the sum over the WHOLE capacity, priority[ids] = (loss + eps)^alpha.
 * `seed` is the sampler's Philox seed (the Python classes use cfg.seed + 104729). */
typedef struct a0_rbuf a0_rbuf;
typedef struct a0_rbuf_desc {
    long long size; int obs_bytes, B, prioritize;      /* replay.size, C*H*W, learner.batch_size, 0 = uniform / 1 = prioritized, sum-tree / 2 = prioritized, the reference's flat vector */
    double alpha, eps, beta0; long long total_steps;   /* replay.alpha / eps / beta0 (config.py:118-124), trainer.total_steps (the beta schedule's length) */
    unsigned long long seed;
} a0_rbuf_desc;
typedef struct a0_batch { const long long* idx; const int* slot; const int* act; const float* rew; const float* done; const float* prio; const float* weights; } a0_batch;
int a0_rbuf_create(const a0_rbuf_desc* desc, a0_rbuf** out);
/* ReplayDataset.__init__ (replay.py:14-30) over ring buffers the caller already holds (each may be NULL: library-owned): frames [size * 2 * obs_bytes] u8, act i32 / rew / done f32 [size], tree f32
 * [2 * 2^ceil(log2 size)] (prioritize == 1; with prioritize == 2 this argument is the flat priority vector f32 [size], all ones for an empty ring), max_p f32 [1] (must hold the
 * caller's current max priority: 1 for an empty ring) */
int a0_rbuf_create_on(const a0_rbuf_desc* desc, uint8_t* frames, int* act, float* rew, float* done, float* tree, float* max_p, a0_rbuf** out);
int a0_rbuf_destroy(a0_rbuf* replay);
long long a0_rbuf_len(const a0_rbuf* replay);
long long a0_rbuf_write_cursor(const a0_rbuf* replay);
int a0_rbuf_info(const a0_rbuf* replay, long long* top, long long* written, double* beta);
/* the ring's device buffers (any pointer may be NULL): frames [size][2 * obs_bytes] u8, act int32 / rew / done fp32 [size], the sum-tree [2 * cap2] and max_p [1] */
int a0_rbuf_buffers(a0_rbuf* replay, uint8_t** frames, int** act, float** rew, float** done, float** tree, float** max_p);
/* copies into caller buffers (device pointers, any may be NULL): rows [0, rows) of frames / act / rew / done, the sum-tree (2 * cap2 floats), max_p */
int a0_rbuf_read(const a0_rbuf* replay, long long rows, uint8_t* frames_out, int* act_out, float* rew_out, float* done_out, float* tree_out, float* max_p_out, void* stream);
/* ReplayDataset.extend for n transitions already written into the ring at the write cursor (a0_actor_rollout does that) */
int a0_rbuf_commit(a0_rbuf* replay, long long n, void* stream);
/* (round 5) ReplayDataset.extend for a rollout that an asynchronous actor wrote into ANOTHER ring (`stage`: an a0_rbuf of a few rollouts' rows; the launch schedule,
 * launch.py:47-62): rows [start_row, start_row + n) of `stage` are copied to this ring's write cursor and committed like a0_rbuf_commit.  The caller orders the call
 * behind the rollout's end. */
int a0_rbuf_extend_from(a0_rbuf* replay, const a0_rbuf* stage, long long start_row, long long n, void* stream);
/* one batch into the handle's persistent batch buffers (device pointers in *out, valid until the next sample) */
int a0_rbuf_sample(a0_rbuf* replay, a0_batch* out, void* stream);
/* uniform replay: the next n <= 32 batches — exactly the batches n consecutive a0_rbuf_sample calls would return (trainer.py:63-72: they do not depend on the updates
 * between them) — drawn by ONE launch into n persistent batch buffers (valid until the next sample call of either kind).  Prioritized replay: A0_EINVAL. */
int a0_rbuf_sample_block(a0_rbuf* replay, int n, a0_batch* out, void* stream);
/* ReplayDataset.update_priority with the last batch's indices and the learner's per-sample losses; learner_state: a0_learner_get's status words or NULL */
int a0_rbuf_update_priority(a0_rbuf* replay, const float* loss, const int* learner_state, void* stream);

/* a0_actor = Actor (agent.py:19-90) on the device-resident synthetic env for the heads a0_learner covers (scalar; categorical with or without NoisyNet; implicit quantile; fully parameterised quantile): observations, epsilon-greedy Philox streams (seed, rank), n-step ring,
 * episode statistics.  a0_actor_rollout = Actor.sample: T steps acting with the learner's online network, transitions written into the replay ring at its write
 * cursor (then a0_rbuf_commit(replay, T * E)); a0_actor_collect waits for the stream and returns qs [T] and the finished episodes' returns (host memory). */
typedef struct a0_actor a0_actor;
typedef struct a0_actor_desc {
    int E, T, A, dueling, n_step;                      /* actor.num_envs, actor.sample_steps, cfg.action_dim, learner.dueling_head, learner.n_step_q */
    double discount;                                   /* learner.discount */
    unsigned long long seed; unsigned int rank;        /* cfg.seed, this process's rank */
    int env_task;                                      /* A0_ENV_TASK_* */
    int reset_noise_freq;                              /* learner.reset_noise_freq (NoisyNet learners; 0 = the reference's default 4) */
} a0_actor_desc;
int a0_actor_create(const a0_actor_desc* desc, a0_actor** out);
int a0_actor_destroy(a0_actor* actor);
/* (round 5) every workspace a rollout with `learner` needs, allocated at set-up (a0_actor_rollout on an unbound actor binds it first — the one call after create that
 * allocates); own_network != 0: the actor also gets its OWN copy of the network (parameters, weight copies, NoisyNet noise and composed weights), which
 * a0_actor_snapshot refreshes from the learner — the launch schedule's actor.futures.sample(eps, state_dict) (launch.py:34-36,58-62).  Without it the actor acts with
 * the learner's online network, NoisyNet buffers included, like the reference's train actor on the main schedule (trainer.py:41-44: the SAME module). */
int a0_actor_bind(a0_actor* actor, const a0_learner* learner, int own_network);
int a0_actor_snapshot(a0_actor* actor, const a0_learner* learner, void* stream);
/* (the learner is not const: like the reference's single-process main, the actor acts with the learner's own network object — a NoisyNet actor redraws that
 * network's noise every reset_noise_freq steps from ITS Philox stream 4 and recomposes the effective weights, agent.py:52-53) */
int a0_actor_rollout(a0_actor* actor, a0_learner* learner, a0_rbuf* replay, float epsilon, void* stream);
int a0_actor_collect(a0_actor* actor, float* qs_host, float* returns_host, int max_returns, int* n_returns, void* stream);
/* a0_actor_collect (the rs / qs of Actor.sample, agent.py:85-90) in two halves, so that the next rollout can be enqueued before the host waits: _begin enqueues the copies of the statistics into page-locked
 * buffers of the handle and records an event (call it BEFORE the next a0_actor_rollout, which reuses the device buffers); _end waits for that event only */
int a0_actor_collect_begin(a0_actor* actor, void* stream);
int a0_actor_collect_end(a0_actor* actor, float* qs_host, float* returns_host, int max_returns, int* n_returns);

/* data parallelism: this rank's NaN flag as a float (1.0 / 0.0) that rides at the tail of a SUM-reduced gradient bucket; the reduced value
 * comes back through extra_nan_flag (nonzero = some rank saw a NaN: every rank skips the step), NULL on one GPU */
int a0_nan_flag_export(const int* state, float* out, void* stream);
int a0_rmsprop_step(float* params, const float* grads, float* square_avg, long long n, double lr, double alpha, double eps,
                    double max_grad_norm, float* clip_scratch, void* stream);
int a0_target_sync(float* target, const float* online, long long n, const int* state, int force, void* stream);
/* NoisyLinear.reset_noise / forward weight composition and its gradient fan-out (model.py:54-62,73-87) */
int a0_noisy_compose(const float* mu, const float* sigma, float* eff, int N, int K, int r0, int r1, const float* noise_in,
                     const float* noise_out_w, const float* noise_out_b, void* stream);
int a0_noisy_grad_sigma(const float* gmu, float* gsigma, int N, int K, int r0, int r1, const float* noise_in,
                        const float* noise_out_w, const float* noise_out_b, void* stream);
/* up to three NoisyLinear modules (first_dense, q_head, value_head) in one launch; host arrays of per-module arguments.
 * grad = 0: a0_noisy_compose per module; grad = 1: a0_noisy_grad_sigma per module (mu = gmu, eff = gsigma, sigma unused) */
int a0_noisy_multi(int grad, int nmod, const float* const* mu, const float* const* sigma, float* const* eff, const int* N, const int* K, const int* r0,
                   const int* r1, const float* const* noise_in, const float* const* noise_out_w, const float* const* noise_out_b, void* stream);

/* ---------------------------------------------------------------- replay (agent0/deepq/replay.py:14-59, trainer.py:63-72,91-96) */
int a0_replay_insert(uint8_t* frames, long long cap, int obs_bytes, long long start_slot, int n, const uint8_t* obs,
                     const uint8_t* obs_next, const int* act, const float* rew, const float* done, int* r_act,
                     float* r_rew, float* r_done, const long long* ctrl, void* stream);
int a0_replay_lookup(const long long* idx, int B, long long top, long long head, long long cap, int* slot, const int* r_act,
                     const float* r_rew, const float* r_done, const float* priority, int* act, float* rew, float* done,
                     float* prio, long long* idx_out, void* stream);
int a0_replay_gather(const uint8_t* frames, int row_bytes, const int* slot, int B, uint8_t* out, void* stream);
/* index generation + metadata lookup + row gather in ONE launch (trainer.py:63-72 + replay.py:32-37): mode 0 = element start+b of the
 * epoch's Feistel permutation of [0, n_perm), mode 1 = stratified sum-tree draw with xi[b] */
int a0_replay_sample_gather(int mode, unsigned long long start, unsigned long long n_perm, unsigned int seed, const float* tree, long long cap2,
                            const float* xi, long long top, long long head, long long cap, const uint8_t* frames, int row_bytes, const int* r_act,
                            const float* r_rew, const float* r_done, const float* priority, int B, uint8_t* out, long long* idx_out, int* slot_out,
                            int* act, float* rew, float* done, float* prio, void* stream);
/* uniform sampling without the row copy (the learner's conv1 reads ring rows through slot_out): == a0_perm_batch + a0_replay_lookup */
int a0_replay_sample_slots(unsigned long long start, unsigned long long n_perm, unsigned int seed, long long top, long long head, long long cap,
                           const int* r_act, const float* r_rew, const float* r_done, const float* priority, int B, long long* idx_out, int* slot_out,
                           int* act, float* rew, float* done, float* prio, void* stream);
/* n <= 32 such batches in ONE launch (host arrays start / n_perm / seed [n]; outputs [n][B]): uniform replay's batches do not depend on the updates between them
 * (the DataLoader's shuffled epochs, trainer.py:63-72), so a whole update block's sampling is one launch */
int a0_replay_sample_slots_multi(int n, const unsigned long long* start, const unsigned long long* n_perm, const unsigned int* seed, long long top, long long head, long long cap,
                                 const int* r_act, const float* r_rew, const float* r_done, int B, long long* idx_out, int* slot_out, int* act, float* rew, float* done,
                                 float* prio, void* stream);
int a0_fill_f32(float* p, long long n, float v, void* stream);
int a0_priority_update(float* priority, const long long* ids, const float* loss, int B, float eps, float alpha,
                       float* pstate, const int* state, void* stream);
int a0_priority_tail(float* priority, long long size, long long n, const float* pstate, float alpha, void* stream);
int a0_sum_f32(const float* x, long long n, float* scratch256, float* out, void* stream);
int a0_is_weights(const float* prio, int B, const float* psum, long long top, float beta, float* w, void* stream);
int a0_perm_batch(unsigned long long start, int count, unsigned long long n, unsigned int seed, long long* out, void* stream);
/* sum-tree (new component, contract in oracle/sumtree.c): tree float[2*cap2] */
/* n <= 1024 leaves per call (larger batches: consecutive calls in batch order, so that a later duplicate still wins).  state (optional,
 * the learner's status words): the call is a no-op when state[3] != 0, i.e. when the update was skipped on a NaN loss (agent.py:152-158) */
int a0_sumtree_set(float* tree, long long cap2, const long long* idx, const float* val, int n, const int* state, void* stream);
/* leaves (start + i) % size, i < n, all set to val[0] (device scalar) and their ancestors recomputed: the rollout's new transitions enter at
 * max_p^alpha (replay.py:45-53) in one launch; same tree as a0_sumtree_set on those pairs */
int a0_sumtree_set_range(float* tree, long long cap2, long long start, long long n, long long size, const float* val, void* stream);
int a0_sumtree_rebuild(float* tree, long long cap2, void* stream);
int a0_sumtree_sample(const float* tree, long long cap2, const float* xi, int B, long long* out_idx, float* out_p, void* stream);
/* one prioritized batch in one launch: stratified uniforms from the sampler's Philox stream, sum-tree descent, slot + metadata lookup and
 * importance weights (trainer.py:91-94); == a0_rng_uniform + a0_sumtree_sample + a0_replay_lookup + a0_is_weights.  B <= 1024. */
int a0_sumtree_sample_batch(unsigned long long seed, unsigned int stream, unsigned long long offset, float* tree, long long cap2, int B, long long top,
                            long long cap, float beta, const int* r_act, const float* r_rew, const float* r_done, long long* idx_out, int* slot_out,
                            int* act, float* rew, float* done, float* prio, float* w, int rebuild_top, void* stream_h);
/* (round 4) the kernel stages the levels with <= 2048 nodes in LDS — recomputed from level 2048 (or the leaves of a smaller tree) — and walks them there;
 * rebuild_top = 1 also writes them back: the launch a0_sumtree_set_from_loss(defer_top = 1) left out, so that replay.py:55-59 followed by the next
 * trainer.py:63-72 batch is two launches */
/* val = (loss + eps)^alpha, pstate[0] = max(pstate[0], max loss) (replay.py:55-59); no-op when state && state[3] (NaN-skipped update) */
int a0_priority_from_loss(const float* loss, int n, float eps, float alpha, float* val, float* pstate, const int* state, void* stream);
/* a0_priority_from_loss + a0_sumtree_set in two launches instead of three (replay.py:55-59 on the sum-tree): the per-subtree kernel forms (loss + eps)^alpha while it
 * stages the batch and keeps pstate[0] = max_p; n <= 1024; for trees with a0_sumtree_set_from_loss_ok(cap2) == 1 (64 ... 1024 leaves per subtree below the top 2048 nodes) */
int a0_sumtree_set_from_loss_ok(long long cap2);
int a0_sumtree_set_from_loss(float* tree, long long cap2, const long long* idx, const float* loss, int n, float eps, float alpha, float* pstate, const int* state,
                             int defer_top, void* stream);
/* defer_top = 1 (replay.py:55-59 followed by the next batch of trainer.py:63-72 in two launches): leaves and subtrees only (one launch); tree[1 .. 2047] are stale until a0_sumtree_sample_batch(rebuild_top = 1), a0_sumtree_set_range (which
 * recomputes the top from level 2048 anyway) or a0_sumtree_top_rebuild has run — every other reader of the top levels must be preceded by one of them */
int a0_sumtree_top_rebuild(float* tree, long long cap2, void* stream);

/* ---------------------------------------------------------------- actor (agent0/deepq/agent.py:25-39,57-73) */
int a0_actor_egreedy(const int* greedy, const int* rand_action, const float* u, float eps, int E, int* action,
                     const float* qmax, float* qs_out, void* stream);
/* Fused actor tail for scalar heads (dqn / mdqn), Actor.act (agent.py:25-39) after the encoder: fc1 (split-K GEMM into `scratch`,
 * a0_actor_qhead_scratch(E, K) floats) + bias + ReLU, q head (W2 [A(+1)][512], b2), dueling combine (model.py:123-131), first-max
 * argmax, epsilon-greedy draw from the Philox streams (same draws as a0_actor_egreedy_rng).  qmax[e] = max_a q(e, a). */
long long a0_actor_qhead_scratch(int E, int K);
int a0_actor_qhead(const float* feat, int E, int K, const float* W1, const float* b1, const float* W2, const float* b2, int A, int dueling,
                   float* scratch, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                   unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax, void* stream);
/* a0_actor_qhead + a0_env_synth_step_commit in two launches instead of three (the fc1 GEMM, then ONE kernel with a workgroup per env: one
 * wave runs the tail and the env's scalar work with the chosen action while the others already write the new frame / stack / replay row).
 * Arguments: those of a0_actor_qhead, then those of a0_env_synth_step_commit (the action is taken from, and written to, `action`). */
int a0_actor_qhead_env_step(const float* feat, int E, int K, const float* W1, const float* b1, const float* W2, const float* b2, int A, int dueling,
                            float* scratch, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                            unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax,
                            unsigned long long env_seed, unsigned int rank, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                            float* final_mask, float* final_ret, int n, long long steps, double gamma, int* ring_act, float* ring_rew, float* ring_done,
                            const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act, float* r_rew, float* r_done, int task, void* stream);
/* out[t] = mean over e of x[t][e] (per-step mean max-Q of a rollout, agent.py:38,88) */
int a0_mean_rows(const float* x, int T, int E, float* out, void* stream);

/* the same with both draws generated in-kernel from Philox streams (bit-identical to a0_rng_randint + a0_rng_uniform + a0_actor_egreedy) */
/* (round 5) a0_actor_qhead_env_step whose second launch goes on to ENCODE the env's new observation: fc1 GEMM over `feat`, then ONE launch per env for tail + env
 * step + the fused encoder over obs_out into act3_next [E][3136] (may be `feat`) — the next actor step's features, so that a step is two launches instead of three
 * (agent.py:25-39,52-81 and model.py:90-105 for the next observation).  wt / w: the acting network's weight copies and encoder weights as for
 * a0_net_encoder_fwd_fused (4 x 84 x 84 observations only).  Same bytes and features as a0_actor_qhead_env_step + a0_net_encoder_fwd_fused. */
int a0_actor_qhead_env_step_enc(const float* feat, int E, int K, const float* W1, const float* b1, const float* W2, const float* b2, int A, int dueling,
                                float* scratch, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                                unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax,
                                unsigned long long env_seed, unsigned int rank, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                                float* final_mask, float* final_ret, int n, long long steps, double gamma, int* ring_act, float* ring_rew, float* ring_done,
                                const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act, float* r_rew, float* r_done, int task,
                                const float* wt, const a0_encoder_weights* w, float* act3_next, void* stream);
/* (round 5) a0_actor_dist_tail_env_step (c51 / qr) whose kernel goes on to encode the env's new observation into act3_next [E][3136], as a0_actor_qhead_env_step_enc
 * does for scalar heads: the same bytes and features as a0_actor_dist_tail_env_step + a0_net_encoder_fwd_fused, one launch and one kernel boundary less per step. */
int a0_actor_dist_tail_env_step_enc(const float* slabs, long long slab_stride, int nslab, const float* bias, int ld, int A, int T, int dueling, int mode,
                                    const float* atoms, int E, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                                    unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax,
                                    unsigned long long env_seed, unsigned int rank, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                                    float* final_mask, float* final_ret, int n, long long steps, double gamma, int* ring_act, float* ring_rew, float* ring_done,
                                    const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act, float* r_rew, float* r_done, int task,
                                    const float* wt, const a0_encoder_weights* w, float* act3_next, void* stream);
/* (round 6) the same for the quantile networks' tail (a0_actor_quantile_tail_env_step: iqn mode 1, fqf mode 3 with `taus` [E][T + 1]): tail + env step + replay row +
 * the new observation's encoder in one launch — an iqn actor step is then a0_tau_cos_features | embedding GEMM | fc1 GEMM | head GEMM | this. */
int a0_actor_quantile_tail_env_step_enc(const float* slabs, long long slab_stride, int nslab, const float* bias, int ld, int A, int T, int dueling, int mode,
                                        const float* taus, int E, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                                        unsigned long long off_u, float eps, const long long* ctrl, const float* eps_ptr, int* action, float* qmax,
                                        unsigned long long env_seed, unsigned int rank, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                                        float* final_mask, float* final_ret, int n, long long steps, double gamma, int* ring_act, float* ring_rew, float* ring_done,
                                        const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act, float* r_rew, float* r_done, int task,
                                        const float* wt, const a0_encoder_weights* w, float* act3_next, void* stream);
int a0_actor_egreedy_rng(const int* greedy, unsigned long long seed, unsigned int stream_a, unsigned int stream_u, unsigned long long off_a,
                         unsigned long long off_u, int A, float eps, int E, int* action, const float* qmax, float* qs_out,
                         const long long* ctrl, const float* eps_ptr, void* stream);
int a0_actor_nstep(int E, int n, long long steps, double gamma, const int* action, const float* reward, const float* terminal,
                   const float* truncated, const float* life_loss, int* ring_act, float* ring_rew, float* ring_done,
                   int* out_act, float* out_rew, float* out_done, const long long* ctrl, void* stream);

/* Device frame stack for host environments.  Replaces gymnasium FrameStack's host-side stacking plus the
 * per-step upload of the whole (E, nstack, H, W) batch (atari_wrappers.py:63, agent.py:27): out[e] = prev[e][1:] || newest[e] for envs
 * with advance[e] != 0 (their stack moved on by exactly one frame); rows with advance[e] == 0 are left untouched (uploaded whole).
 * prev / out: [E][nstack][frame_bytes] u8, newest: [E][frame_bytes]; frame_bytes a multiple of 16, pointers 16-byte aligned. */
int a0_env_frame_stack(const uint8_t* prev, const uint8_t* newest, const float* advance, uint8_t* out, int E, int nstack, long long frame_bytes,
                       void* stream);

/* The two PCIe legs of a host-environment step as one call each (the host thread's enqueue time is on the critical path between "the
 * workers have finished" and "the workers see the next actions").
 * a0_env_pool_upload: agent.py:27 `torch.from_numpy(obs).to(device)` — newest frames [E][frame_bytes] and scalars [n_scal][E] host -> device,
 *   whole stacks (obs_host rows [E][nstack][frame_bytes]) for the envs whose row `advance_row` of the HOST scalars is 0, then a0_env_frame_stack
 *   into `out` from `prev`.  Host pointers must be page-locked; *n_whole (optional) receives the number of whole stacks uploaded.
 * a0_env_pool_send: the direction AsyncVectorEnv.step_async pickles through pipes (atari_wrappers.py:59-69) — E actions, then the 8-byte step
 *   word, stored into page-locked host memory through its DEVICE address (a0_host_device_pointer) by two kernels in stream order: a worker that
 *   reads the new word reads the new actions.  No host-side wait in either call. */
int a0_host_device_pointer(void* host, void** dev);
int a0_env_pool_upload(const uint8_t* new_host, uint8_t* new_dev, const float* scal_host, float* scal_dev, int n_scal, int advance_row,
                       const uint8_t* obs_host, const uint8_t* prev, uint8_t* out, int E, int nstack, long long frame_bytes, int* n_whole,
                       void* stream);
int a0_env_pool_send(const int* action, int* act_host_dev, int E, long long* ctl_host_dev, long long word, void* stream);

/* ---------------------------------------------------------------- device RNG + synthetic env (no reference counterpart) */
int a0_rng_u32(unsigned long long seed, unsigned int stream_id, unsigned long long offset, unsigned int* out, long long n, void* stream);
int a0_rng_uniform(unsigned long long seed, unsigned int stream_id, unsigned long long offset, float* out, long long n, void* stream);
int a0_rng_randint(unsigned long long seed, unsigned int stream_id, unsigned long long offset, int hi, int* out, long long n, void* stream);
int a0_rng_normal(unsigned long long seed, unsigned int stream_id, unsigned long long offset, float stdv, float* out, long long n, void* stream);
int a0_rng_uniform_ctrl(unsigned long long seed, unsigned int stream_id, unsigned long long offset, float* out, long long n, const long long* ctrl,
                        int ctrl_idx, void* stream);
int a0_rng_normal_ctrl(unsigned long long seed, unsigned int stream_id, unsigned long long offset, float stdv, float* out, long long n,
                       const long long* ctrl, int ctrl_idx, void* stream);
/* Synthetic env (oracle/synth_env.c defines it).  `task` selects the reward: A0_ENV_TASK_STREAM = an action-independent stream (P(-1, +1, 0) = .05, .05, .9:
 * the bench workload), A0_ENV_TASK_BLOCK = learnable: +1 when the action equals the quadrant (mod A) of the bright 8x8 block in the newest frame of the
 * observation it was chosen on, -1 for the next class, 0 otherwise (chance level 0, optimum +1 per step).  Terminals / life losses do not depend on it. */
#define A0_ENV_TASK_STREAM 0
#define A0_ENV_TASK_BLOCK 1
/* (round 5) A0_ENV_TASK_CHASE = a task with TEMPORAL credit: the action MOVES the block on a 4 x 4 lattice (a % 4: up, down, left, right), +1 only on arrival at the
 * bottom-right cell, then a respawn three to six moves away; the env reads its state back from the newest frame of the observation (csrc/synth_env.h).  In the kernels
 * that merge the actor's tail with the env step the frame waves wait at a workgroup barrier for the action under this task (the other tasks' frames do not depend on
 * it and keep the barrier-free overlap). */
#define A0_ENV_TASK_CHASE 2
int a0_env_synth_reset(unsigned long long seed, unsigned int rank, int E, uint8_t* obs, float* ep_ret, void* stream);
int a0_env_synth_reset_task(unsigned long long seed, unsigned int rank, int E, uint8_t* obs, float* ep_ret, int task, void* stream);      /* (round 5) a0_env_synth_reset for `task`: the chase task's first frame shows the block at the env's start cell */
int a0_env_synth_step(unsigned long long seed, unsigned int rank, int E, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out,
                      float* ep_ret, float* reward, float* terminal, float* truncated, float* life_loss, float* final_mask,
                      float* final_ret, const int* action, int A, int task, const long long* ctrl, void* stream);
/* a0_env_synth_step + a0_actor_nstep (agent.py:57-73) + a0_replay_insert (agent.py:78-81, replay.py:45-53) in one launch for rollouts
 * driven by the synthetic env: obs0 = first observation of the emitted transition (obs_in for n = 1).  ctrl adds its ENV_STEP,
 * ACTOR_STEPS and REPLAY_SLOT words to g, steps and start_slot. */
int a0_env_synth_step_commit(unsigned long long seed, unsigned int rank, int E, unsigned int g, const uint8_t* obs_in, uint8_t* obs_out, float* ep_ret,
                             float* final_mask, float* final_ret, int n, long long steps, double gamma, const int* action, int* ring_act,
                             float* ring_rew, float* ring_done, const uint8_t* obs0, uint8_t* frames, long long cap, long long start_slot, int* r_act,
                             float* r_rew, float* r_done, int A, int task, const long long* ctrl, void* stream);

/* ---------------------------------------------------------------- data-parallel gradient exchange (no reference counterpart)
 * One process per GPU; every rank holds a full replica and SUM-reduces its flat fp32 gradient buffer once per update over RCCL/xGMI.  The
 * reference reduces its loss by SUM (agent.py:154), so a SUM all-reduce with Adam eps = 1e-2/(world*B) is one reference step on the global
 * batch.  a0_dp_allreduce is enqueued on the caller's stream (in place, asynchronous, capturable into the update's hipGraph).  Rendezvous:
 * rank 0 fills a 128-byte HOST blob with a0_dp_unique_id and hands it to the other ranks by its own means; every rank then calls a0_dp_init
 * (collective; binds to the calling thread's current device) and gets an opaque communicator handle, 0 on error. */
int a0_dp_unique_id(void* host_id128);
long long a0_dp_init(const void* host_id128, int rank, int world);
int a0_dp_allreduce(long long comm, float* buf, long long n, void* stream);
int a0_dp_destroy(long long comm);
/* (round 5) what RCCL reports for the communicator: host_out3 = {ncclCommCount, ncclCommUserRank, ncclCommCuDevice}; bench.py prints it beside an all-reduce of ones */
int a0_dp_info(long long comm, int* host_out3);

#ifdef __cplusplus
}
#endif
#endif /* AGENT0_HIP_H */
