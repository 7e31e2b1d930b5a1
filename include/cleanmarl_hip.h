/* cleanmarl_hip.h -- C-ABI of libcleanmarl_hip.so (gfx950 / MI355X only).
 *
 * The reference (AmineAndam04/cleanmarl) has NO operator / FFI interface: the whole learner is
 * inline code inside `if __name__ == "__main__":` of cleanmarl/mappo_multienvs.py (and the
 * ippo_ / *_lstm_ siblings).  Each entry point below replaces one inline program region of that
 * script; the region is cited as file:line.  INTEGRATION.md shows the ctypes stub a maintainer
 * of the reference would add at each of those program points.
 *
 * Conventions
 *   - every function returns 0 on success, <0 on error; cm_last_error() gives a thread-local text.
 *   - the CALLER owns all memory: arguments are raw DEVICE pointers (e.g. torch tensor.data_ptr()),
 *     explicit dims, and the hipStream_t to launch on (torch.cuda.current_stream().cuda_stream).
 *   - no hidden allocation, no implicit host sync: functions that need scratch take a caller-provided workspace sized by
 *     the matching *_workspace_bytes() query.  The only state the library keeps between calls are the process-wide options
 *     of cm_set_option (atomic ints, documented below), the measurement hooks cm_clock_probe / cm_prof_set_buffer, and
 *     nonces that tag hand-off words (no result depends on their value).
 *   - all device arithmetic is fp32; scalar hyper-parameters cross the ABI as double so that expressions the
 *     reference evaluates in Python float64 (e.g. 1 - td_lambda, 1 +- ppo_clip) are rounded to fp32 once.
 *     Device data layout (docs/KERNEL_NOTES.md §2), E envs, A agents, T steps:
 *       obs     float  [E][A][T][Do]      state  float [E][T][Ds]      reward float [E][T]
 *       avail   uint8  [E][A][T][K]       action int32 [E][A][T]       logp   float [E][A][T]
 *       values  float  [E][Av][T] (Av = 1 MAPPO / A IPPO)   ret, adv float [E][A][T]
 *       ep_len  int32  [E]   (mask[e][t] = t < ep_len[e]; reference b_mask is always a prefix)
 *     which is the axis permutation ref[b,t,a,f] == dev[e=b,a,t,f] of the reference batch
 *     (cleanmarl/mappo_multienvs.py:113-132).
 *   - network parameters are ONE flat fp32 buffer in torch `module.parameters()` order
 *       MLP: W0[H][Din] b0[H] {Wl[H][H] bl[H]} x L  Wout[Dout][H] bout[Dout]
 *            (cleanmarl/mappo_multienvs.py:160-170, 186-195)
 *       GRU actor: fc1.W[H][Din] fc1.b[H] W_ih[3H][H] W_hh[3H][H] b_ih[3H] b_hh[3H] fc2.W[K][H] fc2.b[K]
 *            (cleanmarl/mappo_lstm_multienvs.py:162-168)
 *     gradients use the same flat layout.
 */
#ifndef CLEANMARL_HIP_H
#define CLEANMARL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* cm_stream_t; /* hipStream_t */

/* statistics slots appended after the flat gradient by the *_fwd_bwd kernels (un-normalised sums) */
enum {
    CM_STAT_PG = 0,     /* sum_{e,t} mask * mean_a min(A*rho, A*clip(rho))      mappo_multienvs.py:538-546 */
    CM_STAT_ENT = 1,    /* sum mask * mean_a H(pi)                              :549-550 */
    CM_STAT_KL = 2,     /* sum mask * mean_a ((rho-1) - log rho)                :561-564 */
    CM_STAT_CLIP = 3,   /* sum mask * mean_a [|rho-1| > eps]                    :565-570 */
    CM_STAT_VLOSS = 4,  /* sum mask * mean_a (V - R)^2                          :554-558 */
    CM_STAT_COUNT = 5,  /* number of valid (env,t) pairs seen = b_mask.sum()    :572 */
    CM_NUM_STATS = 8
};

enum { CM_OPT_ADAM = 0, CM_OPT_ADAMW = 1, CM_OPT_SGD = 2, CM_OPT_RMSPROP = 3 };  /* RMSprop: beta2 := alpha (torch default 0.99) */

const char* cm_last_error(void);
int cm_version(void);  /* 101 since round 5 (cm_opt_step_t grew by stats_out; cm_optimizer_step_peer takes timeout_s / status; cm_clock_probe) */
/* Schedule / arithmetic options.  The library picks schedules from the problem size (thresholds measured on MI355X);
 * a caller can force one -- for A/B measurements, tests, or a box where the thresholds sit elsewhere.  The library never reads the
 * process environment.  Options are process-wide, take effect at the next launch and may be changed at any time
 * (value "auto" / the first value listed is the default):
 *   "mlp_forms"        auto | hand | loop      hand-ordered vs compiler-scheduled LDS reads of the fused MLP product loops
 *   "critic_schedule"  auto | fused | split | fused2   one-pass wide-input critic (csrc/cm_critic_fused.h) vs the two-kernel split schedule;
 *                                              fused2: the one-pass kernel with TWO row tiles per iteration for two-chunk inputs (opt-in: measured
 *                                              5 % slower than "fused" at config 4; same sums up to the association of a workgroup's tiles)
 *   "gru_tile"         auto | 64 | 32 | 8w | split | nosplit   auto: pipelined 32-row sweeps while tiles + helper workgroups fit the CUs (head / weight gradients
 *                                              on the idle CUs), eight-wave 32-row forward above, 64-row streaming sweeps from 512 64-row tiles;
 *                                              32: four-wave 32-row sweeps, 8w: eight-wave forward, 64: the 64-row sweeps at any batch size;
 *                                              the pipelined forward sweep is split at its dependence on h: helper waves run fc1 and the W_ih
 *                                              products a step ahead of the recurrence waves; split: the same split as a throughput launch in
 *                                              front of the chain (measured slower), nosplit: the unsplit sweep of round 5 -- all bit-identical
 *   "rollout_tile"     auto | 64 | 16 | 16s | 64s   tiling of the fused rollout (64 / 16: four-wave workgroups; 64s / 16s: four compute
 *                                              waves + a writer and a scorer wave, the defaults); the fused GRU rollout: 64 / 16 = its
 *                                              four-wave kernel, anything else = the six-wave one (bit-identical buffers)
 *   "mfma"             fp32 | bf16x3 | bf16    GEMM arithmetic of the PPO training passes: exact fp32 MFMA; error-compensated bf16
 *                                              (3 MFMAs per product, ~3e-6 of sum|a b|, inside the 1e-4 parity bar); single-pass bf16
 *                                              (1 MFMA per product, operands rounded to 8 bits: ~4e-3, its own looser parity tier);
 *                                              fp32 accumulate either way -- docs/KERNEL_NOTES.md section 8
 *   "wide_schedule"    auto | fused | layered | fused_r3
 *                                              MLPs of 65 .. 128 hidden units with one hidden->hidden layer (the reference's COMA critic
 *                                              default, cleanmarl/coma_multienvs.py:35): one-launch fused tile (csrc/cm_mlp_fused128.h, the
 *                                              default) vs the layer-by-layer schedule every wider / deeper shape runs (csrc/cm_mlp_wide.h);
 *                                              fused_r3 = the fused tile with round 3's forward kernel and COMA's separate S / z0 launches (A/B runs)
 *   "dw0_batch"        auto | 8 | 4            rows in flight per lane of the streaming layer-0 weight-gradient product of the split
 *                                              critic schedule (csrc/cm_mlp_split.h): 16 rows (234 registers) or 8 rows (fits one wave on
 *                                              every SIMD beside the six-wave rollout the two-stream schedule runs it under); auto = 8 rows
 *                                              up to 2^16 rows per launch.  Same summation order, bit-identical results.
 * cm_set_option returns 0, or -1 for an unknown key / value; cm_get_option returns the current value's name (NULL: unknown key). */
int cm_set_option(const char* key, const char* value);
const char* cm_get_option(const char* key);
/* current "mfma" option: 0 = fp32, 1 = bf16x3, 2 = bf16 */
int cm_mfma_mode(void);
/* A HIP stream of the LOWEST priority the device offers (hipStreamCreateWithPriority, non-blocking), for work that has slack and
 * should only fill compute units the caller's main stream leaves idle: the critic epochs of iteration i run on it under the rollout
 * of iteration i + 1 (cleanmarl_amd/learner.py).  The reference has no counterpart (single CPU thread, mappo_multienvs.py:521-594).
 * Returns NULL and sets cm_last_error() on failure; destroy with cm_stream_destroy. */
cm_stream_t cm_stream_create_low_priority(void);
int cm_stream_destroy(cm_stream_t stream);
/* number of floats in a flat MLP parameter buffer */
int64_t cm_mlp_param_count(int din, int hidden, int n_hidden_layers, int dout);
int64_t cm_gru_param_count(int din, int hidden, int dout);

/* ---- a3 / a5: Actor.logits / Critic.forward  (cleanmarl/mappo_multienvs.py:178-183, 197-200) ----
 * y[rows][dout] = MLP(x[rows][din]).  If avail != NULL (uint8 [rows][dout]) entries with avail==0 are
 * filled with -1e9 (masked_fill, :182).  Used for the value pass that feeds the TD(lambda) scan
 * (replaces the 2*E*T single-row critic calls at :492-504). */
int cm_mlp_forward(const float* x, int64_t rows, int din, int hidden, int n_hidden_layers, int dout,
                   const float* params, const uint8_t* avail, float* y, cm_stream_t stream);
/* The same with a caller workspace, which also covers the LAYERED schedule: hidden widths 65..256 (the reference's COMA critic
 * defaults to 128, cleanmarl/coma_multienvs.py:35) or more than two hidden->hidden layers run layer by layer with activations in
 * the workspace (csrc/cm_mlp_wide.h).  The query returns 0 for shapes the fused kernel covers (ws may then be NULL). */
size_t cm_mlp_forward_workspace_bytes(int64_t rows, int din, int hidden, int n_hidden_layers, int dout);
int cm_mlp_forward_ws(const float* x, int64_t rows, int din, int hidden, int n_hidden_layers, int dout,
                      const float* params, const uint8_t* avail, float* y, void* ws, size_t ws_bytes, cm_stream_t stream);

/* ---- a4: Actor.act  (cleanmarl/mappo_multienvs.py:172-176, called at :409-414) ----
 * Fused actor MLP + masked_fill + Categorical sample + log_prob for `rows` (env,agent) pairs.
 * x row r lives at x + r*x_row_stride floats (so the kernel can read obs[e][a][t][:] in place with
 * stride T*Do); outputs are written at action + r*out_stride, logp + r*out_stride (stride T writes
 * actions[e][a][t] in place).  Sampling: inverse-CDF with one Philox4x32-10 uniform keyed by
 * (seed, global_row = row_offset + r, t) -- identical for any GPU count. */
int cm_policy_act(const float* x, int64_t x_row_stride, const uint8_t* avail, int64_t avail_row_stride,
                  int64_t rows, int din, int hidden, int n_hidden_layers, int n_actions,
                  const float* params, uint64_t seed, int64_t row_offset, int t,
                  int32_t* action, float* logp, int64_t out_stride, cm_stream_t stream);
/* The three act entry points (cm_policy_act, cm_policy_act_eps, cm_policy_act_greedy) behind one with a caller workspace, which
 * also covers actors of the layered schedule (hidden 65..256, any depth, or 33..64 actions): eps == 0 samples Categorical(logits), eps in (0, 1]
 * samples COMA's mixture, eps < 0 takes the argmax.  Same Philox keying for every shape; the query is 0 for fused shapes. */
size_t cm_policy_act_workspace_bytes(int64_t rows, int din, int hidden, int n_hidden_layers, int n_actions);
int cm_policy_act_ws(const float* x, int64_t x_row_stride, const uint8_t* avail, int64_t avail_row_stride,
                     int64_t rows, int din, int hidden, int n_hidden_layers, int n_actions, const float* params,
                     double eps, uint64_t seed, int64_t row_offset, int t, int32_t* action, float* logp,
                     int64_t out_stride, void* ws, size_t ws_bytes, cm_stream_t stream);

/* Greedy variant (argmax of the masked logits, first maximum; logp of that action): the build's --greedy_eval option --
 * the reference's eval loop (cleanmarl/mappo_multienvs.py:614-650) always samples. */
int cm_policy_act_greedy(const float* x, int64_t x_row_stride, const uint8_t* avail, int64_t avail_row_stride,
                         int64_t rows, int din, int hidden, int n_hidden_layers, int n_actions, const float* params,
                         int32_t* action, float* logp, int64_t out_stride, cm_stream_t stream);
/* a4, whole episode in one launch: for envs whose observations do not depend on the actions (shape env) every step
 * can be sampled at once.  x [n_seq][T][din], avail [n_seq][T][n_actions] contiguous; row (s, t) uses the Philox key
 * (seed, row_offset + s, t), i.e. the draws of T calls of cm_policy_act with t = 0..T-1.  action / logp [n_seq][T]. */
int cm_policy_act_episode(const float* x, const uint8_t* avail, int64_t n_seq, int T, int din, int hidden,
                          int n_hidden_layers, int n_actions, const float* params, uint64_t seed, int64_t row_offset,
                          int32_t* action, float* logp, cm_stream_t stream);

/* ---- a6: TD(lambda) return + advantage  (cleanmarl/mappo_multienvs.py:484-504; ippo :483-503) ----
 * R_t = r_t + gamma*(lam*R_{t+1} + (1-lam)*V_{t+1}),  R = V = 0 beyond the last valid step,
 * A_t = R_t - V_t, zeros on padded steps.  Av = 1 broadcasts one value sequence to all A agents. */
int cm_td_lambda_scan(const float* reward, const float* values, const int32_t* ep_len,
                      int E, int A, int Av, int T, double gamma, double lam,
                      float* ret, float* adv, cm_stream_t stream);

/* ---- a2 / a7: masked moments + normalisation ----
 * Moments of the AGENT-MEAN y[e,t] = mean_a x[e][a][t] over the valid (e,t) pairs, as a float64 triple
 * out = (count, mean, M2 = sum (y - mean)^2) that env shards can merge Chan-style across GPUs before the
 * unbiased std is taken (cleanmarl/mappo_multienvs.py:143-146 with A = 1 for rewards, :505-512). */
size_t cm_masked_moments_workspace_bytes(int E, int A, int T);
int cm_masked_moments(const float* x, const int32_t* ep_len, int E, int A, int T,
                      double* out_count_mean_m2 /* device, 3 doubles */, void* ws, size_t ws_bytes,
                      cm_stream_t stream);
/* x = (x - mean) / (std_unbiased + eps) applied to valid entries only (valid_only=1, reward path :146)
 * or to every entry (valid_only=0, advantage / return path :508, :512).  count/mean/M2 are read from
 * device memory so no host sync is needed. */
int cm_normalize(float* x, const int32_t* ep_len, int E, int A, int T,
                 const double* count_mean_m2, float eps, int valid_only, cm_stream_t stream);

/* ---- a8 / a9: PPO actor loss forward + backward  (cleanmarl/mappo_multienvs.py:527-551, 561-582) ----
 * One full-batch pass over rows = E*A*T.  Writes per-workgroup partial gradients + statistics into
 * the workspace; cm_reduce_partials folds them into grad_and_stats[P + CM_NUM_STATS].
 * Gradients / statistics are UN-NORMALISED masked sums of the agent-mean (i.e. already divided by A
 * but not by N = b_mask.sum()); the division by the (global) N happens in cm_grad_norm_clip_adam so
 * that env-sharded ranks can all-reduce the buffer first. */
size_t cm_mlp_train_workspace_bytes(int din, int hidden, int n_hidden_layers, int dout);
/* rows-aware query (use this one): also sizes the layered schedule of wide / deep actors (hidden 65..256, any depth, or 33..64 actions) */
size_t cm_ppo_actor_workspace_bytes(int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions);
int cm_ppo_actor_fwd_bwd(const float* obs, const uint8_t* avail, const int32_t* action,
                         const float* logp_old, const float* adv, const int32_t* ep_len,
                         int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions,
                         const float* params, double ppo_clip, double entropy_coef,
                         float* grad_and_stats, void* ws, size_t ws_bytes, cm_stream_t stream);

/* ---- a8 / a9: critic MSE forward + backward  (cleanmarl/mappo_multienvs.py:554-558, 582) ----
 * per_agent = 0 (MAPPO): rows = E*T of x = state[E][T][din]; target mean over the A agents of ret.
 * per_agent = 1 (IPPO):  rows = E*A*T of x = obs[E][A][T][din]  (cleanmarl/ippo_multienvs.py:554). */
/* workspace query for cm_critic_fwd_bwd (wide inputs add a dZ0[rows][64] hand-off buffer for the streaming
 * layer-0 weight-gradient kernel, see csrc/cm_mlp_critic.hip) */
size_t cm_critic_workspace_bytes(int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers);
int cm_critic_fwd_bwd(const float* x, const float* ret, const int32_t* ep_len,
                      int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers,
                      const float* params, float* grad_and_stats, void* ws, size_t ws_bytes,
                      cm_stream_t stream);

/* ---- a10 / a11 / a12: norm_d + clip_grad_norm_ + Adam/AdamW step ----
 * (cleanmarl/mappo_multienvs.py:221-224, 584-594).  grad_and_stats holds P un-normalised gradient sums
 * followed by CM_NUM_STATS statistics; slot CM_STAT_COUNT is N.  The kernel scales the gradient by
 * grad_scale / N  (grad_scale = 1 for the MLP scripts; 1/T_chunk for the TBPTT chunk loss of
 * cleanmarl/mappo_lstm_multienvs.py:605-607), stores the pre-clip global L2 norm in out_norm[0],
 * clips to max_norm if max_norm > 0 (coef = min(1, max_norm/(norm+1e-6))), then applies one
 * Adam (weight_decay ignored) or AdamW (decoupled weight_decay) step with bias correction for `step`. */
int cm_grad_norm_clip_adam(float* params, float* grad_and_stats, float* exp_avg, float* exp_avg_sq,
                           int64_t n_params, int step, double lr, double beta1, double beta2, double eps,
                           double weight_decay, int opt_kind, double max_norm, double grad_scale,
                           float* out_norm, cm_stream_t stream);

/* ---- a9 tail + a10 / a11 / a12 fused BEHIND a training pass ("train_step" entry points) ----
 * With one process the gradient of a pass goes straight into that network's optimiser step (mappo_multienvs.py:572-594 without an
 * all-reduce in between): the *_train_step* variants of the three training passes take a cm_opt_step_t and fold the per-workgroup
 * partial gradients, scale by grad_scale / N, take the pre-clip norm and apply the update in ONE launch behind the pass (two when
 * max_norm > 0: the clip coefficient needs the global norm) instead of three.  Results are bit-identical to the pass followed by
 * cm_grad_norm_clip_adam.  grad_and_stats receives the scaled (and clipped) gradient and the un-normalised statistics as before.
 * scratch: caller-owned, cm_opt_step_scratch_bytes() bytes, 16-byte aligned, ZEROED ONCE when allocated and from then on touched
 * only by these launches (tagged hand-off words between the workgroups of one launch; never reset, never read by the host);
 * one scratch per optimiser if steps of different networks may overlap on different streams.
 * cm_optimizer_step is the same single launch for a gradient that is already reduced (e.g. all-reduced across ranks). */
typedef struct cm_opt_step {
    float* params;      /* [n_params] updated in place (the pass may read the same buffer: the update launch follows it) */
    float* exp_avg;     /* [n_params] Adam m            (unused by SGD / RMSprop) */
    float* exp_avg_sq;  /* [n_params] Adam v / RMSprop square_avg (unused by SGD) */
    float* out_norm;    /* [1] pre-clip global L2 norm of the scaled gradient */
    void* scratch;
    double lr, beta1, beta2, eps, weight_decay, max_norm, grad_scale;
    int32_t step;       /* 1-based optimiser step (bias correction) */
    int32_t opt_kind;   /* CM_OPT_* */
    float* stats_out;   /* optional [CM_NUM_STATS]: the un-normalised statistic sums (what follows the gradient in grad_and_stats) are ALSO
                           written here by the step launch -- e.g. one row of a per-epoch record buffer, so that no copy launch has to collect
                           them afterwards (mappo_multienvs.py:597-612's logged scalars); NULL: not written */
} cm_opt_step_t;
size_t cm_opt_step_scratch_bytes(void);
int cm_optimizer_step(float* grad_and_stats, int64_t n_params, const cm_opt_step_t* opt, cm_stream_t stream);
int cm_ppo_actor_train_step_ld(const float* obs, int64_t obs_ld, const uint8_t* avail, const int32_t* action,
                               const float* logp_old, const float* adv, const int32_t* ep_len,
                               int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions,
                               double ppo_clip, double entropy_coef, float* grad_and_stats, void* ws, size_t ws_bytes,
                               const cm_opt_step_t* opt, cm_stream_t stream);
int cm_critic_train_step_ld(const float* x, int64_t x_ld, const float* ret, const int32_t* ep_len,
                            int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers,
                            float* grad_and_stats, void* ws, size_t ws_bytes, const cm_opt_step_t* opt, cm_stream_t stream);

/* ---- measurement aid: the shader clock of the dominant kernel, read by the kernel itself ----
 * ticks != NULL (device memory, 4 x 512 words): workgroup w of every following cm_ppo_actor_fwd_bwd* / cm_ppo_actor_train_step* launch
 * (fused shapes; at most 512 workgroups) writes four 64-bit words to ticks[4 w ..]: t[0] = the shader cycles (s_memtime) of its tile
 * loop, t[1] = HW_REG_XCC_ID << 32 | HW_REG_HW_ID (where it ran), t[2] / t[3] = s_memrealtime at entry / exit of the loop (the constant
 * 100 MHz reference clock, common to all workgroups).  t[0] / ((t[3] - t[2]) / 1e8) is the shader clock the workgroup ran at; the spread
 * of t[2] / t[3] over the workgroups is the launch's ramp and tail.  bench.py reports roofline.shader_clock_ghz and roofline.workgroup_span from it.  NULL switches the probe off
 * (the default; an off probe costs one scalar compare). */
int cm_clock_probe(uint64_t* ticks);

/* ---- one-shot peer all-reduce of the [gradient | statistics] buffer (csrc/cm_peer.hip) ----
 * The exchange step of an env-sharded run without a collective library on the data path: every rank owns a MAILBOX (fine-grained device
 * memory; 2 x world slots of n floats + one tag word each) that its peers map with hipIpc.  Per optimiser step a rank
 *   1. reduces its pass's partials into grad_and_stats as usual (cm_*_fwd_bwd*),
 *   2. cm_peer_push: copies that buffer into slot [seq & 1][rank] of EVERY mailbox and then publishes the slot's tag {seq},
 *   3. cm_optimizer_step_peer: one launch that waits for the `world` tags of its OWN mailbox, folds the slots in rank order (every rank
 *      the same order: bit-identical parameters everywhere), scales by grad_scale / N, takes the norm and applies the update.
 *      The wait is bounded by WALL time (timeout_s seconds of the 100 MHz reference clock; <= 0: 30 s -- generous on purpose: a peer
 *      that is slow, e.g. writing a checkpoint, is healthy).  A wait that runs out SKIPS the step: parameters, optimiser state and
 *      grad_and_stats keep their values, out_norm becomes NaN and `seq` is written to *status (optional; a word in page-locked host
 *      memory or device memory, zero-initialised by the caller) -- the caller reads it on the host and stops the run.
 * seq: 1, 2, 3, ... per mailbox, the same on every rank (two slot sets alternate by its parity; a peer is never more than one step
 * ahead).  Setup (once): cm_peer_mailbox_alloc -> exchange the cm_peer_handle_bytes() handle bytes by any means (torch.distributed
 * all_gather_object in cleanmarl_amd/dist.py) -> cm_peer_mailbox_open per peer.  At most 16 ranks; ranks may share a device. */
size_t cm_peer_handle_bytes(void);
size_t cm_peer_mailbox_bytes(int world, int64_t n_floats);
int cm_peer_mailbox_alloc(size_t bytes, void** mailbox, void* handle_out);
int cm_peer_mailbox_open(const void* handle, void** mailbox);
int cm_peer_mailbox_close(void* mailbox);
int cm_peer_mailbox_free(void* mailbox);
int cm_peer_push(const float* buf, int64_t n_floats, int rank, int world, void* const* mailboxes, uint32_t seq, cm_stream_t stream);
int cm_optimizer_step_peer(float* grad_and_stats, int64_t n_params, void* own_mailbox, int world, uint32_t seq,
                           const cm_opt_step_t* opt, double timeout_s, uint32_t* status, cm_stream_t stream);

/* ---- a13 / a14: GRU actor, TBPTT chunk  (cleanmarl/mappo_lstm_multienvs.py:162-184, 562-620) ----
 * Forward + backward-through-time over steps [t0, t1) for all E*A sequences starting from the detached
 * hidden state h_in[E*A][H]; writes h_out (= h at t1, to be used detached for the next chunk) and the
 * flat gradient + statistics of the chunk loss  sum_{t in chunk} (-pg_t - c*ent_t)  (un-normalised). */
size_t cm_gru_workspace_bytes(int E, int A, int din, int hidden, int n_actions, int chunk_len);
int cm_gru_actor_chunk_fwd_bwd(const float* obs, const uint8_t* avail, const int32_t* action,
                               const float* logp_old, const float* adv, const int32_t* ep_len,
                               int E, int A, int T, int t0, int t1, int din, int hidden, int n_actions,
                               const float* params, const float* h_in, float* h_out,
                               double ppo_clip, double entropy_coef,
                               float* grad_and_stats, void* ws, size_t ws_bytes, cm_stream_t stream);
/* the chunk pass followed by the actor's optimiser step of mappo_lstm_multienvs.py:608-618 (grad_scale = 1 / chunk length), see cm_opt_step_t */
int cm_gru_actor_chunk_train_step(const float* obs, const uint8_t* avail, const int32_t* action,
                                  const float* logp_old, const float* adv, const int32_t* ep_len,
                                  int E, int A, int T, int t0, int t1, int din, int hidden, int n_actions,
                                  const float* h_in, float* h_out, double ppo_clip, double entropy_coef,
                                  float* grad_and_stats, void* ws, size_t ws_bytes, const cm_opt_step_t* opt, cm_stream_t stream);
/* single rollout step of the GRU actor (mappo_lstm_multienvs.py:170-174, 421-425): h updated in place */
int cm_gru_policy_act(const float* x, int64_t x_row_stride, const uint8_t* avail, int64_t avail_row_stride,
                      int64_t rows, int din, int hidden, int n_actions, const float* params, float* h,
                      uint64_t seed, int64_t row_offset, int t,
                      int32_t* action, float* logp, int64_t out_stride, cm_stream_t stream);

/* The GRU entry points above keep a 64-wide actor on <= 64 observation columns in registers / LDS (fused sweeps).  The reference's Actor
 * takes any input_dim / hidden_dim (mappo_lstm_multienvs.py:162-184): wider observations or 65..256 hidden units run a LAYERED schedule
 * (csrc/cm_gru_wide.hip: the h-independent parts batched over the chunk as GEMMs, one small GEMM group + one element-wise launch per
 * recurrent step) behind the SAME training entry points -- cm_gru_workspace_bytes sizes either -- and behind this workspace-taking
 * act entry point (the query is what the layered schedule needs; fused shapes accept ws = NULL unless eps < 0).
 * eps = 0 samples Categorical(logits) with the usual Philox keying, eps < 0 takes the argmax (the build's --greedy_eval). */
size_t cm_gru_policy_act_workspace_bytes(int64_t rows, int din, int hidden, int n_actions);
int cm_gru_policy_act_ws(const float* x, int64_t x_row_stride, const uint8_t* avail, int64_t avail_row_stride,
                         int64_t rows, int din, int hidden, int n_actions, const float* params, float* h, double eps,
                         uint64_t seed, int64_t row_offset, int t, int32_t* action, float* logp, int64_t out_stride,
                         void* ws, size_t ws_bytes, cm_stream_t stream);

/* ---- the counter RNG behind every sampler and synthetic env: Philox4x32-10 (Salmon et al., SC'11; Random123).  The reference draws from
 * torch's global generator (Categorical.sample, cleanmarl/mappo_multienvs.py:172-176) and re-seeds its workers from OS entropy (:251), so
 * its streams are not reproducible; this build keys every draw by (seed, global row, t, stream id).  These two entry points evaluate the
 * library's generator on explicit words -- ctr_key [n][6] = c0 c1 c2 c3 k0 k1, out [n][4] -- on the host and on the device (device
 * pointers), so that tests can pin it to the published known-answer vectors. */
int cm_philox4x32_host(const uint32_t* ctr_key, int64_t n, uint32_t* out);
int cm_philox4x32_device(const uint32_t* ctr_key, int64_t n, uint32_t* out, cm_stream_t stream);

/* ---- a15: on-device synthetic MPE-like environment (replaces the pipe round trips at
 * cleanmarl/mappo_multienvs.py:393-453 for the synthetic configs; CommonInterface semantics of
 * cleanmarl/env/common_interface.py:5-23 and the obs/state construction of
 * cleanmarl/env/pettingzoo_wrapper.py:93-98 for MPE simple_spread: Do = 6A (+A ids), Ds = 6A*A). */
/* env_state: float [E][6*A] = agent pos(2A) vel(2A) landmark pos(2A).  reset writes obs/state at t=0. */
int cm_synth_env_reset(float* env_state, int E, int A, int agent_ids, uint64_t seed, int64_t env_offset,
                       int64_t episode, float* obs, float* state, int T, cm_stream_t stream);
/* consumes action[e][a][t], writes reward[e][t] and (if t+1 < T) obs/state at t+1. */
int cm_synth_env_step(float* env_state, const int32_t* action, int E, int A, int agent_ids, int t, int T,
                      float* reward, float* obs, float* state, cm_stream_t stream);

/* ---- a15: "shape" env -- fixed-shape stand-in for envs with wide observations, a separate global state and
 * availability masks (SMAClite-like, BASELINE config 4).  obs ~ N(0,1) (+ one-hot ids), state ~ N(0,1),
 * avail ~ Bernoulli(avail_p) with action 0 always legal: all a pure function of (seed, env, episode, t), so one
 * launch fills the whole episode; the team reward (noise + fraction of agents choosing action t mod K) is
 * computed from the sampled actions afterwards.  CPU twin: cleanmarl_amd/env/synthetic.py::SyntheticShapeEnv. */
int cm_shape_env_fill(int E, int A, int T, int obs_raw, int agent_ids, int state_dim, int n_actions, double avail_p,
                      uint64_t seed, int64_t env_offset, int64_t episode, float* obs, float* state, uint8_t* avail,
                      cm_stream_t stream);
int cm_shape_env_reward(int E, int A, int T, int n_actions, uint64_t seed, int64_t env_offset, int64_t episode,
                        const int32_t* action, float* reward, cm_stream_t stream);

/* ---- a15 (fused): the WHOLE rollout of the synthetic env in one persistent launch ----
 * Equivalent to cm_synth_env_reset followed by T x (cm_policy_act; cm_synth_env_step) with the same seeds
 * (replaces cleanmarl/mappo_multienvs.py:393-453 + the collate of :109-157): each workgroup keeps floor(64/A)
 * envs in LDS for all T steps, actor weights LDS-stationary.  Supported when 6A(+A) <= 64, hidden <= 64 and
 * n_hidden_layers <= 1 (query with cm_rollout_spread_supported); other shapes use the per-step entry points.
 * act_seed keys the action sampler (the per-step path passes the same value as `seed` of cm_policy_act). */
int cm_rollout_spread_supported(int A, int agent_ids, int hidden, int n_hidden_layers);
int cm_rollout_spread(float* env_state, int E, int A, int T, int agent_ids, uint64_t seed, uint64_t act_seed,
                      int64_t env_offset, int64_t episode, const float* params, int hidden, int n_hidden_layers,
                      float* obs, float* state, int32_t* action, float* logp, float* reward, cm_stream_t stream);

/* The same fused rollout for the GRU actor (cleanmarl/mappo_lstm_multienvs.py:392-479 on the synthetic configs): equivalent to
 * cm_synth_env_reset + T x (cm_gru_policy_act with h = 0 at t = 0; cm_synth_env_step) with the same seeds; 32-row tiles,
 * all GRU weight blocks LDS-resident for the whole episode.  Supported when A <= 32, 6A(+A) <= 64, hidden <= 64. */
int cm_gru_rollout_spread_supported(int A, int agent_ids, int hidden);
int cm_gru_rollout_spread(float* env_state, int E, int A, int T, int agent_ids, uint64_t seed, uint64_t act_seed,
                          int64_t env_offset, int64_t episode, const float* params, int hidden,
                          float* obs, float* state, int32_t* action, float* logp, float* reward, cm_stream_t stream);
/* ... with a row stride for the state buffer (state_ld >= 6 A A floats; the observations stay contiguous: the recurrent kernels read them
 * as such).  A stride that is a multiple of 4 gives the critic's passes (cm_critic_fwd_bwd_ld, cm_mlp_forward_ld) 16-byte aligned rows:
 * 150 -> 152 floats at 5 agents, where the unpadded rows cost the critic epoch 160 us instead of 111 us (scalar loads). */
int cm_gru_rollout_spread_ld(float* env_state, int E, int A, int T, int agent_ids, uint64_t seed, uint64_t act_seed,
                             int64_t env_offset, int64_t episode, const float* params, int hidden,
                             float* obs, float* state, int64_t state_ld, int32_t* action, float* logp, float* reward, cm_stream_t stream);

/* ---- padded leading dimensions ("_ld" variants) -----------------------------------------------------------------------
 * The reference's feature widths are whatever the env gives (21, 35, 115 at BASELINE configs 2 / 5 / 4): rows of obs [E][A][T][Do]
 * are then not 16-byte aligned and every tile load of the MLP kernels falls back to 4-byte loads (config 4: the actor pass at 47 %
 * of the fp32 MFMA peak instead of 58 %).  These variants take the LEADING DIMENSION of the feature axis separately from its
 * width: obs [E][A][T][obs_ld], state [E][T][state_ld] with obs_ld >= Do, state_ld >= Ds; a leading dimension that is a multiple of 4
 * (>= the width rounded up to 4) with ZERO padding columns puts the input rows on 16-byte loads.  The weights keep the reference's
 * layout (W0 is [H][din]).  x_ld == din is the unpadded entry point.  Rollout kernels write the padding columns of a step's row as
 * zeros or leave them untouched (allocate the buffers zeroed).  This permutes / pads only the storage of the batch the reference
 * builds at cleanmarl/mappo_multienvs.py:113-132. */
int cm_mlp_forward_ld(const float* x, int64_t x_ld, int64_t rows, int din, int hidden, int n_hidden_layers, int dout,
                      const float* params, const uint8_t* avail, float* y, void* ws, size_t ws_bytes, cm_stream_t stream);
/* ws / ws_bytes: optional scratch of cm_w0_image_bytes(din, hidden) bytes (16-byte aligned).  When the input is wider than 64 columns
 * W0 is streamed per tile; if its rows (stride din) are not 16-byte aligned the entry points copy it once per call into a zero-padded
 * image with an aligned leading dimension there (the training passes keep theirs in their workspace).  NULL / too small: W0 chunks
 * stay on 4-byte loads -- slower, same results. */
size_t cm_w0_image_bytes(int din, int hidden);
/* cm_mlp_forward_ld for a launch the CALLER has ordered behind everything else on the device -- the value pass V(s_t) of
 * cleanmarl/mappo_multienvs.py:492-504 at the head of an update, after the join with the critic's stream.  Same arguments, same results;
 * the hint lets a full persistent grid take the unequal static tile split whatever its row count (beside another stream's kernels that
 * split costs 15 - 25 %, which is why the generic entry point only splits launches of >= 2^21 rows). */
int cm_mlp_forward_solo_ld(const float* x, int64_t x_ld, int64_t rows, int din, int hidden, int n_hidden_layers, int dout,
                           const float* params, const uint8_t* avail, float* y, void* ws, size_t ws_bytes, cm_stream_t stream);
int cm_policy_act_episode_ld(const float* x, int64_t x_ld, const uint8_t* avail, int64_t n_seq, int T, int din, int hidden,
                             int n_hidden_layers, int n_actions, const float* params, uint64_t seed, int64_t row_offset,
                             int32_t* action, float* logp, void* ws, size_t ws_bytes, cm_stream_t stream);
/* MFMA work the fused actor pass issues per row (flop, tile padding included; 0 for shapes on the layered schedule): reported by
 * bench.py beside the algorithmic flop of SURVEY.md 8(d) so that the padding share of the matrix-pipe time is visible.  The figure is
 * that of the instantiation a launch of >= 2^21 rows takes (single-chunk inputs, <= 8 actions: the head on the 4x4x1 MFMA, 8 padded
 * head columns: 3 072 flop per row for the head); smaller launches of those shapes run the 16x16x4 head (16 padded columns: 5 120 flop per row). */
double cm_ppo_actor_issued_flop_per_row(int din, int hidden, int n_hidden_layers, int n_actions);
int cm_ppo_actor_fwd_bwd_ld(const float* obs, int64_t obs_ld, const uint8_t* avail, const int32_t* action,
                            const float* logp_old, const float* adv, const int32_t* ep_len,
                            int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions,
                            const float* params, double ppo_clip, double entropy_coef,
                            float* grad_and_stats, void* ws, size_t ws_bytes, cm_stream_t stream);
/* cm_critic_fwd_bwd_ld picks one of three schedules with the same sums (summation order differs; parity tests at 1e-4): the fused tile
 * kernel (din <= 128), the two-kernel split schedule (wider inputs) and -- for 65 .. 448 input columns on 16-byte aligned rows, one
 * hidden layer, from 131072 rows on -- the one-pass kernel of csrc/cm_critic_fused.h, which reads x from HBM once.  The padding columns
 * [din, x_ld) must hold FINITE values (the library's own rollouts write zeros): they meet zero weights, never a mask.
 * cm_set_option("critic_schedule", "fused" | "split") forces the one-pass / two-kernel schedule. */
int cm_critic_fwd_bwd_ld(const float* x, int64_t x_ld, const float* ret, const int32_t* ep_len,
                         int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers,
                         const float* params, float* grad_and_stats, void* ws, size_t ws_bytes, cm_stream_t stream);
/* ---- the critic's first epoch without its layer-0 product (round 6) ----
 * The value pass at the head of an update (cleanmarl/mappo_multienvs.py:484-504) and the first critic epoch (:554-558) evaluate the critic on the same
 * rows with the same parameters.  cm_value_pass_keep_h0_ld = cm_mlp_forward_solo_ld with one output and no mask that ALSO leaves
 * h0 = relu(x W0^T + b0) in h0_out ([rows][64] floats, caller-owned; columns >= hidden are 0; fused-kernel shapes only: hidden <= 64);
 * cm_critic_fwd_bwd_h0_ld / cm_critic_train_step_h0_ld = cm_critic_fwd_bwd_ld / cm_critic_train_step_ld for THAT epoch: the one-pass schedule then
 * reads h0 instead of multiplying x by W0 (39 % of its time at a 384-wide state), the other schedules ignore it.  The caller guarantees that params
 * have not changed since the value pass; later epochs use the plain entry points.  Same sums (h0 comes from another product form: results agree to
 * rounding). */
int cm_value_pass_keep_h0_ld(const float* x, int64_t x_ld, int64_t rows, int din, int hidden, int n_hidden_layers,
                             const float* params, float* y, float* h0_out, void* ws, size_t ws_bytes, cm_stream_t stream);
int cm_critic_fwd_bwd_h0_ld(const float* x, int64_t x_ld, const float* h0, const float* ret, const int32_t* ep_len,
                            int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers,
                            const float* params, float* grad_and_stats, void* ws, size_t ws_bytes, cm_stream_t stream);
int cm_critic_train_step_h0_ld(const float* x, int64_t x_ld, const float* h0, const float* ret, const int32_t* ep_len,
                               int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers,
                               float* grad_and_stats, void* ws, size_t ws_bytes, const cm_opt_step_t* opt, cm_stream_t stream);
/* eps = 0: cm_rollout_spread; eps in (0, 1]: cm_rollout_spread_eps; eps < 0: greedy -- every agent takes the first maximal masked logit
 * (the evaluation rollouts of --greedy_eval: cleanmarl/mappo_multienvs.py:614-650 as ONE launch over num_eval_ep environments) */
int cm_rollout_spread_ld(float* env_state, int E, int A, int T, int agent_ids, uint64_t seed, uint64_t act_seed,
                         int64_t env_offset, int64_t episode, const float* params, int hidden, int n_hidden_layers, double eps,
                         float* obs, int64_t obs_ld, float* state, int64_t state_ld, int32_t* action, float* logp, float* reward,
                         cm_stream_t stream);
int cm_shape_env_fill_ld(int E, int A, int T, int obs_raw, int agent_ids, int state_dim, int n_actions, double avail_p,
                         uint64_t seed, int64_t env_offset, int64_t episode, float* obs, int64_t obs_ld, float* state,
                         int64_t state_ld, uint8_t* avail, cm_stream_t stream);

/* ---- COMA  (cleanmarl/coma_multienvs.py, cleanmarl/coma.py) -----------------------------------
 * Device layouts as above: obs [E][A][T][Do], state [E][T][Ds], action [E][A][T] int32, avail [E][A][T][K] u8,
 * reward [E][T], ep_len [E]; K-output tensors are [E][A][T][K]. */
/* Actor.act with exploration (coma_multienvs.py:177-186): probs = (1-eps) softmax(masked logits) + eps * avail / n_avail;
 * same Philox keying as cm_policy_act; logp = log(probs[action]). */
int cm_policy_act_eps(const float* x, int64_t x_row_stride, const uint8_t* avail, int64_t avail_row_stride,
                      int64_t rows, int din, int hidden, int n_hidden_layers, int n_actions,
                      const float* params, double eps, uint64_t seed, int64_t row_offset, int t,
                      int32_t* action, float* logp, int64_t out_stride, cm_stream_t stream);
/* cm_rollout_spread with the epsilon-mixed policy (whole COMA rollout of the synthetic env in one persistent launch). */
int cm_rollout_spread_eps(float* env_state, int E, int A, int T, int agent_ids, uint64_t seed, uint64_t act_seed,
                          int64_t env_offset, int64_t episode, const float* params, int hidden, int n_hidden_layers,
                          double eps, float* obs, float* state, int32_t* action, float* logp, float* reward, cm_stream_t stream);
/* Critic.coma_inputs (coma_multienvs.py:222-240): out [E][A][T][Ds + Do + (A-1)K] = state | own obs | one-hot actions of
 * the other agents in agent order.  The Q network itself is cm_mlp_forward on these rows (dout = K; pass avail to get
 * the masked_fill(~avail, -1e9) of the TARGET-critic calls at :565-570 / :590-595). */
int cm_coma_build_inputs(const float* state, const float* obs, const int32_t* action, int E, int A, int T, int Ds, int Do,
                         int n_actions, float* out, cm_stream_t stream);
/* torch.gather(q, -1, action) (coma_multienvs.py:571-575, 596-600): out[row] = q[row][action[row]]. */
int cm_gather_taken(const float* q, const int32_t* action, int64_t rows, int n_actions, float* out, cm_stream_t stream);
/* n-step targets (coma_multienvs.py:581-613): sum_{i<n} gamma^i r_{t+i} + gamma^n qtaken[t+n] if t < len-n, else the
 * discounted reward-to-go; 0 on padded steps.  (The TD(lambda) targets of :556-580 are cm_td_lambda_scan with
 * values := qtaken, Av = A.) */
int cm_nstep_returns(const float* reward, const float* qtaken, const int32_t* ep_len, int E, int A, int T, double gamma,
                     int nsteps, float* ret, cm_stream_t stream);
/* Workspace of the two-schedule fused training passes below (fused for din <= 128, else fused + streaming dW0). */
size_t cm_mlp_split_workspace_bytes(int64_t rows, int din, int hidden, int n_hidden_layers, int dout);
/* Critic step (coma_multienvs.py:620-631): MSE between Q[taken action] and target over valid rows, agent-mean /
 * env-sum like a8; x = cm_coma_build_inputs rows.  grad_and_stats[P + 8]: un-normalised gradient + CM_STAT_VLOSS/COUNT. */
int cm_qcritic_fwd_bwd(const float* x, const int32_t* action, const float* target, const int32_t* ep_len, int E, int A, int T,
                       int din, int hidden, int n_hidden_layers, int n_actions, const float* params, float* grad_and_stats,
                       void* ws, size_t ws_bytes, cm_stream_t stream);
/* Counterfactual advantage (coma_multienvs.py:657-663): adv = q[a] - sum_k softmax(logits)_k q_k from the actor's masked
 * logits and the critic's Q (both [E][A][T][K], via cm_mlp_forward), plus per-time-step raw sums
 * tstats[T][4] = {n valid rows, sum adv, sum adv^2, sum of action indices over ALL rows} in float64 (all-reduce(sum) them
 * across ranks); cm_coma_normalize_adv then applies (adv - mean_t) / (std_t + 1e-8) (unbiased std) on the steps where the
 * reference's condition `b_actions[:, t].sum() > n_agents` (:664) holds. */
size_t cm_coma_advantage_workspace_bytes(int E, int A, int T);
int cm_coma_advantage(const float* logits, const float* q, const int32_t* action, const int32_t* ep_len, int E, int A, int T,
                      int n_actions, float* adv, double* tstats, void* ws, size_t ws_bytes, cm_stream_t stream);
int cm_coma_normalize_adv(float* adv, const double* tstats, int E, int A, int T, cm_stream_t stream);
/* Actor step (coma_multienvs.py:649-676, eps = 0): row loss -log(pi_a + 1e-8) * adv - c * (-(pi log(pi + 1e-8)).mean_k),
 * summed over valid rows of ALL agents.  Stats: CM_STAT_PG = sum log(pi_a + 1e-8) * adv, CM_STAT_ENT = sum of the K-mean
 * entropies, CM_STAT_COUNT = N. */
int cm_coma_actor_fwd_bwd(const float* obs, const uint8_t* avail, const int32_t* action, const float* adv,
                          const int32_t* ep_len, int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions,
                          const float* params, double entropy_coef, float* grad_and_stats, void* ws, size_t ws_bytes,
                          cm_stream_t stream);
/* The same Q network WITHOUT materialising the critic input: W0 x = W0o obs + (W0s state[e,t] + gathered action columns),
 * so the fused kernel runs on the obs block with a per-row addend and the state / action blocks of dW0 come from two
 * streaming GEMMs over dZ0 (see csrc/cm_coma.hip).  params / grad_and_stats use the torch parameter order of
 * Critic(input_dim = Ds + Do + (A-1)K) (coma_multienvs.py:194-208), exactly like cm_mlp_forward / cm_qcritic_fwd_bwd. */
size_t cm_coma_critic_workspace_bytes(int E, int A, int T, int Ds, int Do, int n_actions, int hidden, int n_hidden_layers,
                                      int train);
int cm_coma_q_forward(const float* state, const float* obs, const int32_t* action, const uint8_t* avail, int E, int A, int T,
                      int Ds, int Do, int n_actions, int hidden, int n_hidden_layers, const float* params, float* q, void* ws,
                      size_t ws_bytes, cm_stream_t stream);
int cm_coma_critic_fwd_bwd(const float* state, const float* obs, const int32_t* action, const float* target,
                           const int32_t* ep_len, int E, int A, int T, int Ds, int Do, int n_actions, int hidden,
                           int n_hidden_layers, const float* params, float* grad_and_stats, void* ws, size_t ws_bytes,
                           cm_stream_t stream);
/* "_ld" variants of the three COMA entry points that read observations / states: row strides obs_ld >= Do and state_ld >= Ds (floats) instead
 * of contiguous rows, so that COMA runs on the padded buffers of the device rollouts (cleanmarl_amd.learner.DeviceBatch: leading dimensions
 * rounded up to 4 floats -- 16-byte aligned rows for every kernel of the path, and the six-wave rollout instead of the four-wave one; at the
 * reference's default 3-agent simple_spread shapes the rows are 21 / 54 floats).  Padding columns must hold finite values (zeros wherever the
 * library writes them).  The plain entry points above are these with obs_ld = Do, state_ld = Ds. */
int cm_coma_q_forward_ld(const float* state, int64_t state_ld, const float* obs, int64_t obs_ld, const int32_t* action, const uint8_t* avail,
                         int E, int A, int T, int Ds, int Do, int n_actions, int hidden, int n_hidden_layers, const float* params, float* q,
                         void* ws, size_t ws_bytes, cm_stream_t stream);
int cm_coma_critic_fwd_bwd_ld(const float* state, int64_t state_ld, const float* obs, int64_t obs_ld, const int32_t* action, const float* target,
                              const int32_t* ep_len, int E, int A, int T, int Ds, int Do, int n_actions, int hidden, int n_hidden_layers,
                              const float* params, float* grad_and_stats, void* ws, size_t ws_bytes, cm_stream_t stream);
int cm_coma_actor_fwd_bwd_ld(const float* obs, int64_t obs_ld, const uint8_t* avail, const int32_t* action, const float* adv,
                             const int32_t* ep_len, int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions,
                             const float* params, double entropy_coef, float* grad_and_stats, void* ws, size_t ws_bytes, cm_stream_t stream);

/* soft_update (coma_multienvs.py:266-270): target = polyak * src + (1 - polyak) * target. */
int cm_polyak_update(float* target, const float* src, int64_t n, double polyak, cm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CLEANMARL_HIP_H */
