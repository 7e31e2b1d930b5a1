// cm_mlp_infer.hip -- C-ABI entry points of the forward-only MLP kernels (a3/a4/a5)
#include "cm_mlp_wide.h"
#ifdef CM_PHASE_PROF
extern unsigned long long* g_prof;
#define CM_SET_PROF(a) (a).prof = g_prof
#else
#define CM_SET_PROF(a)
#endif

static int mlp_forward_ld(const float* x, int64_t x_ld, int64_t rows, int din, int hidden, int n_hidden_layers, int dout,
                          const float* params, const uint8_t* avail, float* y, cm_stream_t stream, void* ws = nullptr, size_t ws_bytes = 0,
                          bool solo = false, float* h0_out = nullptr) {
    if (int rc = check_shapes("cm_mlp_forward", din, hidden, n_hidden_layers, dout)) return rc;
    CM_REQUIRE(x_ld >= din, "cm_mlp_forward: leading dimension %lld < din %d", (long long)x_ld, din);
    if (rows <= 0) return 0;
    MlpArgs a = {};
    a.x = x; a.x_stride = x_ld; a.rows = rows; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = dout;
    a.params = params; a.avail = avail; a.avail_stride = dout; a.y = y;
    a.dz0 = h0_out;  // forward kernels: optional [rows][64] copy of the layer-0 activations (cm_value_pass_keep_h0_ld)
    prep_w0_image(a, (float*)ws, ws_bytes / sizeof(float), (hipStream_t)stream);  // no workspace: W0 chunks on 4-byte loads where unaligned
    const size_t lds_bytes = (size_t)make_lds(a.L, a.dout, (a.din + KC - 1) / KC).total * sizeof(float);
    // forward kernels run two workgroups per CU as well (same two-group finish, cm_mlp_kernel.h): the unequal split pays for a launch that
    // has the GPU to itself and costs 15 - 25 % beside another stream's kernels -- only the caller knows which (cm_mlp_forward_solo_ld:
    // the learner's value pass, ordered behind the critic stream); the generic entry points split by size like every other launch
    set_tile_split(a, grid_for(rows), 0, solo);
    launch_infer<M_FWD>(a, grid_for(rows), lds_bytes, (hipStream_t)stream);
    CM_CHECK_LAUNCH("cm_mlp_forward");
    return 0;
}
extern "C" int cm_mlp_forward(const float* x, int64_t rows, int din, int hidden, int n_hidden_layers, int dout,
                              const float* params, const uint8_t* avail, float* y, cm_stream_t stream) {
    return mlp_forward_ld(x, din, rows, din, hidden, n_hidden_layers, dout, params, avail, y, stream);
}

/* cm_mlp_forward with a caller workspace: also covers the shapes of the layered schedule (hidden 65..256, any depth), whose
 * activations live in the workspace.  cm_mlp_forward_workspace_bytes is 0 for shapes the fused kernel covers. */
extern "C" size_t cm_mlp_forward_workspace_bytes(int64_t rows, int din, int hidden, int n_hidden_layers, int dout) {
    return wide_shape(hidden, n_hidden_layers, dout) ? wide_ws_bytes((long)rows, din, hidden, n_hidden_layers, dout, false)
                                              : w0_image_floats(din, hidden) * sizeof(float);  // optional: the padded W0 image of the _ld form
}
/* bytes of the optional scratch of the *_ld inference entry points (padded W0 image; 0 when W0 is not streamed or its rows are aligned) */
extern "C" size_t cm_w0_image_bytes(int din, int hidden) { return w0_image_floats(din, hidden) * sizeof(float); }

extern "C" int cm_mlp_forward_ld(const float* x, int64_t x_ld, int64_t rows, int din, int hidden, int n_hidden_layers, int dout,
                                 const float* params, const uint8_t* avail, float* y, void* ws, size_t ws_bytes, cm_stream_t stream) {
    if (!wide_shape(hidden, n_hidden_layers, dout)) return mlp_forward_ld(x, x_ld, rows, din, hidden, n_hidden_layers, dout, params, avail, y, stream, ws, ws_bytes);
    CM_REQUIRE(x_ld >= din, "cm_mlp_forward_ld: leading dimension %lld < din %d", (long long)x_ld, din);
    if (rows <= 0) return 0;
    MlpArgs a = {};
    a.x = x; a.x_stride = x_ld; a.rows = rows; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = dout;
    a.params = params; a.avail = avail; a.avail_stride = dout; a.y = y;
    return wide_forward(a, ws, ws_bytes, (hipStream_t)stream, "cm_mlp_forward_ws");
}
/* cm_mlp_forward_ld for a launch the caller has ordered behind everything else on the device (the value pass of
 * cleanmarl/mappo_multienvs.py:492-504 at the head of the update, after the join with the critic stream): same results, the full grid may
 * take the unequal static tile split (set_tile_split) whatever its row count. */
extern "C" int cm_mlp_forward_solo_ld(const float* x, int64_t x_ld, int64_t rows, int din, int hidden, int n_hidden_layers, int dout,
                                      const float* params, const uint8_t* avail, float* y, void* ws, size_t ws_bytes, cm_stream_t stream) {
    if (wide_shape(hidden, n_hidden_layers, dout)) return cm_mlp_forward_ld(x, x_ld, rows, din, hidden, n_hidden_layers, dout, params, avail, y, ws, ws_bytes, stream);
    return mlp_forward_ld(x, x_ld, rows, din, hidden, n_hidden_layers, dout, params, avail, y, stream, ws, ws_bytes, true);
}
/* The value pass of an update (cm_mlp_forward_solo_ld with one output and no availability mask) that also leaves the layer-0 activations
 * h0 = relu(x W0^T + b0) of every row in h0_out ([rows][64] floats, columns >= hidden are 0) for the first critic epoch of the same update
 * (cm_critic_fwd_bwd_h0_ld: same parameters, same rows).  Shapes of the fused kernels only (hidden <= 64). */
extern "C" int cm_value_pass_keep_h0_ld(const float* x, int64_t x_ld, int64_t rows, int din, int hidden, int n_hidden_layers,
                                        const float* params, float* y, float* h0_out, void* ws, size_t ws_bytes, cm_stream_t stream) {
    CM_REQUIRE(!wide_shape(hidden, n_hidden_layers, 1), "cm_value_pass_keep_h0_ld: hidden %d / %d layers run on the layered schedule, which keeps no h0", hidden, n_hidden_layers);
    CM_REQUIRE(h0_out != nullptr, "cm_value_pass_keep_h0_ld: h0_out is NULL");
    return mlp_forward_ld(x, x_ld, rows, din, hidden, n_hidden_layers, 1, params, nullptr, y, stream, ws, ws_bytes, true, h0_out);
}
extern "C" int cm_mlp_forward_ws(const float* x, int64_t rows, int din, int hidden, int n_hidden_layers, int dout,
                                 const float* params, const uint8_t* avail, float* y, void* ws, size_t ws_bytes, cm_stream_t stream) {
    return cm_mlp_forward_ld(x, din, rows, din, hidden, n_hidden_layers, dout, params, avail, y, ws, ws_bytes, stream);
}

extern "C" int cm_policy_act(const float* x, int64_t x_row_stride, const uint8_t* avail, int64_t avail_row_stride,
                             int64_t rows, int din, int hidden, int n_hidden_layers, int n_actions,
                             const float* params, uint64_t seed, int64_t row_offset, int t,
                             int32_t* action, float* logp, int64_t out_stride, cm_stream_t stream) {
    if (int rc = check_shapes("cm_policy_act", din, hidden, n_hidden_layers, n_actions)) return rc;
    if (rows <= 0) return 0;
    MlpArgs a = {};
    a.x = x; a.x_stride = x_row_stride; a.rows = rows; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = n_actions;
    a.params = params; a.avail = avail; a.avail_stride = avail_row_stride;
    a.seed = seed; a.row_offset = row_offset; a.t = t; a.action_out = action; a.logp_out = logp; a.out_stride = out_stride;
    const size_t lds_bytes = (size_t)make_lds(a.L, a.dout, (a.din + KC - 1) / KC).total * sizeof(float);
    set_tile_split(a, grid_for(rows), 0);  // forward kernels run two workgroups per CU as well: same two-group finish (cm_mlp_kernel.h)
    launch_infer<M_ACT>(a, grid_for(rows), lds_bytes, (hipStream_t)stream);
    CM_CHECK_LAUNCH("cm_policy_act");
    return 0;
}

/* COMA's Actor.act (cleanmarl/coma_multienvs.py:177-186): sample from (1 - eps) * softmax + eps * uniform over the
 * available actions.  Same Philox keying as cm_policy_act; logp = log of the MIXED probability of the sampled action. */
extern "C" int cm_policy_act_eps(const float* x, int64_t x_row_stride, const uint8_t* avail, int64_t avail_row_stride,
                                 int64_t rows, int din, int hidden, int n_hidden_layers, int n_actions,
                                 const float* params, double eps, uint64_t seed, int64_t row_offset, int t,
                                 int32_t* action, float* logp, int64_t out_stride, cm_stream_t stream) {
    if (int rc = check_shapes("cm_policy_act_eps", din, hidden, n_hidden_layers, n_actions)) return rc;
    CM_REQUIRE(eps >= 0.0 && eps <= 1.0, "cm_policy_act_eps: eps=%g outside [0, 1]", eps);
    if (rows <= 0) return 0;
    MlpArgs a = {};
    a.x = x; a.x_stride = x_row_stride; a.rows = rows; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = n_actions;
    a.params = params; a.avail = avail; a.avail_stride = avail_row_stride; a.act_eps = (float)eps;
    a.seed = seed; a.row_offset = row_offset; a.t = t; a.action_out = action; a.logp_out = logp; a.out_stride = out_stride;
    const size_t lds_bytes = (size_t)make_lds(a.L, a.dout, (a.din + KC - 1) / KC).total * sizeof(float);
    set_tile_split(a, grid_for(rows), 0);  // forward kernels run two workgroups per CU as well: same two-group finish (cm_mlp_kernel.h)
    launch_infer<M_ACT>(a, grid_for(rows), lds_bytes, (hipStream_t)stream);
    CM_CHECK_LAUNCH("cm_policy_act_eps");
    return 0;
}

/* Greedy action (argmax of the masked logits, first maximum) + its log-probability: the build's --greedy_eval option. */
extern "C" int cm_policy_act_greedy(const float* x, int64_t x_row_stride, const uint8_t* avail, int64_t avail_row_stride,
                                    int64_t rows, int din, int hidden, int n_hidden_layers, int n_actions, const float* params,
                                    int32_t* action, float* logp, int64_t out_stride, cm_stream_t stream) {
    if (int rc = check_shapes("cm_policy_act_greedy", din, hidden, n_hidden_layers, n_actions)) return rc;
    if (rows <= 0) return 0;
    MlpArgs a = {};
    a.x = x; a.x_stride = x_row_stride; a.rows = rows; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = n_actions;
    a.params = params; a.avail = avail; a.avail_stride = avail_row_stride; a.act_eps = -1.0f;
    a.action_out = action; a.logp_out = logp; a.out_stride = out_stride;
    const size_t lds_bytes = (size_t)make_lds(a.L, a.dout, (a.din + KC - 1) / KC).total * sizeof(float);
    set_tile_split(a, grid_for(rows), 0);  // forward kernels run two workgroups per CU as well: same two-group finish (cm_mlp_kernel.h)
    launch_infer<M_ACT>(a, grid_for(rows), lds_bytes, (hipStream_t)stream);
    CM_CHECK_LAUNCH("cm_policy_act_greedy");
    return 0;
}

/* cm_policy_act / cm_policy_act_eps / cm_policy_act_greedy behind ONE entry point with a caller workspace, so that actors of the
 * layered schedule (hidden 65..256, any depth) can act too: eps == 0 samples Categorical(logits), eps in (0, 1] samples COMA's
 * mixture, eps < 0 takes the argmax.  Same Philox keying (seed, row_offset + row, t) for every shape.  The query is 0 for
 * shapes the fused kernel covers. */
extern "C" size_t cm_policy_act_workspace_bytes(int64_t rows, int din, int hidden, int n_hidden_layers, int n_actions) {
    return wide_shape(hidden, n_hidden_layers, n_actions) ? wide_ws(rows, din, hidden, n_hidden_layers, n_actions, false, true).total * sizeof(float) : 0;
}

extern "C" int cm_policy_act_ws(const float* x, int64_t x_row_stride, const uint8_t* avail, int64_t avail_row_stride,
                                int64_t rows, int din, int hidden, int n_hidden_layers, int n_actions, const float* params,
                                double eps, uint64_t seed, int64_t row_offset, int t, int32_t* action, float* logp,
                                int64_t out_stride, void* ws, size_t ws_bytes, cm_stream_t stream) {
    CM_REQUIRE(eps <= 1.0, "cm_policy_act_ws: eps=%g > 1", eps);
    if (!wide_shape(hidden, n_hidden_layers, n_actions)) {
        if (eps < 0.0) return cm_policy_act_greedy(x, x_row_stride, avail, avail_row_stride, rows, din, hidden, n_hidden_layers, n_actions,
                                                   params, action, logp, out_stride, stream);
        if (eps > 0.0) return cm_policy_act_eps(x, x_row_stride, avail, avail_row_stride, rows, din, hidden, n_hidden_layers, n_actions,
                                                params, eps, seed, row_offset, t, action, logp, out_stride, stream);
        return cm_policy_act(x, x_row_stride, avail, avail_row_stride, rows, din, hidden, n_hidden_layers, n_actions, params, seed,
                             row_offset, t, action, logp, out_stride, stream);
    }
    if (rows <= 0) return 0;
    MlpArgs a = {};
    a.x = x; a.x_stride = x_row_stride; a.rows = rows; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = n_actions;
    a.params = params; a.avail = avail; a.avail_stride = avail_row_stride; a.act_eps = (float)eps;
    a.seed = seed; a.row_offset = row_offset; a.t = t; a.action_out = action; a.logp_out = logp; a.out_stride = out_stride;
    return wide_act(a, ws, ws_bytes, (hipStream_t)stream, "cm_policy_act_ws");
}

/* Act for EVERY step of an episode in one launch when the observations do not depend on the actions (e.g. the shape
 * env): x is [n_seq][T][din] contiguous, avail [n_seq][T][K]; row (s, t) draws Philox(seed, row_offset + s, t), i.e.
 * exactly what T calls of cm_policy_act with t = 0..T-1 draw.  Outputs action / logp [n_seq][T]. */
extern "C" int cm_policy_act_episode_ld(const float* x, int64_t x_ld, const uint8_t* avail, int64_t n_seq, int T, int din, int hidden,
                                        int n_hidden_layers, int n_actions, const float* params, uint64_t seed, int64_t row_offset,
                                        int32_t* action, float* logp, void* ws, size_t ws_bytes, cm_stream_t stream) {
    if (int rc = check_shapes("cm_policy_act_episode", din, hidden, n_hidden_layers, n_actions)) return rc;
    CM_REQUIRE(T > 0 && x_ld >= din, "cm_policy_act_episode: T=%d ld=%lld din=%d", T, (long long)x_ld, din);
    if (n_seq <= 0) return 0;
    if (int rc = check_rows("cm_policy_act_episode", n_seq * T)) return rc;
    MlpArgs a = {};
    a.x = x; a.x_stride = x_ld; a.rows = n_seq * T; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = n_actions;
    a.params = params; a.avail = avail; a.avail_stride = n_actions;
    a.seed = seed; a.row_offset = row_offset; a.t = 0; a.t_decode = T; a.action_out = action; a.logp_out = logp; a.out_stride = 1;
    prep_w0_image(a, (float*)ws, ws_bytes / sizeof(float), (hipStream_t)stream);
    const size_t lds_bytes = (size_t)make_lds(a.L, a.dout, (a.din + KC - 1) / KC).total * sizeof(float);
    CM_SET_PROF(a);
    set_tile_split(a, grid_for(a.rows), 0);  // forward kernels run two workgroups per CU as well: same two-group finish (cm_mlp_kernel.h)
    launch_infer<M_ACT>(a, grid_for(a.rows), lds_bytes, (hipStream_t)stream);
    CM_CHECK_LAUNCH("cm_policy_act_episode");
    return 0;
}
extern "C" int cm_policy_act_episode(const float* x, const uint8_t* avail, int64_t n_seq, int T, int din, int hidden,
                                     int n_hidden_layers, int n_actions, const float* params, uint64_t seed, int64_t row_offset,
                                     int32_t* action, float* logp, cm_stream_t stream) {
    return cm_policy_act_episode_ld(x, din, avail, n_seq, T, din, hidden, n_hidden_layers, n_actions, params, seed, row_offset, action, logp, nullptr, 0, stream);
}
