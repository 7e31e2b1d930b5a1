// cm_mlp_critic.hip -- cm_critic_fwd_bwd (a8/a9, critic side); schedules in cm_mlp_split.h
#include "cm_mlp_wide.h"
#include "cm_critic_fused.h"

extern "C" size_t cm_critic_workspace_bytes(int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers) {
    const long rows = per_agent ? (long)E * A * T : (long)E * T;
    if (wide_shape(hidden, n_hidden_layers)) return wide_ws_bytes(rows, din, hidden, n_hidden_layers, 1, true);
    return split_ws_bytes(rows, din, hidden, n_hidden_layers, 1);
}

static int critic_pass(const float* x, int64_t x_ld, const float* ret, const int32_t* ep_len,
                       int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers,
                       const float* params, float* grad_and_stats, void* ws, size_t ws_bytes,
                       cm_stream_t stream, const cm_opt_step_t* opt, const float* h0 = nullptr) {
    CM_REQUIRE(x_ld >= din, "cm_critic_fwd_bwd: leading dimension %lld < din %d", (long long)x_ld, din);
    const bool wide = wide_shape(hidden, n_hidden_layers);  // layered schedule (cm_mlp_wide.h)
    if (!wide) if (int rc = check_shapes("cm_critic_fwd_bwd", din, hidden, n_hidden_layers, 1)) return rc;
    CM_REQUIRE(E > 0 && A > 0 && T > 0, "cm_critic_fwd_bwd: bad dims E=%d A=%d T=%d", E, A, T);
    const long rows = per_agent ? (long)E * A * T : (long)E * T;
    if (int rc = check_rows("cm_critic_fwd_bwd", rows)) return rc;
    MlpArgs a = {};
    a.x = x; a.x_stride = x_ld; a.rows = rows;
    a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = 1;
    a.params = params; a.ret = ret; a.ep_len = ep_len; a.A = A; a.T = T; a.per_agent = per_agent ? 1 : 0;
    if (wide) {
        const int rc = wide_train<M_CRITIC>(a, grad_and_stats, ws, ws_bytes, (hipStream_t)stream, "cm_critic_fwd_bwd");
        if (rc || !opt) return rc;
        cm_copy_stats_out(opt, grad_and_stats, cm_mlp_param_count(din, hidden, n_hidden_layers, 1), (hipStream_t)stream);
        return cm_grad_norm_clip_adam(opt->params, grad_and_stats, opt->exp_avg, opt->exp_avg_sq, cm_mlp_param_count(din, hidden, n_hidden_layers, 1),
                                      opt->step, opt->lr, opt->beta1, opt->beta2, opt->eps, opt->weight_decay, opt->opt_kind, opt->max_norm,
                                      opt->grad_scale, opt->out_norm, stream);
    }
    // wide inputs (65 .. 448 columns, one hidden layer): ONE pass over the input with W0 and dW0 in registers (cm_critic_fused.h);
    // -- from CM_FUSED_MIN_ROWS rows on: the kernel wants whole CUs (one 256-thread workgroup with 512 registers per lane and ~140 KB
    // of LDS), so a small batch neither amortises its per-workgroup prologue / partial-gradient row nor shares CUs with the rollout
    // it is overlapped with (learner.overlap_critic).  cm_set_option("critic_schedule", "fused" / "split") forces either schedule (A/B runs, tests).
    const int sched = cm_option(CM_OPTION_CRITIC_SCHEDULE);
    const bool force_fused = sched == 1 || sched == 3, force_split = sched == 2;  // 3 = "fused2": the one-pass kernel with two row tiles per iteration where the shape allows (opt-in)
    if (critic_fused_shape(a) && !force_split && (force_fused || a.rows >= CM_FUSED_MIN_ROWS)) {
        a.dz0 = const_cast<float*>(h0);  // the one-pass kernel without its layer-0 product (k_critic_fused<NC, true>); the other schedules recompute h0
        return run_critic_fused(a, grad_and_stats, ws, ws_bytes, (hipStream_t)stream, "cm_critic_fwd_bwd", opt);
    }
    return run_train<M_CRITIC>(a, grad_and_stats, ws, ws_bytes, (hipStream_t)stream, "cm_critic_fwd_bwd", opt);
}
extern "C" int cm_critic_fwd_bwd_ld(const float* x, int64_t x_ld, const float* ret, const int32_t* ep_len,
                                    int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers,
                                    const float* params, float* grad_and_stats, void* ws, size_t ws_bytes,
                                    cm_stream_t stream) {
    return critic_pass(x, x_ld, ret, ep_len, E, A, T, per_agent, din, hidden, n_hidden_layers, params, grad_and_stats, ws, ws_bytes, stream, nullptr);
}
extern "C" int cm_critic_train_step_ld(const float* x, int64_t x_ld, const float* ret, const int32_t* ep_len,
                                       int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers,
                                       float* grad_and_stats, void* ws, size_t ws_bytes, const cm_opt_step_t* opt, cm_stream_t stream) {
    CM_REQUIRE(opt && opt->params, "cm_critic_train_step: cm_opt_step_t / params is NULL");
    return critic_pass(x, x_ld, ret, ep_len, E, A, T, per_agent, din, hidden, n_hidden_layers, opt->params, grad_and_stats, ws, ws_bytes, stream, opt);
}
/* The first critic epoch of an update with the layer-0 activations the value pass left behind (cm_value_pass_keep_h0_ld on the SAME parameters and rows;
 * h0 = [rows][64] floats): same sums as cm_critic_fwd_bwd_ld / cm_critic_train_step_ld.  h0 is used by the one-pass schedule only (the others recompute it);
 * NULL = the plain entry points. */
extern "C" int cm_critic_fwd_bwd_h0_ld(const float* x, int64_t x_ld, const float* h0, const float* ret, const int32_t* ep_len,
                                       int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers,
                                       const float* params, float* grad_and_stats, void* ws, size_t ws_bytes, cm_stream_t stream) {
    return critic_pass(x, x_ld, ret, ep_len, E, A, T, per_agent, din, hidden, n_hidden_layers, params, grad_and_stats, ws, ws_bytes, stream, nullptr, h0);
}
extern "C" int cm_critic_train_step_h0_ld(const float* x, int64_t x_ld, const float* h0, const float* ret, const int32_t* ep_len,
                                          int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers,
                                          float* grad_and_stats, void* ws, size_t ws_bytes, const cm_opt_step_t* opt, cm_stream_t stream) {
    CM_REQUIRE(opt && opt->params, "cm_critic_train_step: cm_opt_step_t / params is NULL");
    return critic_pass(x, x_ld, ret, ep_len, E, A, T, per_agent, din, hidden, n_hidden_layers, opt->params, grad_and_stats, ws, ws_bytes, stream, opt, h0);
}
extern "C" int cm_critic_fwd_bwd(const float* x, const float* ret, const int32_t* ep_len,
                                 int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers,
                                 const float* params, float* grad_and_stats, void* ws, size_t ws_bytes,
                                 cm_stream_t stream) {
    return cm_critic_fwd_bwd_ld(x, din, ret, ep_len, E, A, T, per_agent, din, hidden, n_hidden_layers, params, grad_and_stats, ws, ws_bytes, stream);
}
