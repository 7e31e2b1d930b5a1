// cm_mlp_actor.hip -- cm_ppo_actor_fwd_bwd (a8/a9, actor side)
#include "cm_mlp_wide.h"

#ifdef CM_PHASE_PROF
unsigned long long* g_prof = nullptr;
extern "C" void cm_prof_set_buffer(unsigned long long* p) { g_prof = p; }
#endif

// production clock probe (include/cleanmarl_hip.h): shader clock of the actor pass, measured by the pass itself
static unsigned long long* g_clk = nullptr;
extern "C" int cm_clock_probe(uint64_t* ticks) { g_clk = (unsigned long long*)ticks; return 0; }

extern "C" size_t cm_mlp_train_workspace_bytes(int din, int hidden, int n_hidden_layers, int dout) {
    return train_ws_bytes(din, hidden, n_hidden_layers, dout);
}

/* rows-aware query: the layered schedule of wide / deep actors keeps its activations in the workspace */
extern "C" size_t cm_ppo_actor_workspace_bytes(int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions) {
    if (wide_shape(hidden, n_hidden_layers, n_actions)) return wide_ws_bytes((long)E * A * T, din, hidden, n_hidden_layers, n_actions, true);
    return train_ws_bytes(din, hidden, n_hidden_layers, n_actions);
}

// the stand-alone optimiser step behind a schedule that cannot carry it in its reduction launch (layered shapes)
static int step_after(int rc, float* grad_and_stats, int64_t P, const cm_opt_step_t* o, cm_stream_t stream) {
    if (rc || !o) return rc;
    cm_copy_stats_out(o, grad_and_stats, P, (hipStream_t)stream);
    return cm_grad_norm_clip_adam(o->params, grad_and_stats, o->exp_avg, o->exp_avg_sq, P, o->step, o->lr, o->beta1, o->beta2, o->eps,
                                  o->weight_decay, o->opt_kind, o->max_norm, o->grad_scale, o->out_norm, stream);
}

static int actor_pass(const float* obs, int64_t obs_ld, const uint8_t* avail, const int32_t* action,
                      const float* logp_old, const float* adv, const int32_t* ep_len,
                      int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions,
                      const float* params, double ppo_clip, double entropy_coef,
                      float* grad_and_stats, void* ws, size_t ws_bytes, cm_stream_t stream, const cm_opt_step_t* opt) {
    CM_REQUIRE(E > 0 && A > 0 && T > 0 && obs_ld >= din, "cm_ppo_actor_fwd_bwd: bad dims E=%d A=%d T=%d ld=%lld din=%d", E, A, T, (long long)obs_ld, din);
    if (wide_shape(hidden, n_hidden_layers, n_actions)) {  // layered schedule (cm_mlp_wide.h)
        MlpArgs a = {};
        a.x = obs; a.x_stride = obs_ld; a.rows = (long)E * A * T; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = n_actions;
        a.params = params; a.avail = avail; a.avail_stride = n_actions;
        a.action = action; a.logp_old = logp_old; a.adv = adv; a.ep_len = ep_len; a.A = A; a.T = T; a.per_agent = 1;
        a.clip_lo = (float)(1.0 - ppo_clip); a.clip_hi = (float)(1.0 + ppo_clip); a.clip_eps = (float)ppo_clip;
        a.ent_coef = (float)entropy_coef;
        return step_after(wide_train<M_ACTOR>(a, grad_and_stats, ws, ws_bytes, (hipStream_t)stream, "cm_ppo_actor_fwd_bwd"), grad_and_stats,
                          cm_mlp_param_count(din, hidden, n_hidden_layers, n_actions), opt, stream);
    }
    if (int rc = check_shapes("cm_ppo_actor_fwd_bwd", din, hidden, n_hidden_layers, n_actions)) return rc;
    const size_t need = train_ws_bytes(din, hidden, n_hidden_layers, n_actions);
    CM_REQUIRE(ws && ws_bytes >= need, "cm_ppo_actor_fwd_bwd: workspace too small (%zu < %zu)", ws_bytes, need);
    const int64_t P = cm_mlp_param_count(din, hidden, n_hidden_layers, n_actions);
    MlpArgs a = {};
    a.x = obs; a.x_stride = obs_ld; a.rows = (long)E * A * T; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = n_actions;
    a.params = params; a.avail = avail; a.avail_stride = n_actions;
    a.action = action; a.logp_old = logp_old; a.adv = adv; a.ep_len = ep_len; a.A = A; a.T = T; a.per_agent = 1;
    a.clip_lo = (float)(1.0 - ppo_clip); a.clip_hi = (float)(1.0 + ppo_clip); a.clip_eps = (float)ppo_clip;
    a.ent_coef = (float)entropy_coef;
    a.partial = (float*)ws; a.PS = (int)((P + CM_NUM_STATS + 63) / 64 * 64);
    prep_w0_image(a, train_w0_scratch(ws, P), w0_image_floats(a.din, a.H), (hipStream_t)stream);
#ifdef CM_PHASE_PROF
    a.prof = g_prof;
#endif
    a.clk = g_clk;
    const int grid = grid_for(a.rows, (a.din + KC - 1) / KC);
    set_tile_split(a, grid, (a.din + KC - 1) / KC);
    const size_t lds_bytes = (size_t)make_lds(a.L, a.dout, (a.din + KC - 1) / KC).total * sizeof(float);
    if (int rc = launch_train<M_ACTOR>(a, grid, lds_bytes, (hipStream_t)stream)) return rc;
    CM_CHECK_LAUNCH("cm_ppo_actor_fwd_bwd");
    return finish_train(a, grid, P, grad_and_stats, (hipStream_t)stream, "cm_ppo_actor_fwd_bwd", 0, opt);
}

extern "C" int cm_ppo_actor_fwd_bwd_ld(const float* obs, int64_t obs_ld, const uint8_t* avail, const int32_t* action,
                                       const float* logp_old, const float* adv, const int32_t* ep_len,
                                       int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions,
                                       const float* params, double ppo_clip, double entropy_coef,
                                       float* grad_and_stats, void* ws, size_t ws_bytes, cm_stream_t stream) {
    return actor_pass(obs, obs_ld, avail, action, logp_old, adv, ep_len, E, A, T, din, hidden, n_hidden_layers, n_actions, params, ppo_clip,
                      entropy_coef, grad_and_stats, ws, ws_bytes, stream, nullptr);
}

extern "C" int cm_ppo_actor_train_step_ld(const float* obs, int64_t obs_ld, const uint8_t* avail, const int32_t* action,
                                          const float* logp_old, const float* adv, const int32_t* ep_len,
                                          int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions,
                                          double ppo_clip, double entropy_coef, float* grad_and_stats, void* ws, size_t ws_bytes,
                                          const cm_opt_step_t* opt, cm_stream_t stream) {
    CM_REQUIRE(opt && opt->params, "cm_ppo_actor_train_step: cm_opt_step_t / params is NULL");
    return actor_pass(obs, obs_ld, avail, action, logp_old, adv, ep_len, E, A, T, din, hidden, n_hidden_layers, n_actions, opt->params, ppo_clip,
                      entropy_coef, grad_and_stats, ws, ws_bytes, stream, opt);
}

extern "C" int cm_ppo_actor_fwd_bwd(const float* obs, const uint8_t* avail, const int32_t* action,
                                    const float* logp_old, const float* adv, const int32_t* ep_len,
                                    int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions,
                                    const float* params, double ppo_clip, double entropy_coef,
                                    float* grad_and_stats, void* ws, size_t ws_bytes, cm_stream_t stream) {
    return cm_ppo_actor_fwd_bwd_ld(obs, din, avail, action, logp_old, adv, ep_len, E, A, T, din, hidden, n_hidden_layers, n_actions, params, ppo_clip,
                                   entropy_coef, grad_and_stats, ws, ws_bytes, stream);
}

/* MFMA work the fused actor pass ISSUES per row (flop), padding included -- next to the algorithmic 2 P_a + 2 P_a + 2 (P_a - Do H) of
 * SURVEY.md 8(d) that bench.py's roofline is quoted on.  Counted from the tile schedule of k_mlp<NCH, M_ACTOR> (cm_mlp_kernel.h), per
 * 64-row tile and wave, 4 waves per tile: forward layer 0 ceil(w / 8) k-steps x 4 v_mfma_f32_32x32x2 per 64-column chunk of width w,
 * 32 per hidden layer; logits and dWout on v_mfma_f32_16x16x4, 16 each per 16 padded head rows (K <= 8 -> 16 rows, else 32); dZ_L
 * 4 x 32x32x2 per 8 padded head columns (K <= 8 -> 8, else 32); per hidden layer 32 (weight gradient) + 32 (data path); layer-0 weight
 * gradient 32 per 64-column chunk (the chunk's padding columns included).  32x32x2 = 4096 flop, 16x16x4 = 2048 flop.  0 for shapes that
 * run on the layered schedule. */
extern "C" double cm_ppo_actor_issued_flop_per_row(int din, int hidden, int n_hidden_layers, int n_actions) {
    if (wide_shape(hidden, n_hidden_layers, n_actions) || hidden > HP || n_hidden_layers > LMAX || n_actions > KMAX || din <= 0) return 0.0;
    const int nch = (din + KC - 1) / KC;
    const int KP = n_actions <= 8 ? 8 : KMAX, WR = n_actions <= 8 ? 16 : KMAX;
    long m32 = 0, m16 = 0;
    for (int c = 0; c < nch; ++c) {
        const int w = (din - c * KC) < KC ? (din - c * KC) : KC;
        m32 += 4 * ((w + 7) >> 3);   // forward layer 0, chunk c
        m32 += 32;                   // dW0, chunk c
    }
    m32 += (long)n_hidden_layers * (32 + 32 + 32);  // forward, weight gradient, data path of every hidden layer
    long m4 = 0;
    if (CM_HEAD_44 != 0 && nch == 1 && n_hidden_layers <= 1 && n_actions <= 8) {
        m4 += 32 + 32 + 32;          // the hand-ordered instantiation (launches of >= 2^21 rows): logits, dWout, dZ_L on the 4x4x1 MFMA, 8 head columns
    } else {
        m16 += 2 * 16 * (WR / 16);   // logits + dWout on the 16x16x4 MFMA, 16 head columns per block
        m32 += 4 * (KP / 8);         // dZ_L
    }
    return (double)(4 * (m32 * 4096 + m16 * 2048 + m4 * 512)) / (double)TM;
}
