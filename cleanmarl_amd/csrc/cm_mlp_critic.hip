// cm_mlp_critic.hip -- cm_critic_fwd_bwd (a8/a9, critic side)
#include "cm_mlp_train.h"

extern "C" int cm_critic_fwd_bwd(const float* x, const float* ret, const int32_t* ep_len,
                                 int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers,
                                 const float* params, float* grad_and_stats, void* ws, size_t ws_bytes,
                                 cm_stream_t stream) {
    if (int rc = check_shapes("cm_critic_fwd_bwd", din, hidden, n_hidden_layers, 1)) return rc;
    CM_REQUIRE(E > 0 && A > 0 && T > 0, "cm_critic_fwd_bwd: bad dims E=%d A=%d T=%d", E, A, T);
    const size_t need = train_ws_bytes(din, hidden, n_hidden_layers, 1);
    CM_REQUIRE(ws && ws_bytes >= need, "cm_critic_fwd_bwd: workspace too small (%zu < %zu)", ws_bytes, need);
    const int64_t P = cm_mlp_param_count(din, hidden, n_hidden_layers, 1);
    MlpArgs a = {};
    a.x = x; a.x_stride = din; a.rows = per_agent ? (long)E * A * T : (long)E * T;
    a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = 1;
    a.params = params; a.ret = ret; a.ep_len = ep_len; a.A = A; a.T = T; a.per_agent = per_agent ? 1 : 0;
    a.partial = (float*)ws; a.PS = (int)((P + CM_NUM_STATS + 63) / 64 * 64);
#ifdef CM_PHASE_PROF
    a.prof = g_prof;
#endif
    const int grid = grid_for(a.rows, (a.din + KC - 1) / KC);
    const size_t lds_bytes = (size_t)make_lds(a.L, a.dout).total * sizeof(float);
    if (int rc = launch_train<M_CRITIC>(a, grid, lds_bytes, (hipStream_t)stream)) return rc;
    CM_CHECK_LAUNCH("cm_critic_fwd_bwd");
    return finish_train(a, grid, P, grad_and_stats, (hipStream_t)stream, "cm_critic_fwd_bwd");
}
