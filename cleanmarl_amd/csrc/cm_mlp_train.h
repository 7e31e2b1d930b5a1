// cm_mlp_train.h -- shared host helpers of the fused training entry points
#pragma once
#include "cm_mlp_kernel.h"

#ifdef CM_PHASE_PROF
extern unsigned long long* g_prof;
#endif

inline size_t train_ws_bytes(int din, int hidden, int n_hidden_layers, int dout) {
    const int64_t P = cm_mlp_param_count(din, hidden, n_hidden_layers, dout);
    const size_t PS = (size_t)((P + CM_NUM_STATS + 63) / 64 * 64);
    return ((size_t)MAX_GRID * PS + w0_image_floats(din, hidden)) * sizeof(float);  // partial rows | padded W0 image (prep_w0_image)
}
inline float* train_w0_scratch(void* ws, int64_t P) { return (float*)ws + (size_t)MAX_GRID * (size_t)((P + CM_NUM_STATS + 63) / 64 * 64); }

// opt != NULL: the optimiser step rides on the reduction launch (cm_optim.hip: k_reduce_step), see cm_opt_step_t in the header
inline int finish_train(const MlpArgs& a, int grid, int64_t P, float* grad_and_stats, hipStream_t s, const char* who, int i0 = 0,
                        const cm_opt_step_t* opt = nullptr) {
    if (opt) return cm_launch_reduce_step(a.partial, grid, a.PS, nullptr, 0, 0, 0, P, grad_and_stats, opt, s, who);
    const int n = (int)(P + CM_NUM_STATS);
    hipLaunchKernelGGL(k_reduce_partials, dim3((n - i0 + RED_COLS - 1) / RED_COLS), dim3(RED_COLS * RED_GROUPS), 0, s, a.partial, grid, a.PS, i0, n, grad_and_stats);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) CM_FAIL(-2, "%s: reduce launch failed: %s", who, hipGetErrorString(e));
    return 0;
}
