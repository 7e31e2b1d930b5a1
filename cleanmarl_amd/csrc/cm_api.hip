// cm_api.hip -- error plumbing + small C-ABI helpers of libcleanmarl_hip.so
#include "cm_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void cm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cm_last_error(void) { return g_err; }
extern "C" int cm_version(void) { return 100; }

// 0 = exact fp32 MFMA (default), 1 = CM_MFMA=bf16x3 (error-compensated bf16 GEMM loops in the PPO training passes)
extern "C" int cm_mfma_mode(void) {
    static const int mode = [] { const char* e = getenv("CM_MFMA"); return (e && strcmp(e, "bf16x3") == 0) ? 1 : 0; }();
    return mode;
}

extern "C" int64_t cm_mlp_param_count(int din, int hidden, int n_hidden_layers, int dout) {
    return (int64_t)din * hidden + hidden + (int64_t)n_hidden_layers * ((int64_t)hidden * hidden + hidden) +
           (int64_t)hidden * dout + dout;
}
extern "C" int64_t cm_gru_param_count(int din, int hidden, int dout) {
    return (int64_t)din * hidden + hidden + 6LL * hidden * hidden + 6LL * hidden + (int64_t)hidden * dout + dout;
}

// lowest-priority stream for slack work (see include/cleanmarl_hip.h)
extern "C" cm_stream_t cm_stream_create_low_priority(void) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = 0; }
    hipStream_t s = nullptr;
    const hipError_t e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, least);
    if (e != hipSuccess) { cm_set_error("cm_stream_create_low_priority: %s", hipGetErrorString(e)); return nullptr; }
    return (cm_stream_t)s;
}
extern "C" int cm_stream_destroy(cm_stream_t stream) {
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) CM_FAIL(-2, "cm_stream_destroy: %s", hipGetErrorString(e));
    return 0;
}
