// cm_api.hip -- error plumbing + small C-ABI helpers of libcleanmarl_hip.so
#include "cm_common.h"
#include <string.h>
#include <atomic>

static thread_local char g_err[512] = "";

void cm_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* cm_last_error(void) { return g_err; }
extern "C" int cm_version(void) { return 101; }  // 101: cm_opt_step_t::stats_out, cm_optimizer_step_peer(timeout_s, status), cm_clock_probe

// ---- schedule / arithmetic options (include/cleanmarl_hip.h: cm_set_option).  The library never reads the process environment:
// a caller that wants to force a schedule says so through this entry point (cleanmarl_amd/_native.py maps its CM_* test hooks onto it).
namespace {
struct OptDef { const char* key; const char* const* names; const int* values; int n; };
const char* const kFormsN[] = {"auto", "hand", "loop"};          const int kFormsV[] = {0, 1, 2};
const char* const kCriticN[] = {"auto", "fused", "split", "fused2"}; const int kCriticV[] = {0, 1, 2, 3};
const char* const kGruN[] = {"auto", "64", "32", "8w", "split", "nosplit"}; const int kGruV[] = {0, 64, 32, 8, 1, 2};
const char* const kRollN[] = {"auto", "64", "16", "16s", "64s"}; const int kRollV[] = {0, 64, 16, 17, 65};
const char* const kMfmaN[] = {"fp32", "bf16x3", "bf16"};         const int kMfmaV[] = {0, 1, 2};
const char* const kWideN[] = {"auto", "fused", "layered", "fused_r3"}; const int kWideV[] = {0, 1, 2, 3};
const char* const kDw0N[] = {"auto", "8", "4"};                  const int kDw0V[] = {0, 8, 4};
const char* const kGridN[] = {"auto", "512", "384", "256", "192", "128", "64"}; const int kGridV[] = {0, 512, 384, 256, 192, 128, 64};
const char* const kSplitN[] = {"auto", "50", "52", "53", "54", "55", "56", "57", "58", "60"}; const int kSplitV[] = {0, 50, 52, 53, 54, 55, 56, 57, 58, 60};
const OptDef kOpts[CM_OPTION_COUNT] = {
    {"mlp_forms", kFormsN, kFormsV, 3}, {"critic_schedule", kCriticN, kCriticV, 4}, {"gru_tile", kGruN, kGruV, 6},
    {"rollout_tile", kRollN, kRollV, 5}, {"mfma", kMfmaN, kMfmaV, 3}, {"wide_schedule", kWideN, kWideV, 4},
    {"dw0_batch", kDw0N, kDw0V, 3}, {"dw0_grid", kGridN, kGridV, 7}, {"train_grid", kGridN, kGridV, 7},
    {"tile_split", kSplitN, kSplitV, 10}};
std::atomic<int> g_opt[CM_OPTION_COUNT];  // zero-initialised: every option starts at its first value
}  // namespace

int cm_option(int which) { return g_opt[which].load(std::memory_order_relaxed); }

extern "C" int cm_set_option(const char* key, const char* value) {
    CM_REQUIRE(key && value, "cm_set_option: NULL key / value");
    for (int o = 0; o < CM_OPTION_COUNT; ++o) {
        if (strcmp(key, kOpts[o].key) != 0) continue;
        for (int i = 0; i < kOpts[o].n; ++i)
            if (strcmp(value, kOpts[o].names[i]) == 0) { g_opt[o].store(kOpts[o].values[i], std::memory_order_relaxed); return 0; }
        CM_FAIL(-1, "cm_set_option: option %s has no value \"%s\"", key, value);
    }
    CM_FAIL(-1, "cm_set_option: unknown option \"%s\"", key);
}
extern "C" const char* cm_get_option(const char* key) {
    for (int o = 0; key && o < CM_OPTION_COUNT; ++o) {
        if (strcmp(key, kOpts[o].key) != 0) continue;
        const int v = cm_option(o);
        for (int i = 0; i < kOpts[o].n; ++i) if (kOpts[o].values[i] == v) return kOpts[o].names[i];
    }
    cm_set_error("cm_get_option: unknown option \"%s\"", key ? key : "(null)");
    return nullptr;
}

// 0 = exact fp32 MFMA (default), 1 = option mfma=bf16x3 (error-compensated bf16 GEMM loops in the PPO training passes), 2 = mfma=bf16 (single pass)
extern "C" int cm_mfma_mode(void) { return cm_option(CM_OPTION_MFMA); }

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

// ---- known-answer access to the counter RNG (include/cleanmarl_hip.h): cm_philox4x32 of cm_common.h, the generator behind every
// sampler and synthetic env of the library, evaluated on explicit (counter, key) words -- on the host and on the device -- so that
// tests can hold it to the published Random123 Philox4x32-10 vectors instead of to its own numpy twin.
extern "C" int cm_philox4x32_host(const uint32_t* ctr_key, int64_t n, uint32_t* out) {
    CM_REQUIRE(ctr_key && out && n >= 0, "cm_philox4x32_host: bad arguments");
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t* c = ctr_key + 6 * i;
        const cm_u4 r = cm_philox4x32(c[0], c[1], c[2], c[3], c[4], c[5]);
        out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
    }
    return 0;
}
namespace {
__global__ void k_philox_kat(const uint32_t* __restrict__ ctr_key, int64_t n, uint32_t* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t* c = ctr_key + 6 * i;
    const cm_u4 r = cm_philox4x32(c[0], c[1], c[2], c[3], c[4], c[5]);
    out[4 * i] = r.x; out[4 * i + 1] = r.y; out[4 * i + 2] = r.z; out[4 * i + 3] = r.w;
}
}  // namespace
extern "C" int cm_philox4x32_device(const uint32_t* ctr_key, int64_t n, uint32_t* out, cm_stream_t stream) {
    CM_REQUIRE(ctr_key && out && n >= 0, "cm_philox4x32_device: bad arguments");
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_philox_kat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ctr_key, n, out);
    CM_CHECK_LAUNCH("cm_philox4x32_device");
    return 0;
}
