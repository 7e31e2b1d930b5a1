// cm_scan.hip -- TD(lambda) reverse scan, masked moments, normalisation (HBM-bound kernels).
//
// Replaces the Python double loop of cleanmarl/mappo_multienvs.py:484-504 (2*E*T single-row critic
// calls) by ONE batched value pass (cm_mlp_forward) + this scan.  The recurrence
//     R_t = b_t + a * R_{t+1},   a = gamma*lambda,   b_t = r_t + gamma*(1-lambda)*V_{t+1}
// is linear with a constant coefficient, so each wavefront solves one sequence with a three-phase
// scan: (1) every lane scans its own chunk of c = ceil(T/64) consecutive steps out of LDS,
// (2) a 6-step Kogge-Stone suffix scan over the 64 lane carries with __shfl_down and coefficient
// a^c, a^2c, ..., (3) lanes fold the incoming carry into their chunk and write R and A = R - V.
#include "cm_common.h"

#define SCAN_WAVES 4

__global__ __launch_bounds__(SCAN_WAVES * 64) void k_td_lambda_scan(
    const float* __restrict__ reward, const float* __restrict__ values, const int* __restrict__ ep_len,
    int E, int A, int Av, int T, float a, float gv, float* __restrict__ ret, float* __restrict__ adv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int seq = blockIdx.x * SCAN_WAVES + wave;  // (e, av)
    const bool active = seq < E * Av;                // wave-uniform
    const int e = active ? seq / Av : 0, av = active ? seq % Av : 0;
    const int L = active ? min(max(ep_len[e], 0), T) : 0;
    float* sb = smem + wave * 2 * T;  // b_t, later R_t
    float* sv = sb + T;               // V_t (masked)
    const float* rp = reward + (size_t)e * T;
    const float* vp = values + (size_t)(active ? seq : 0) * T;
    // stage: coalesced loads, masked
    for (int t = lane; t < T; t += 64) sv[t] = (t < L) ? vp[t] : 0.0f;
    __syncthreads();
    for (int t = lane; t < T; t += 64) {
        const float vn = (t + 1 < L) ? sv[t + 1] : 0.0f;
        sb[t] = (t < L) ? (rp[t] + gv * vn) : 0.0f;
    }
    __syncthreads();
    // phase 1: local reverse scan of this lane's chunk [t0, t1)
    const int c = (T + 63) >> 6;
    const int t0 = lane * c, t1 = min(t0 + c, T);
    float run = 0.0f;
    for (int t = t1 - 1; t >= t0; --t) { run = fmaf(a, run, sb[t]); sb[t] = run; }
    // phase 2: suffix scan over lane carries, S_i = run_i + a^c * S_{i+1}
    float ac = 1.0f;
    for (int i = 0; i < c; ++i) ac *= a;
    float S = run, coef = ac;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_down(S, d, 64);
        if (lane + d < 64) S = fmaf(coef, o, S);
        coef *= coef;
    }
    float carry = __shfl_down(S, 1, 64);
    if (lane == 63) carry = 0.0f;
    // phase 3: fold the carry: R_t = R^loc_t + a^(t1 - t) * carry
    float p = a;
    for (int t = t1 - 1; t >= t0; --t) { sb[t] = fmaf(p, carry, sb[t]); p *= a; }
    __syncthreads();
    // write back (coalesced); Av == 1 broadcasts the sequence to all A agents
    if (!active) return;
    const int na = (Av == 1) ? A : 1;
    for (int k = 0; k < na; ++k) {
        const size_t base = ((size_t)e * A + (Av == 1 ? k : av)) * T;
        for (int t = lane; t < T; t += 64) {
            const bool valid = t < L;
            const float r = valid ? sb[t] : 0.0f;
            ret[base + t] = r;
            adv[base + t] = valid ? (r - sv[t]) : 0.0f;
        }
    }
}

extern "C" int cm_td_lambda_scan(const float* reward, const float* values, const int32_t* ep_len,
                                 int E, int A, int Av, int T, double gamma, double lam,
                                 float* ret, float* adv, cm_stream_t stream) {
    CM_REQUIRE(E > 0 && A > 0 && T > 0 && (Av == 1 || Av == A), "cm_td_lambda_scan: bad dims E=%d A=%d Av=%d T=%d", E, A, Av, T);
    CM_REQUIRE((size_t)SCAN_WAVES * 2 * T * sizeof(float) <= 160 * 1024, "cm_td_lambda_scan: T=%d too long for LDS staging", T);
    const float a = (float)(gamma * lam);
    const float gv = (float)(gamma * (1.0 - lam));  // (1 - lambda) evaluated in float64 as the reference does
    const int nseq = E * Av;
    const int grid = (nseq + SCAN_WAVES - 1) / SCAN_WAVES;
    const size_t lds = (size_t)SCAN_WAVES * 2 * T * sizeof(float);
    if (lds > 64 * 1024)  // dynamic LDS above the 64 KB default needs the attribute (T > 2048 at four waves per workgroup)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_td_lambda_scan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_td_lambda_scan, dim3(grid), dim3(SCAN_WAVES * 64), lds, (hipStream_t)stream,
                       reward, values, ep_len, E, A, Av, T, a, gv, ret, adv);
    CM_CHECK_LAUNCH("cm_td_lambda_scan");
    return 0;
}

// ---------------------------------------------------------------- masked moments of the agent-mean
// y[e,t] = mean_a x[e,a,t] over valid (e,t)  ->  (count, mean, M2) in float64, mergeable across shards.
#define MOM_BLOCK 256
#define MOM_MAX_GRID 1024

__global__ __launch_bounds__(MOM_BLOCK) void k_moments_partial(const float* __restrict__ x, const int* __restrict__ ep_len,
                                                               int E, int A, int T, double* __restrict__ part) {
    double s = 0.0, s2 = 0.0, n = 0.0;
    const long total = (long)E * T;
    const float invA = 1.0f / (float)A;
    for (long i = (long)blockIdx.x * MOM_BLOCK + threadIdx.x; i < total; i += (long)gridDim.x * MOM_BLOCK) {
        const int e = (int)(i / T), t = (int)(i % T);
        if (t < ep_len[e]) {
            float acc = 0.0f;
            for (int a = 0; a < A; ++a) acc += x[((size_t)e * A + a) * T + t];
            const float y = (A == 1) ? acc : acc * invA;
            s += (double)y; s2 += (double)y * (double)y; n += 1.0;
        }
    }
    __shared__ double sh[3][MOM_BLOCK / 64];
    s = cm_wave_sum_d(s); s2 = cm_wave_sum_d(s2); n = cm_wave_sum_d(n);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = s2; sh[2][threadIdx.x >> 6] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a0 = 0, a1 = 0, a2 = 0;
        for (int w = 0; w < MOM_BLOCK / 64; ++w) { a0 += sh[0][w]; a1 += sh[1][w]; a2 += sh[2][w]; }
        part[3 * blockIdx.x + 0] = a0; part[3 * blockIdx.x + 1] = a1; part[3 * blockIdx.x + 2] = a2;
    }
}

__global__ __launch_bounds__(64) void k_moments_final(const double* __restrict__ part, int nparts, double* __restrict__ out) {
    double s = 0.0, s2 = 0.0, n = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) { s += part[3 * i]; s2 += part[3 * i + 1]; n += part[3 * i + 2]; }
    s = cm_wave_sum_d(s); s2 = cm_wave_sum_d(s2); n = cm_wave_sum_d(n);
    if (threadIdx.x == 0) {
        const double mean = n > 0 ? s / n : 0.0;
        out[0] = n; out[1] = mean; out[2] = s2 - n * mean * mean;  // M2 = sum (y - mean)^2
    }
}

static int moments_grid(int E, int T) {
    long total = (long)E * T;
    long g = (total + MOM_BLOCK - 1) / MOM_BLOCK;
    return (int)(g < 1 ? 1 : (g > MOM_MAX_GRID ? MOM_MAX_GRID : g));
}

extern "C" size_t cm_masked_moments_workspace_bytes(int E, int A, int T) {
    (void)A;
    return (size_t)moments_grid(E, T) * 3 * sizeof(double);
}

extern "C" int cm_masked_moments(const float* x, const int32_t* ep_len, int E, int A, int T,
                                 double* out, void* ws, size_t ws_bytes, cm_stream_t stream) {
    CM_REQUIRE(E > 0 && A > 0 && T > 0, "cm_masked_moments: bad dims");
    const int grid = moments_grid(E, T);
    CM_REQUIRE(ws && ws_bytes >= (size_t)grid * 3 * sizeof(double), "cm_masked_moments: workspace too small (%zu < %zu)",
               ws_bytes, (size_t)grid * 3 * sizeof(double));
    hipLaunchKernelGGL(k_moments_partial, dim3(grid), dim3(MOM_BLOCK), 0, (hipStream_t)stream, x, ep_len, E, A, T, (double*)ws);
    CM_CHECK_LAUNCH("cm_masked_moments/partial");
    hipLaunchKernelGGL(k_moments_final, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)ws, grid, out);
    CM_CHECK_LAUNCH("cm_masked_moments/final");
    return 0;
}

__global__ __launch_bounds__(256) void k_normalize(float* __restrict__ x, const int* __restrict__ ep_len, int E, int A, int T,
                                                   const double* __restrict__ mom, float eps, int valid_only) {
    const double n = mom[0];
    const float mean = (float)mom[1];
    const float sd = (float)sqrt(mom[2] / (n - 1.0));  // unbiased, torch.std default
    const float denom = sd + eps;
    const long total = (long)E * A * T;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        if (valid_only) {
            const int t = (int)(i % T);
            const int e = (int)(i / ((long)A * T));
            if (t >= ep_len[e]) continue;
        }
        x[i] = (x[i] - mean) / denom;
    }
}

extern "C" int cm_normalize(float* x, const int32_t* ep_len, int E, int A, int T, const double* mom, float eps,
                            int valid_only, cm_stream_t stream) {
    CM_REQUIRE(E > 0 && A > 0 && T > 0, "cm_normalize: bad dims");
    long total = (long)E * A * T;
    long g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_normalize, dim3((int)g), dim3(256), 0, (hipStream_t)stream, x, ep_len, E, A, T, mom, eps, valid_only);
    CM_CHECK_LAUNCH("cm_normalize");
    return 0;
}
