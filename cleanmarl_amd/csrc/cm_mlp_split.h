// cm_mlp_split.h -- the two schedules of a fused MLP training pass, shared by cm_critic_fwd_bwd, cm_qcritic_fwd_bwd and
// cm_coma_actor_fwd_bwd:
//   * Din <= 128 (<= 2 chunks): everything fused in k_mlp<NCH, MODE> (layer-0 weight-gradient accumulators fit
//     the 256-register budget of two workgroups per CU).
//   * wider inputs (MAPPO's central state, e.g. 384 at config 3; COMA's critic input state|obs|other agents' actions):
//     k_mlp<0, MODE> streams X once for the forward pass, runs the backward pass down to dZ0 and writes dZ0[rows][64]
//     to HBM; the layer-0 weight gradient dW0[64 x Din] = dZ0^T X is then a pure streaming GEMM (k_dw0_stream): no LDS,
//     no barriers, every wave owns a set of 32-column tiles of X, operands go straight from HBM/L2 into MFMA registers
//     with a two-batch software pipeline.  X is read twice from HBM (2 x 0.8 GB at config 3) instead of being held
//     hostage by 96 accumulator registers per lane that limit the fused kernel to one workgroup per CU.
#pragma once
#include "cm_mlp_train.h"

namespace {


constexpr int RB = 8;  // row pairs per software-pipeline batch (16 rows)

// wave w of the workgroup owns k-tiles {w, w+4, ...} (32 columns each); both 32-row halves of the 64 hidden units.
// GEN = false: the tuned layer-0 form (dZ0 row stride HP, X row stride din).  GEN = true (cm_mlp_wide.h): explicit row strides,
// one or two 32-row halves of dZ.
// RBT: row pairs per batch.  The 8-pair form keeps 80 operand registers per lane in flight (234 registers at KTW = 3): a wave of it finds
// no slot on a SIMD that holds two waves of the six-wave rollout (2 x 144 of 512 registers), so beside the next iteration's rollout -- where
// the two-stream schedule puts the critic's epochs -- the WHOLE workgroup waited for the rollout's workgroup to retire (kernel trace of the
// 512-env share: 310 us instead of 70).  The 4-pair form (RBT = 4) fits one wave on every SIMD there; stream_dw() picks it up to 2^16 rows -- the sizes whose rollouts take the six-wave form; anywhere else it is the slower one
// (half the loads in flight: config 5's critic at 2^17 rows 95 -> 151 us).
template <int KTW, bool GEN = false, int RBT = RB>
__global__ __launch_bounds__(NTHREADS, 2) void k_dw0_stream(const float* __restrict__ dz0, const float* __restrict__ x,
                                                            long rows, int din, int H, long rows_per_wg,
                                                            float* __restrict__ partial, int PS2, int col0,
                                                            int ldz_ = HP, long ldx_ = 0) {
    const long ldz = GEN ? ldz_ : HP, ldx = ldx_ ? ldx_ : din;  // ldx_: leading dimension of X when its rows are padded (*_ld entry points)
    const bool two = GEN ? (H > 32) : true;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const long row_lo = (long)blockIdx.x * rows_per_wg;
    const long row_hi = min(rows, row_lo + rows_per_wg);
    f32x16 acc[2][KTW];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < KTW; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[i][j][g] = 0.0f;
    int col[KTW];
    bool cok[KTW];
#pragma unroll
    for (int j = 0; j < KTW; ++j) { col[j] = col0 + 32 * (w + 4 * j) + r; cok[j] = col[j] < din; }

    float a0[RBT], a1[RBT], b[RBT][KTW], na0[RBT], na1[RBT], nb[RBT][KTW];
    auto load = [&](long base, float (&A0)[RBT], float (&A1)[RBT], float (&B)[RBT][KTW]) {
#pragma unroll
        for (int p = 0; p < RBT; ++p) {
            const long row = base + 2 * p + h;
            const bool ok = row < row_hi;
            A0[p] = ok ? dz0[row * ldz + r] : 0.0f;
            A1[p] = (ok && two) ? dz0[row * ldz + 32 + r] : 0.0f;
#pragma unroll
            for (int j = 0; j < KTW; ++j) B[p][j] = (ok && cok[j]) ? x[row * ldx + col[j]] : 0.0f;
        }
    };
    if (row_lo < row_hi) load(row_lo, a0, a1, b);
    for (long base = row_lo; base < row_hi; base += 2 * RBT) {
        if (base + 2 * RBT < row_hi) load(base + 2 * RBT, na0, na1, nb);  // next batch in flight under this batch's MFMAs
#pragma unroll
        for (int p = 0; p < RBT; ++p)
#pragma unroll
            for (int j = 0; j < KTW; ++j) {
                acc[0][j] = mfma32(a0[p], b[p][j], acc[0][j]);
                acc[1][j] = mfma32(a1[p], b[p][j], acc[1][j]);
            }
#pragma unroll
        for (int p = 0; p < RBT; ++p) {
            a0[p] = na0[p]; a1[p] = na1[p];
#pragma unroll
            for (int j = 0; j < KTW; ++j) b[p][j] = nb[p][j];
        }
    }
    float* out = partial + (size_t)blockIdx.x * PS2;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < KTW; ++j)
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int n = 32 * i + (g & 3) + 8 * (g >> 2) + 4 * h;
                if (n < H && cok[j]) out[n * din + col[j]] = acc[i][j][g];
            }
}

constexpr int DW0_GRID = 512;

inline bool use_split(int din) { return (din + KC - 1) / KC > CM_WG2_MAX_NCH; }

// dW[H x din] = dz0[rows][HP]^T X[rows][din]: per-workgroup partials -> out[H * din]
// grid2_out != NULL: leave the partial rows unreduced and report their count (the caller folds them, e.g. with the optimiser step)
template <bool GEN = false>
inline int stream_dw(const float* dz0, const float* x, long rows, int din, int H, float* part2, float* out, hipStream_t s, const char* who,
                     int ldz = HP, long ldx = 0, int* grid2_out = nullptr) {
    const int gopt = cm_option(CM_OPTION_DW0_GRID);  // A/B: workgroups of the streaming kernel (each writes an H x din partial row)
    const int gmax = gopt ? gopt : DW0_GRID;
    long rpw = (rows + gmax - 1) / gmax;
    rpw = (rpw + 2 * RB - 1) / (2 * RB) * (2 * RB);
    const int grid2 = (int)((rows + rpw - 1) / rpw);
    const int PS2 = H * din;
    // rows in flight per lane: see k_dw0_stream (same accumulation order either way)
    const int batch = cm_option(CM_OPTION_DW0_BATCH);
    const bool light = batch ? batch == 4 : rows <= (1L << 16);
    for (int col0 = 0; col0 < din; col0 += 512) {  // one launch per 512-column window of X (4 waves x 4 tiles x 32 columns)
        const int nkt = (min(512, din - col0) + 31) / 32, ktw = (nkt + 3) / 4;
#define CM_DW0_LAUNCH(K)                                                                                                                      \
    if (light) hipLaunchKernelGGL((k_dw0_stream<K, GEN, 4>), dim3(grid2), dim3(NTHREADS), 0, s, dz0, x, rows, din, H, rpw, part2, PS2, col0, ldz, ldx); \
    else hipLaunchKernelGGL((k_dw0_stream<K, GEN>), dim3(grid2), dim3(NTHREADS), 0, s, dz0, x, rows, din, H, rpw, part2, PS2, col0, ldz, ldx)
        switch (ktw) {
            case 1: CM_DW0_LAUNCH(1); break;
            case 2: CM_DW0_LAUNCH(2); break;
            case 3: CM_DW0_LAUNCH(3); break;
            default: CM_DW0_LAUNCH(4); break;
        }
#undef CM_DW0_LAUNCH
    }
    CM_CHECK_LAUNCH(who);
    if (grid2_out) { *grid2_out = grid2; return 0; }
    hipLaunchKernelGGL(k_reduce_partials, dim3((PS2 + RED_COLS - 1) / RED_COLS), dim3(RED_COLS * RED_GROUPS), 0, s, part2, grid2, PS2, 0, PS2, out);
    CM_CHECK_LAUNCH(who);
    return 0;
}
inline size_t stream_dw_ws_floats(int H, int din) { return (size_t)DW0_GRID * H * din; }

inline size_t split_ws_bytes(long rows, int din, int hidden, int L, int dout) {
    size_t b = train_ws_bytes(din, hidden, L, dout);
    if (use_split(din)) b += ((size_t)rows * HP + (size_t)DW0_GRID * hidden * din) * sizeof(float);
    return b;
}

// a: fully populated MlpArgs except partial / PS / dz0 / prof.  Launches the schedule, reduces into grad_and_stats[P + 8].
template <int MODE>
inline int run_train(MlpArgs a, float* grad_and_stats, void* ws, size_t ws_bytes, hipStream_t s, const char* who, const cm_opt_step_t* opt = nullptr) {
    const size_t need = split_ws_bytes(a.rows, a.din, a.H, a.L, a.dout);
    CM_REQUIRE(ws && ws_bytes >= need, "%s: workspace too small (%zu < %zu)", who, ws_bytes, need);
    const int64_t P = cm_mlp_param_count(a.din, a.H, a.L, a.dout);
    a.partial = (float*)ws; a.PS = (int)((P + CM_NUM_STATS + 63) / 64 * 64);
    prep_w0_image(a, train_w0_scratch(ws, P), w0_image_floats(a.din, a.H), s);
#ifdef CM_PHASE_PROF
    a.prof = g_prof;
#endif
    const int nch = (a.din + KC - 1) / KC;
    const size_t lds_bytes = (size_t)make_lds(a.L, a.dout, nch).total * sizeof(float);
    if (!use_split(a.din)) {
        const int grid = grid_for(a.rows, nch);
        if (int rc = launch_train_small<MODE>(a, grid, lds_bytes, s)) return rc;
        CM_CHECK_LAUNCH(who);
        return finish_train(a, grid, P, grad_and_stats, s, who, 0, opt);
    }
    // ---- split schedule: fused kernel without dW0 (two workgroups per CU) ...
    float* own = (float*)ws + (size_t)MAX_GRID * a.PS + w0_image_floats(a.din, a.H);
    if (!a.dz0) a.dz0 = own;  // the caller may want dZ0 for its own use (COMA's factored critic input)
    float* part2 = own + (size_t)a.rows * HP;
    int grid = grid_for(a.rows, 0);
    if (const int gopt = cm_option(CM_OPTION_TRAIN_GRID)) grid = grid < gopt ? grid : gopt;  // A/B: persistent workgroups of the split schedule's fused kernel
    launch_variant<0, MODE>(a, grid, lds_bytes, s);
    CM_CHECK_LAUNCH(who);
    if (opt) {  // both partial sets folded by the optimiser-step launch
        int grid2 = 0;
        if (int rc = stream_dw(a.dz0, a.x, a.rows, a.din, a.H, part2, grad_and_stats, s, who, HP, a.x_stride, &grid2)) return rc;
        return cm_launch_reduce_step(a.partial, grid, a.PS, part2, grid2, a.H * a.din, a.H * a.din, P, grad_and_stats, opt, s, who);
    }
    if (int rc = finish_train(a, grid, P, grad_and_stats, s, who, a.H * a.din)) return rc;  // all but W0
    // ---- ... then the streaming layer-0 weight gradient
    return stream_dw(a.dz0, a.x, a.rows, a.din, a.H, part2, grad_and_stats, s, who, HP, a.x_stride);
}

}  // namespace
