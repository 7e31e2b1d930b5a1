// cm_mlp_kernel.h -- fused actor / critic MLP kernel template for gfx950 (fp32 MFMA, weights LDS-stationary).
//
// One kernel template covers the four program regions of cleanmarl/mappo_multienvs.py that run an MLP:
//   M_FWD    Actor.logits / Critic.forward         (:178-183, :197-200)   -> cm_mlp_forward
//   M_ACT    Actor.act (sample + log_prob)         (:172-176, :409-414)   -> cm_policy_act
//   M_ACTOR  PPO clipped-surrogate fwd + bwd       (:527-551, :561-582)   -> cm_ppo_actor_fwd_bwd
//   M_CRITIC value MSE fwd + bwd                   (:554-558, :582)       -> cm_critic_fwd_bwd
//
// Design (docs/KERNEL_NOTES.md §3): a workgroup = 4 wavefronts (2 along rows x 2 along hidden columns) owns a tile of
// TM = 64 rows and walks the whole network for that tile with every activation resident in LDS; rows
// never round-trip to HBM between layers or between forward and backward.  All GEMMs run on
// v_mfma_f32_32x32x2_f32 (exact fp32, == an fmaf chain), so results stay within fp32 round-off of the
// reference's CPU PyTorch run.  Three MFMA forms are used, each wave owning one 32x32 output tile:
//   rowpar_nt : Y[64 x 64]  = A[64 x K] * W[64 x K]^T        (forward layers)
//   rowpar_tn : dX[64 x 64] = dZ[64 x 64] * W[64 x 64]       (backward data path)
//   colred    : dW[64 x 64] += dZ[64 rows x 64]^T * X[64 rows x 64]  (weight gradients; the accumulators
//               stay in registers across ALL row tiles of the persistent workgroup and are written once)
// The head (K <= 32 outputs) runs on v_mfma_f32_16x16x4_f32 (logits) and 32x32x2 (backward); the softmax / PPO /
// MSE math runs on the VALU with the K logits of a row spread over 4 lanes (DPP quad reductions).
// Hidden widths H <= 64 are zero-padded to 64 in LDS (dead units have zero activations and zero
// gradients), the input width is processed in chunks of 64 columns.
#pragma once
#include "cm_common.h"

namespace {

constexpr int HP = 64;    // padded hidden width
constexpr int TM = 64;    // rows per tile
constexpr int KC = 64;    // input chunk width
constexpr int LDT = 68;   // LDS row stride in floats (4*17: conflict-free ds_read_b128 down a column of rows)
constexpr int LMAX = 2;   // max hidden->hidden layers (kernels are compiled for LCAP = 1 or 2)
constexpr int KMAX = 32;  // max head width
constexpr int LSP = 32;   // row stride of the per-row head scratch in the GRU kernels (= KMAX, zero padded)
constexpr int KJMAX = KMAX / 4;  // head outputs owned per lane (4 lanes per row); kernels compiled for KJ = 2 or 8
constexpr int NTHREADS = 256;
// Workgroups per CU: two independent workgroups overlap each other's VALU and MFMA phases (needs <= 80 KB of LDS
// and <= 256 registers); kernels with many layer-0 gradient chunks keep one workgroup per CU and 512 registers.
#ifndef CM_WG2_MAX_NCH
#define CM_WG2_MAX_NCH 2
#endif
constexpr int wgs_per_cu(int nch) { return nch <= CM_WG2_MAX_NCH ? 2 : 1; }
#ifndef CM_TILE_SPLIT_DEFAULT
#define CM_TILE_SPLIT_DEFAULT 56  // per cent of the tiles for the first half of a full grid (set_tile_split); 50 = equal
#endif
#ifndef CM_HEAD_FAST
#define CM_HEAD_FAST 0  // 1 = the PPO head's exp / log / reciprocal as single hardware instructions (v_exp_f32, v_log_f32, v_rcp_f32: 1 ulp each) instead of libm's
#endif
#ifndef CM_HEAD_44
#define CM_HEAD_44 1  // wave-private training heads (K <= 8) on v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 blocks per instruction: 8 head outputs cost 8, not 16,
                      // columns of MFMA work); 0 = the 16x16x4 products of round 4, kept for A/B builds
#endif
#ifndef CM_HEAD_WP
#define CM_HEAD_WP 1  // wave-private head backward (head_bwd_wave); 0 = the workgroup-wide products of rounds 1 - 3, kept for A/B builds
#endif

// COMA (cleanmarl/coma_multienvs.py): M_QCRITIC = MSE on the Q of the TAKEN action of a K-output critic (:620-631),
// M_COMA_ACTOR = counterfactual policy gradient -log(pi_a + 1e-8) * adv - c * mean_k entropy (:649-676)
enum Mode { M_FWD = 0, M_ACT = 1, M_ACTOR = 2, M_CRITIC = 3, M_QCRITIC = 4, M_COMA_ACTOR = 5 };

struct MlpArgs {
    const float* x; long x_stride; long rows;
    int din, H, L, dout;
    const float* params;
    // M_FWD
    const uint8_t* avail; long avail_stride; float* y;
    // M_ACT
    unsigned long long seed; long row_offset; int t; int* action_out; float* logp_out; long out_stride;
    const float* z0_add;  // M_FWD / M_QCRITIC: optional [rows][HP] addend of the layer-0 pre-activation (COMA's factored critic input)
    float act_eps;  // > 0: COMA's epsilon-mixed sampling (cm_policy_act_eps); < 0: greedy (cm_policy_act_greedy)
    int t_decode;  // > 0: rows are (sequence, t) pairs with t = row % t_decode (whole-episode act pass), else a.t
    // training
    const int* action; const float* logp_old; const float* adv; const float* ret; const int* ep_len;
    int A, T, per_agent;
    float clip_lo, clip_hi, clip_eps, ent_coef;
    float* partial; int PS;  // per-workgroup partial gradients + stats, row stride PS floats
    float* dz0;  // training kernels instantiated with NCH == 0: layer-0 pre-activation gradient [rows][HP] goes to HBM
                 // and the layer-0 weight gradient is computed by the streaming kernel k_dw0_stream instead
    unsigned long long* prof;  // CM_PHASE_PROF builds only: [grid][16] cycle counters
    long split_tiles;  // > 0: UNEQUAL static split of the tiles over the two workgroups of a CU (set_tile_split): workgroups [0, grid / 2)
                       // stride over tiles [0, split_tiles), workgroups [grid / 2, grid) over [split_tiles, ntiles)
    unsigned long long* clk;   // production clock probe (cm_clock_probe; NULL = off): workgroup w of an M_ACTOR launch writes clk[4 w ..] =
                               // {shader cycles (s_memtime) of its tile loop, XCC_ID << 32 | HW_ID, s_memrealtime at entry, at exit (100 MHz reference ticks)}
    // zero-padded image of W0 with a leading dimension that is a multiple of 4 floats ([H][w0_ld], built per call by prep_w0_image):
    // set when W0 is STREAMED (din > 64) and its rows are not 16-byte aligned (din % 4 != 0) while the input's rows are (*_ld entry
    // points) -- the streamed chunks then come from this image on 16-byte loads instead of 16 4-byte loads per thread and tile
    const float* w0p; int w0_ld;
};

struct Offsets {
    int W0, b0, Wl0, lstep, Wout, bout, P;
    __host__ __device__ int Wl(int l) const { return Wl0 + l * lstep; }            // hidden layer l (0-based) weight
    __host__ __device__ int bl(int l) const { return Wl0 + l * lstep + lstep - hdim; }  // and bias
    int hdim;
};
__host__ __device__ inline Offsets make_offsets(int din, int H, int L, int dout) {
    Offsets o;
    o.hdim = H;
    o.W0 = 0; o.b0 = H * din;
    o.Wl0 = o.b0 + H; o.lstep = H * H + H;
    o.Wout = o.Wl0 + L * o.lstep; o.bout = o.Wout + dout * H; o.P = o.bout + dout;
    return o;
}

// LDS carve (floats)
struct Lds {
    int Xs, W0s, Hs0, Ws, wout, b0, bl0, bout, ls, red, total;
    // hidden layer 1's activations live in the X buffer: X is dead once layer 0 has consumed it and is
    // re-streamed (L2 hit) for the layer-0 weight gradient -- this is what gets the footprint under 80 KB
    __host__ __device__ int Hs(int l) const { return l == 0 ? Hs0 : (l == 1 ? Xs : Hs0 + (l - 1) * TM * LDT); }
    __host__ __device__ int bl(int l) const { return bl0 + l * HP; }
};
constexpr int WLD = 68;  // row stride of the head weight image (conflict-free ds_read_b128 down a column of rows)
__host__ __device__ inline int head_kp(int dout) { return dout <= 8 ? 8 : KMAX; }      // ls row stride (= KJ * 4)
__host__ __device__ inline int head_wr(int dout) { return dout <= 8 ? 16 : KMAX; }     // zero-padded rows of the head weights
__host__ __device__ inline Lds make_lds(int L, int dout, int nch) {
    Lds s; int p = 0;
    s.Xs = p; p += TM * LDT;
    s.W0s = p; p += HP * LDT;
    s.Hs0 = p; p += (L >= 1 ? L : 1) * TM * LDT;
    s.Ws = p; if (L > 0) p += HP * LDT;
    s.wout = p; p += head_wr(dout) * WLD;  // rows >= dout are zero (MFMA operand padding)
    s.b0 = p; p += HP;
    s.bl0 = p; p += L * HP;
    s.bout = p; p += KMAX;
    // the per-row head scratch is only live between the forward of layer 0 and the backward of the hidden layers;
    // when W0 is streamed per tile (nch > 1) its chunk buffer is free in exactly that window -> alias (keeps K > 8
    // heads under the 80 KB that two workgroups per CU allow)
    if (nch > 1) s.ls = s.W0s; else { s.ls = p; p += TM * head_kp(dout); }
    p = (p + 3) & ~3;
    s.red = s.Xs;  // 256 floats of scratch for the final partial write: the X buffer is dead by then
    s.total = p;
    return s;
}

// exp / log / reciprocal of the PPO head's per-row math (CM_HEAD_FAST)
__device__ __forceinline__ float head_exp(float x) { return CM_HEAD_FAST ? __expf(x) : expf(x); }
__device__ __forceinline__ float head_log(float x) { return CM_HEAD_FAST ? __logf(x) : logf(x); }
__device__ __forceinline__ float head_rcp(float x) { return CM_HEAD_FAST ? __builtin_amdgcn_rcpf(x) : 1.0f / x; }

// 4-lane (quad) butterflies on the VALU via DPP quad_perm -- no LDS round trip like ds_bpermute
__device__ __forceinline__ float quad_xor1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_xor2(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_sum(float v) { v += quad_xor1(v); v += quad_xor2(v); return v; }
__device__ __forceinline__ float quad_max(float v) { v = fmaxf(v, quad_xor1(v)); v = fmaxf(v, quad_xor2(v)); return v; }
__device__ __forceinline__ float quad_shr1(float v) {  // lane h of a quad <- lane h - 1 (lane 0 keeps its own)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x90, 0xF, 0xF, true));
}
__device__ __forceinline__ float quad_bcast3(float v) {  // every lane of a quad <- lane 3
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xFF, 0xF, 0xF, true));
}
__device__ __forceinline__ int quad_imin(int v) {
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true)); return min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ int quad_imax(int v) {
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true)); return max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));
}

// cm_categorical_sample for K > 8 on the FOUR lanes of a row (lane hq holds the masked logits of actions k = 4 j + hq in z[j]): the same
// operations in the same order -- s = e_0 + e_1 + ... left to right (one exp per action), the first available action whose running sum
// exceeds u * s, log_prob = z[a] - (max + log s) -- with the running sum handed from lane to lane inside the quad (DPP), so that a
// lane evaluates K / 4 exponentials instead of one lane 2 K (cm_common.h sums twice for K > 8).  An unavailable action adds exactly 0,
// so the running sum over all actions equals the serial sum over the available ones at every available k.  Every lane returns the row's
// result.  (Config 4's act pass spent 41 % of its time in the one-lane sampler: profiles/r03_phase_act.txt.)
template <int KJ>
__device__ __forceinline__ void quad_categorical_sample(const float (&z)[KJ], int K, int hq, float u, int* action, float* logp) {
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < KJ; ++j) if (4 * j + hq < K) m = fmaxf(m, z[j]);
    m = quad_max(m);
    float pre[KJ];
    float run = 0.0f;
#pragma unroll
    for (int j = 0; j < KJ; ++j) {
        pre[j] = run;
        if (4 * j < K) {  // wave-uniform: blocks beyond the last action are skipped
            const float e = (4 * j + hq < K) ? expf(z[j] - m) : 0.0f;
            float p = run + e;                         // lane 0 of the quad
            float q = quad_shr1(p); if (hq == 1) p = q + e;
            q = quad_shr1(p); if (hq == 2) p = q + e;
            q = quad_shr1(p); if (hq == 3) p = q + e;
            pre[j] = p;                                // inclusive running sum at action 4 j + hq
            run = quad_bcast3(p);
        }
    }
    const float s = run, thr = u * s;
    int first = 1 << 20, last = 0;
#pragma unroll
    for (int j = KJ - 1; j >= 0; --j) {
        const int k = 4 * j + hq;
        if (k < K && z[j] > -5e8f) {
            if (thr < pre[j]) first = k;
            if (last == 0) last = k;                   // descending j: the largest available k of this lane comes first
        }
    }
    first = quad_imin(first); last = quad_imax(last);
    const int chosen = first < (1 << 20) ? first : last;
    float zc = -INFINITY;
#pragma unroll
    for (int j = 0; j < KJ; ++j) if (4 * j + hq == chosen) zc = z[j];
    zc = quad_max(zc);
    *action = chosen;
    *logp = zc - (m + logf(s));
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// All three MFMA loops are software-pipelined by hand: the LDS operands of batch i+1 are requested before the
// MFMAs of batch i issue, so the ~100-cycle ds_read latency hides under the 4 x 64-cycle MFMA batch instead of
// stalling the matrix pipe once per loop iteration.  Written in C++ that order did not survive the compiler (round 2, ISA of the actor
// kernel): the forward loop waited for one of the two FRESH reads of every batch (`s_waitcnt lgkmcnt(1)` right behind them: the
// `a = an` rotation had become a copy of the in-flight register's successor) and the weight-gradient loop issued read, lgkmcnt(0),
// 2 MFMAs.  The reads are therefore issued through cm_common.h's asm helpers (program order kept, waits placed by hand, destination
// registers never touched in flight -- checked on the generated code by tools/lint_lds_hazards.py); every pipeline is straight-line code.
//
// acc[32x32] += A[32 rows][8*kb] * B[32 rows(n)][8*kb]^T ; A,B row-major in LDS with stride LDT.
// k is consumed in the permuted order {8j+i, 8j+4+i}: lane half h reads floats [8j+4h, 8j+4h+4) as one b128.
template <int J, int KB>
__device__ __forceinline__ void nt_steps(f32x16& acc, unsigned aa, unsigned ba, f32x4& a0, f32x4& b0, f32x4& a1, f32x4& b1) {
    if constexpr (J < KB) {
        f32x4& ca = (J & 1) ? a1 : a0; f32x4& cb = (J & 1) ? b1 : b0;
        if constexpr (J + 1 < KB) {
            f32x4& na = (J & 1) ? a0 : a1; f32x4& nb = (J & 1) ? b0 : b1;
            na = cf_lds128<32 * (J + 1)>(aa); nb = cf_lds128<32 * (J + 1)>(ba);
            cf_wait<2>(ca, cb);
        } else {
            cf_wait<0>(ca, cb);
        }
        acc = mfma32(ca[0], cb[0], acc);
        acc = mfma32(ca[1], cb[1], acc);
        acc = mfma32(ca[2], cb[2], acc);
        acc = mfma32(ca[3], cb[3], acc);
        nt_steps<J + 1, KB>(acc, aa, ba, a0, b0, a1, b1);
    }
}
template <int KB>
__device__ __forceinline__ void rowpar_nt_pipe(f32x16& acc, unsigned aa, unsigned ba) {
    f32x4 a0 = cf_lds128<0>(aa), b0 = cf_lds128<0>(ba), a1, b1;
    nt_steps<0, KB>(acc, aa, ba, a0, b0, a1, b1);
}
__device__ __forceinline__ void rowpar_nt_hand(f32x16& acc, const float* As, const float* Bs, int kb) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const unsigned aa = cf_lds_addr(As + r * LDT + 4 * h), ba = cf_lds_addr(Bs + r * LDT + 4 * h);
    switch (kb) {  // wave-uniform; each case is one straight-line pipeline with its first read inside
        case 8: rowpar_nt_pipe<8>(acc, aa, ba); break;
        case 7: rowpar_nt_pipe<7>(acc, aa, ba); break;
        case 6: rowpar_nt_pipe<6>(acc, aa, ba); break;
        case 5: rowpar_nt_pipe<5>(acc, aa, ba); break;
        case 4: rowpar_nt_pipe<4>(acc, aa, ba); break;
        case 3: rowpar_nt_pipe<3>(acc, aa, ba); break;
        case 2: rowpar_nt_pipe<2>(acc, aa, ba); break;
        case 1: rowpar_nt_pipe<1>(acc, aa, ba); break;
        default: break;
    }
}

// acc[32x32] += dZ[32 rows][64 (n)] * W[64 (n)][32 cols]  (W row-major [n][k] in LDS, read transposed)
template <int J>
__device__ __forceinline__ void tn_steps(f32x16& acc, unsigned aa, unsigned ba, f32x4& a0, float (&b0)[4], f32x4& a1, float (&b1)[4]) {
    if constexpr (J < HP / 8) {
        f32x4& ca = (J & 1) ? a1 : a0; float (&cb)[4] = (J & 1) ? b1 : b0;
        if constexpr (J + 1 < HP / 8) {
            f32x4& na = (J & 1) ? a0 : a1; float (&nb)[4] = (J & 1) ? b0 : b1;
            na = cf_lds128<32 * (J + 1)>(aa);
            nb[0] = cf_lds32<(8 * (J + 1) + 0) * LDT * 4>(ba); nb[1] = cf_lds32<(8 * (J + 1) + 1) * LDT * 4>(ba);
            nb[2] = cf_lds32<(8 * (J + 1) + 2) * LDT * 4>(ba); nb[3] = cf_lds32<(8 * (J + 1) + 3) * LDT * 4>(ba);
            cf_wait<5>(ca, cb);
        } else {
            cf_wait<0>(ca, cb);
        }
        acc = mfma32(ca[0], cb[0], acc);
        acc = mfma32(ca[1], cb[1], acc);
        acc = mfma32(ca[2], cb[2], acc);
        acc = mfma32(ca[3], cb[3], acc);
        tn_steps<J + 1>(acc, aa, ba, a0, b0, a1, b1);
    }
}
__device__ __forceinline__ void rowpar_tn_hand(f32x16& acc, const float* As, const float* Ws_c0) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const unsigned aa = cf_lds_addr(As + r * LDT + 4 * h), ba = cf_lds_addr(Ws_c0 + (4 * h) * LDT + r);
    f32x4 a0 = cf_lds128<0>(aa), a1;
    float b0[4] = {cf_lds32<0>(ba), cf_lds32<LDT * 4>(ba), cf_lds32<2 * LDT * 4>(ba), cf_lds32<3 * LDT * 4>(ba)}, b1[4];
    tn_steps<0>(acc, aa, ba, a0, b0, a1, b1);
}

// acc[32 (n) x 32 (k)] += sum_rows dZ[row][n0 + i] * X[row][k0 + j]   over the TM rows of the tile
template <int S>
__device__ __forceinline__ void colred_steps(f32x16& acc, unsigned aa, unsigned ba, float (&a0)[4], float (&b0)[4], float (&a1)[4], float (&b1)[4]) {
    if constexpr (S < TM / 8) {
        float (&ca)[4] = (S & 1) ? a1 : a0; float (&cb)[4] = (S & 1) ? b1 : b0;
        if constexpr (S + 1 < TM / 8) {
            float (&na)[4] = (S & 1) ? a0 : a1; float (&nb)[4] = (S & 1) ? b0 : b1;
            na[0] = cf_lds32<2 * (4 * (S + 1) + 0) * LDT * 4>(aa); nb[0] = cf_lds32<2 * (4 * (S + 1) + 0) * LDT * 4>(ba);
            na[1] = cf_lds32<2 * (4 * (S + 1) + 1) * LDT * 4>(aa); nb[1] = cf_lds32<2 * (4 * (S + 1) + 1) * LDT * 4>(ba);
            na[2] = cf_lds32<2 * (4 * (S + 1) + 2) * LDT * 4>(aa); nb[2] = cf_lds32<2 * (4 * (S + 1) + 2) * LDT * 4>(ba);
            na[3] = cf_lds32<2 * (4 * (S + 1) + 3) * LDT * 4>(aa); nb[3] = cf_lds32<2 * (4 * (S + 1) + 3) * LDT * 4>(ba);
            cf_wait<8>(ca, cb);
        } else {
            cf_wait<0>(ca, cb);
        }
        acc = mfma32(ca[0], cb[0], acc);
        acc = mfma32(ca[1], cb[1], acc);
        acc = mfma32(ca[2], cb[2], acc);
        acc = mfma32(ca[3], cb[3], acc);
        colred_steps<S + 1>(acc, aa, ba, a0, b0, a1, b1);
    }
}
__device__ __forceinline__ void colred_hand(f32x16& acc, const float* Zs_n0, const float* Xs_k0) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const unsigned aa = cf_lds_addr(Zs_n0 + h * LDT + r), ba = cf_lds_addr(Xs_k0 + h * LDT + r);
    float a0[4], b0[4], a1[4], b1[4];
    a0[0] = cf_lds32<0>(aa); b0[0] = cf_lds32<0>(ba);
    a0[1] = cf_lds32<2 * LDT * 4>(aa); b0[1] = cf_lds32<2 * LDT * 4>(ba);
    a0[2] = cf_lds32<4 * LDT * 4>(aa); b0[2] = cf_lds32<4 * LDT * 4>(ba);
    a0[3] = cf_lds32<6 * LDT * 4>(aa); b0[3] = cf_lds32<6 * LDT * 4>(ba);
    colred_steps<0>(acc, aa, ba, a0, b0, a1, b1);
}

// compiler-scheduled twins (the C++ loops of round 1).  Both forms produce the same bits; which one a launch uses is the HAND template
// parameter of k_mlp: the hand-ordered forms win when the kernel has the GPU to itself (actor pass of config 3: 1.88 -> 1.85 ms,
// value pass of config 4: 1.49 -> 1.31 ms) and lose when a second kernel shares the CUs (512-env share with the critic on the
// second stream: 1.54 -> 1.69 ms per iteration; config 4's two-chunk actor: +4 %), so launch_variant picks them by shape and size.
__device__ __forceinline__ void rowpar_nt(f32x16& acc, const float* As, const float* Bs, int kb) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float4* ap = reinterpret_cast<const float4*>(As + r * LDT + 4 * h);
    const float4* bp = reinterpret_cast<const float4*>(Bs + r * LDT + 4 * h);
    float4 a = ap[0], b = bp[0];
    for (int j = 0; j < kb; ++j) {
        float4 an = a, bn = b;
        if (j + 1 < kb) { an = ap[2 * (j + 1)]; bn = bp[2 * (j + 1)]; }
        acc = mfma32(a.x, b.x, acc);
        acc = mfma32(a.y, b.y, acc);
        acc = mfma32(a.z, b.z, acc);
        acc = mfma32(a.w, b.w, acc);
        a = an; b = bn;
    }
}

// acc[32x32] += dZ[32 rows][64 (n)] * W[64 (n)][32 cols]  (W row-major [n][k] in LDS, read transposed)
__device__ __forceinline__ void rowpar_tn(f32x16& acc, const float* As, const float* Ws_c0) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float4* ap = reinterpret_cast<const float4*>(As + r * LDT + 4 * h);
    const float* bp = Ws_c0 + (4 * h) * LDT + r;
    float4 a = ap[0];
    float b0 = bp[0], b1 = bp[LDT], b2 = bp[2 * LDT], b3 = bp[3 * LDT];
#pragma unroll
    for (int j = 0; j < HP / 8; ++j) {
        float4 an = a;
        float n0 = b0, n1 = b1, n2 = b2, n3 = b3;
        if (j + 1 < HP / 8) {
            an = ap[2 * (j + 1)];
            n0 = bp[(8 * (j + 1) + 0) * LDT]; n1 = bp[(8 * (j + 1) + 1) * LDT];
            n2 = bp[(8 * (j + 1) + 2) * LDT]; n3 = bp[(8 * (j + 1) + 3) * LDT];
        }
        acc = mfma32(a.x, b0, acc);
        acc = mfma32(a.y, b1, acc);
        acc = mfma32(a.z, b2, acc);
        acc = mfma32(a.w, b3, acc);
        a = an; b0 = n0; b1 = n1; b2 = n2; b3 = n3;
    }
}

// acc[32 (n) x 32 (k)] += sum_rows dZ[row][n0 + i] * X[row][k0 + j]   over the TM rows of the tile
__device__ __forceinline__ void colred(f32x16& acc, const float* Zs_n0, const float* Xs_k0) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float* ap = Zs_n0 + h * LDT + r;
    const float* bp = Xs_k0 + h * LDT + r;
    float a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = ap[2 * i * LDT]; b[i] = bp[2 * i * LDT]; }
#pragma unroll
    for (int kk = 0; kk < TM / 2; kk += 4) {
        float an[4], bn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            an[i] = a[i]; bn[i] = b[i];
            if (kk + 4 < TM / 2) { an[i] = ap[2 * (kk + 4 + i) * LDT]; bn[i] = bp[2 * (kk + 4 + i) * LDT]; }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = mfma32(a[i], b[i], acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = an[i]; b[i] = bn[i]; }
    }
}

template <bool HAND> __device__ __forceinline__ void rowpar_nt_sel(f32x16& acc, const float* As, const float* Bs, int kb) {
    if constexpr (HAND) rowpar_nt_hand(acc, As, Bs, kb); else rowpar_nt(acc, As, Bs, kb);
}
template <bool HAND> __device__ __forceinline__ void rowpar_tn_sel(f32x16& acc, const float* As, const float* Ws_c0) {
    if constexpr (HAND) rowpar_tn_hand(acc, As, Ws_c0); else rowpar_tn(acc, As, Ws_c0);
}
template <bool HAND> __device__ __forceinline__ void colred_sel(f32x16& acc, const float* Zs_n0, const float* Xs_k0) {
    if constexpr (HAND) colred_hand(acc, Zs_n0, Xs_k0); else colred(acc, Zs_n0, Xs_k0);
}

// ------------------------------------------------------------------------------------------------------------------
// Error-compensated bf16 MFMA (opt-in, CM_MFMA=bf16x3; docs/KERNEL_NOTES.md section 8).  Every fp32 value that lives in an LDS operand
// tile is stored as a SPLIT WORD  bf16(x) << 16 | bf16(x - bf16(x))  (same 32-bit footprint, same layouts); the three GEMM
// forms read 8 split words per operand fragment, separate them into a hi and a lo bf16x8 fragment with v_perm_b32 and issue
// lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_bf16 (fp32 accumulate, small terms first).  ~17x fp32 round-off per product
// (2.7e-6 of sum|a b|), 2.95x the fp32 MFMA loop (profiles/r01_h_mfma_split_probe.txt).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
union FragBF { u32x4 u; bf16x8 b; };
template <bool BF> __device__ __forceinline__ float enc(float x) {  // value -> LDS word (bit pattern carried in a float)
    if (!BF) return x;
    const __bf16 hb = (__bf16)x;  // v_cvt_pk_bf16_f32 (round to nearest even)
    const __bf16 lb = (__bf16)(x - (float)hb);
    return __uint_as_float(((unsigned)__builtin_bit_cast(unsigned short, hb) << 16) | (unsigned)__builtin_bit_cast(unsigned short, lb));
}
template <bool BF> __device__ __forceinline__ float dec(float w) {  // LDS word -> value
    if (!BF) return w;
    const unsigned u = __float_as_uint(w);
    return __uint_as_float(u & 0xFFFF0000u) + __uint_as_float(u << 16);
}
__device__ __forceinline__ void frag_split(const unsigned (&w)[8], FragBF& hi, FragBF& lo) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        hi.u[e] = __builtin_amdgcn_perm(w[2 * e + 1], w[2 * e], 0x07060302u);
        lo.u[e] = __builtin_amdgcn_perm(w[2 * e + 1], w[2 * e], 0x05040100u);
    }
}
// ONE: single-pass bf16 (option mfma=bf16: the hi halves only -- one MFMA per product, ~2^-8 relative operand rounding, fp32 accumulate;
// its own, looser parity tier, tests/test_hip_parity.py) instead of the error-compensated three
template <bool ONE = false>
__device__ __forceinline__ f32x16 mfma_bf3(const unsigned (&aw)[8], const unsigned (&bw)[8], f32x16 acc) {
    FragBF ahi, alo, bhi, blo;
    frag_split(aw, ahi, alo);
    frag_split(bw, bhi, blo);
    if (!ONE) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo.b, bhi.b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi.b, blo.b, acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi.b, bhi.b, acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ void ld8(unsigned (&w)[8], const float* p) {  // 8 consecutive words (two b128)
    *reinterpret_cast<u32x4*>(w) = *reinterpret_cast<const u32x4*>(p);
    *reinterpret_cast<u32x4*>(w + 4) = *reinterpret_cast<const u32x4*>(p + 4);
}
__device__ __forceinline__ void ld8s(unsigned (&w)[8], const float* p, int stride) {  // 8 words down a column
#pragma unroll
    for (int e = 0; e < 8; ++e) w[e] = __float_as_uint(p[e * stride]);
}
// split-word twins of rowpar_nt / rowpar_tn / colred: k (or the contracted row index) is consumed 16 at a time, lane half h
// owning elements [16j + 8h, 16j + 8h + 8) of BOTH operands
template <bool ONE = false>
__device__ __forceinline__ void rowpar_nt_bf(f32x16& acc, const float* As, const float* Bs, int k16) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float* ap = As + r * LDT + 8 * h;
    const float* bp = Bs + r * LDT + 8 * h;
    for (int j = 0; j < k16; ++j) {
        unsigned aw[8], bw[8];
        ld8(aw, ap + 16 * j);
        ld8(bw, bp + 16 * j);
        acc = mfma_bf3<ONE>(aw, bw, acc);
    }
}
template <bool ONE = false>
__device__ __forceinline__ void rowpar_tn_bf(f32x16& acc, const float* As, const float* Ws_c0) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float* ap = As + r * LDT + 8 * h;
    const float* bp = Ws_c0 + (8 * h) * LDT + r;
#pragma unroll
    for (int j = 0; j < HP / 16; ++j) {
        unsigned aw[8], bw[8];
        ld8(aw, ap + 16 * j);
        ld8s(bw, bp + 16 * j * LDT, LDT);
        acc = mfma_bf3<ONE>(aw, bw, acc);
    }
}
template <bool ONE = false>
__device__ __forceinline__ void colred_bf(f32x16& acc, const float* Zs_n0, const float* Xs_k0) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float* ap = Zs_n0 + (8 * h) * LDT + r;
    const float* bp = Xs_k0 + (8 * h) * LDT + r;
#pragma unroll
    for (int j = 0; j < TM / 16; ++j) {
        unsigned aw[8], bw[8];
        ld8s(aw, ap + 16 * j * LDT, LDT);
        ld8s(bw, bp + 16 * j * LDT, LDT);
        acc = mfma_bf3<ONE>(aw, bw, acc);
    }
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// logits[16 rows of this wave][16 head outputs] = H_L[rows][64] * Wout[16][64]^T on v_mfma_f32_16x16x4_f32
// (32-cycle issue; 16 of them per tile).  k is consumed in the permuted order {16j + 4g + i}: lane group g = lane>>4
// reads floats [16j + 4g, 16j + 4g + 4) of its row as one b128.  Result: lane (n = lane & 15, g) holds rows 4g..4g+3.
template <bool BF = false>
__device__ __forceinline__ f32x4 head_logits_mfma(const float* HLw /* H_L + 16*wave*LDT */, const float* wts /* wouts + 16*ct*WLD */) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const float4* ap = reinterpret_cast<const float4*>(HLw + n * LDT + 4 * g);
    const float4* bp = reinterpret_cast<const float4*>(wts + n * WLD + 4 * g);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < HP / 16; ++j) {
        const float4 a = ap[4 * j], b = bp[4 * j];
        acc = mfma16(dec<BF>(a.x), b.x, acc);
        acc = mfma16(dec<BF>(a.y), b.y, acc);
        acc = mfma16(dec<BF>(a.z), b.z, acc);
        acc = mfma16(dec<BF>(a.w), b.w, acc);
    }
    return acc;
}

// head weight gradient on the 16x16x4 MFMA: acc[16 (k) x 16 (c)] += sum over the tile's 64 rows of
// dlogits[row][k0 + k] * H[row][c0 + c].  One wave owns one 16-column slice of the hidden units for ALL rows, so no
// cross-wave combine is needed and the accumulator is 4 registers (the 32x32 form spent 16 MFMAs x 64 cycles on a
// K = 5 head; this spends 16 x 32).
template <int KP, bool BF = false>
__device__ __forceinline__ void colred_head16(f32x4& acc, const float* ls, int k0, const float* Hs_c0) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const bool kok = (k0 + n) < KP;
    const float* ap = ls + g * KP + (kok ? k0 + n : 0);
    const float* bp = Hs_c0 + g * LDT + n;
#pragma unroll
    for (int kk = 0; kk < TM / 4; ++kk) {
        const float av = kok ? ap[4 * kk * KP] : 0.0f;
        acc = mfma16(av, dec<BF>(bp[4 * kk * LDT]), acc);
    }
}

// ---- wave-private head backward (CM_HEAD_WP, round 4).  The head's three backward products used to span the workgroup: dWout contracted
// over all 64 rows of the tile per wave (16 hidden columns each) and dZ_L was a 32 x 32 tile per wave, so the dlogits of every wave had to be
// published behind a workgroup barrier, and the in-place relu' write of dZ_L waited behind a second one (other waves still read H_L for
// dWout).  Here every wave finishes ITS 16 rows alone: dWout_w[k][all 64 c] over its own rows (four 16-column tiles: 16 accumulator
// registers per 16 head rows instead of 4, summed over the waves once per launch), dZ_L for its own rows on the 16x16x4 MFMA (result lane
// (c, g) = rows 4g .. 4g+3: written in place at once -- nobody else reads these rows before the barrier that precedes the hidden layers'
// backward), the head bias gradient from the lanes' own registers.  Two of the tile's eleven workgroup barriers go, and the waves of a
// CU drift apart through the head (one wave's products under another's softmax).  Same MFMA count (16 + 8 x KP / 8 16x16x4 per wave).
template <int KP, int NQ, bool BF = false>
__device__ __forceinline__ void head_bwd_wave(f32x4 (&accWo)[NQ][4], const float* lsw /* ls + 16 * wave * KP */, float* HLw /* H_L + 16 * wave * LDT */,
                                              const float* wouts) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    // ---- dWout_w[16 q + k][16 ct + c] += sum over the wave's 16 rows of dlogits[row][16 q + k] * H_L[row][16 ct + c]
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        float hv[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) hv[ct] = dec<BF>(HLw[(4 * kk + g) * LDT + 16 * ct + n]);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const float av = (16 * q + n < KP) ? lsw[(4 * kk + g) * KP + 16 * q + n] : 0.0f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) accWo[q][ct] = mfma16(av, hv[ct], accWo[q][ct]);
        }
    }
    // ---- dZ_L[row][c] = sum_k dlogits[row][k] * Wout[k][c] for the wave's rows, then .* relu'(H_L) in place
    f32x4 dz[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) dz[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < KP / 4; ++j) {
        const float av = lsw[n * KP + 4 * j + g];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) dz[ct] = mfma16(av, wouts[(4 * j + g) * WLD + 16 * ct + n], dz[ct]);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* p = HLw + (4 * g + r) * LDT + 16 * ct + n;
            *p = enc<BF>((dec<BF>(*p) > 0.0f) ? dz[ct][r] : 0.0f);
        }
}

// ---- K <= 8 training heads on v_mfma_f32_4x4x1_16b_f32 (CM_HEAD_44, round 6).  One instruction = 16 independent 4 x 4 outer products: lane 4 b + x supplies
// A[b][i = x] and B[b][j = x], register i of lane 4 b + j receives D[b][i][j] (tools/probes/mfma4x4_layout.hip).  The same 32 MAC per cycle as the larger forms
// (8.75 cycles per instruction from two accumulator chains on), but the padding of a K = 5 head is 8 columns instead of 16: the wave's three head
// products take 32 + 32 + 4 K instructions x 8.75 cycles = 735 cycles at K = 5 where the 16x16x4 forms took 40 x 32 = 1280.
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float row_ror8(float v) {  // lane l of a 16-lane row <- lane l ^ 8
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
}

// The 4x4x1 products are short (4 MFMAs = 35 cycles per 16-byte operand pair), so their LDS reads are hand-issued (cm_common.h: fixed issue order, counted
// waits) several steps ahead; the "memory" clobber keeps the compiler's own LDS stores (the dlogits the head math just wrote) on their side of the reads.
template <int OFF> __device__ __forceinline__ f32x4 hd_lds128(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OFF0, int OFF1> __device__ __forceinline__ f32x2 hd_lds32x2(unsigned addr) {  // offsets in 4-byte units
    f32x2 v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(addr), "n"(OFF0), "n"(OFF1) : "memory");
    return v;
}

// logits of the wave's 16 rows, 8 head outputs: block (rg, kh, og) = rows 4 rg .. 4 rg + 3 x outputs 4 og .. 4 og + 3 over the k half [32 kh, 32 kh + 32)
// (lane = 16 rg + 8 kh + 4 og + x); the bias is the initial accumulator of the kh = 0 blocks, the two halves meet through one DPP row rotation, the kh = 0
// lanes write lsw[row][k] (row stride 8).  Operand pairs in a ring of four, issued four steps ahead.
template <bool BF = false>
__device__ __forceinline__ void head_logits44(float* lsw /* ls + 16 * wave * 8 */, const float* HLw /* H_L + 16 * wave * LDT */, const float* wouts,
                                              const float* bout) {
    static_assert(HP == 64, "eight 16-byte steps per k half");
    const int lane = threadIdx.x & 63, x = lane & 3, og = (lane >> 2) & 1, kh = (lane >> 3) & 1, rg = lane >> 4;
    const float b0 = kh ? 0.0f : bout[4 * og + x];
    const unsigned aa = cf_lds_addr(HLw + (4 * rg + x) * LDT + 32 * kh), ba = cf_lds_addr(wouts + (4 * og + x) * WLD + 32 * kh);
    f32x4 acc0 = {b0, b0, b0, b0}, acc1 = {0.f, 0.f, 0.f, 0.f};
    f32x4 a[4], b[4];
#define CM_LG_ISSUE(m) do { a[(m) & 3] = hd_lds128<16 * (m)>(aa); b[(m) & 3] = hd_lds128<16 * (m)>(ba); } while (0)
#define CM_LG_STEP(m, N) do { cf_wait<N>(a[(m) & 3], b[(m) & 3]); \
        acc0 = mfma4(dec<BF>(a[(m) & 3][0]), b[(m) & 3][0], acc0); acc1 = mfma4(dec<BF>(a[(m) & 3][1]), b[(m) & 3][1], acc1); \
        acc0 = mfma4(dec<BF>(a[(m) & 3][2]), b[(m) & 3][2], acc0); acc1 = mfma4(dec<BF>(a[(m) & 3][3]), b[(m) & 3][3], acc1); } while (0)
    CM_LG_ISSUE(0); CM_LG_ISSUE(1); CM_LG_ISSUE(2); CM_LG_ISSUE(3);
    CM_LG_STEP(0, 6); CM_LG_ISSUE(4);
    CM_LG_STEP(1, 6); CM_LG_ISSUE(5);
    CM_LG_STEP(2, 6); CM_LG_ISSUE(6);
    CM_LG_STEP(3, 6); CM_LG_ISSUE(7);
    CM_LG_STEP(4, 6); CM_LG_STEP(5, 4); CM_LG_STEP(6, 2); CM_LG_STEP(7, 0);
#undef CM_LG_ISSUE
#undef CM_LG_STEP
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = acc0[i] + acc1[i];
        v += row_ror8(v);
        if (!kh) lsw[(4 * rg + i) * 8 + 4 * og + x] = v;
    }
}

// head backward of the wave's 16 rows (see head_bwd_wave): dWout partial in accWo[m] -- block (rh, kg, cg) of lane 4 b + x, b = 8 rh + 4 kg + cg: register i =
// dWout[4 kg + i][16 cg + 4 x + m] over the rows 8 rh .. 8 rh + 7 (the two row halves are separate partials, summed with the waves' at the end of the
// launch); dZ_L -- block (rg, cg), b = 4 rg + cg: dz[m] register i = dZ_L[4 rg + i][16 cg + 4 x + m], written in place on 16-byte accesses.  All 8 head rows of
// the weight image take part (rows >= dout are zero): no branch inside the hand-issued pipeline.  At most 14 reads in flight (lgkmcnt counts to 15).
template <bool BF = false>
__device__ __forceinline__ void head_bwd_wave44(f32x4 (&accWo)[4], const float* lsw, float* HLw, const float* wouts) {
    const int lane = threadIdx.x & 63, x = lane & 3, b = lane >> 2, cg = b & 3, kg = (b >> 2) & 1, rh = b >> 3, rg = b >> 2;
    const unsigned da = cf_lds_addr(lsw + 8 * rh * 8 + 4 * kg + x);          // dlogits[8 rh + s][4 kg + x]: + 32 s bytes
    const unsigned ha = cf_lds_addr(HLw + 8 * rh * LDT + 16 * cg + 4 * x);   // H_L[8 rh + s][16 cg + 4 x ..]: + 4 LDT s bytes
    const unsigned za = cf_lds_addr(lsw + (4 * rg + x) * 8);                 // dlogits[4 rg + x][0 .. 7]
    const unsigned wa = cf_lds_addr(wouts + 16 * cg + 4 * x);                // Wout[k][16 cg + 4 x ..]: + 4 WLD k bytes
    const unsigned pa = cf_lds_addr(HLw + 4 * rg * LDT + 16 * cg + 4 * x);   // H_L[4 rg + i][16 cg + 4 x ..]: + 4 LDT i bytes
    f32x2 d2[4];
    f32x4 hv[8], az[2], wv[8], hz[4];
    d2[0] = hd_lds32x2<0, 8>(da); d2[1] = hd_lds32x2<16, 24>(da); d2[2] = hd_lds32x2<32, 40>(da); d2[3] = hd_lds32x2<48, 56>(da);
#define CM_HB_H(s_) hv[s_] = hd_lds128<4 * LDT * (s_)>(ha)
#define CM_HB_W(k_) wv[k_] = hd_lds128<4 * WLD * (k_)>(wa)
#define CM_HB_DW(s_, N) do { cf_wait<N>(hv[s_]); const float av_ = d2[(s_) >> 1][(s_) & 1]; \
        accWo[0] = mfma4(av_, dec<BF>(hv[s_][0]), accWo[0]); accWo[1] = mfma4(av_, dec<BF>(hv[s_][1]), accWo[1]); \
        accWo[2] = mfma4(av_, dec<BF>(hv[s_][2]), accWo[2]); accWo[3] = mfma4(av_, dec<BF>(hv[s_][3]), accWo[3]); } while (0)
    CM_HB_H(0); CM_HB_H(1); CM_HB_H(2); CM_HB_H(3); CM_HB_H(4); CM_HB_H(5); CM_HB_H(6); CM_HB_H(7);   // 12 in flight
    asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(d2[0]), "+v"(d2[1]), "+v"(d2[2]), "+v"(d2[3]));      // the four dlogits pairs
    CM_HB_DW(0, 7); CM_HB_DW(1, 6); CM_HB_DW(2, 5); CM_HB_DW(3, 4);
    az[0] = hd_lds128<0>(za); az[1] = hd_lds128<16>(za);
    CM_HB_W(0); CM_HB_W(1); CM_HB_W(2); CM_HB_W(3); CM_HB_W(4);                                     // 4 + 7 in flight
    CM_HB_DW(4, 10); CM_HB_DW(5, 9); CM_HB_DW(6, 8); CM_HB_DW(7, 7);
    CM_HB_W(5); CM_HB_W(6); CM_HB_W(7);
    hz[0] = hd_lds128<0>(pa); hz[1] = hd_lds128<4 * LDT>(pa); hz[2] = hd_lds128<8 * LDT>(pa); hz[3] = hd_lds128<12 * LDT>(pa);   // 14 in flight
    f32x4 dz[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) dz[m] = f32x4{0.f, 0.f, 0.f, 0.f};
#define CM_HB_DZ(k_, N) do { cf_wait<N>(az[(k_) >> 2], wv[k_]); const float av_ = az[(k_) >> 2][(k_) & 3]; \
        dz[0] = mfma4(av_, wv[k_][0], dz[0]); dz[1] = mfma4(av_, wv[k_][1], dz[1]); \
        dz[2] = mfma4(av_, wv[k_][2], dz[2]); dz[3] = mfma4(av_, wv[k_][3], dz[3]); } while (0)
    CM_HB_DZ(0, 11); CM_HB_DZ(1, 10); CM_HB_DZ(2, 9); CM_HB_DZ(3, 8); CM_HB_DZ(4, 7); CM_HB_DZ(5, 6); CM_HB_DZ(6, 5); CM_HB_DZ(7, 4);
    cf_wait<0>(hz[0], hz[1]); cf_wait<0>(hz[2], hz[3]);
#undef CM_HB_H
#undef CM_HB_W
#undef CM_HB_DW
#undef CM_HB_DZ
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 o;
        o.x = enc<BF>((dec<BF>(hz[i][0]) > 0.0f) ? dz[0][i] : 0.0f);
        o.y = enc<BF>((dec<BF>(hz[i][1]) > 0.0f) ? dz[1][i] : 0.0f);
        o.z = enc<BF>((dec<BF>(hz[i][2]) > 0.0f) ? dz[2][i] : 0.0f);
        o.w = enc<BF>((dec<BF>(hz[i][3]) > 0.0f) ? dz[3][i] : 0.0f);
        *reinterpret_cast<float4*>(HLw + (4 * rg + i) * LDT + 16 * cg + 4 * x) = o;
    }
}

// acc[32 rows x 32 cols] = dlogits[32 rows][KP] * Wout[KP][32 cols]   (backward through the head, K = KP)
template <int KP>
__device__ __forceinline__ void head_bwd_mfma(f32x16& acc, const float* ls_r0, const float* wts_c0) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < KP / 8; ++j) {
        const float4 a = *reinterpret_cast<const float4*>(ls_r0 + r * KP + 8 * j + 4 * h);
        const float* bp = wts_c0 + (8 * j + 4 * h) * WLD + r;
        acc = mfma32(a.x, bp[0], acc);
        acc = mfma32(a.y, bp[WLD], acc);
        acc = mfma32(a.z, bp[2 * WLD], acc);
        acc = mfma32(a.w, bp[3 * WLD], acc);
    }
}

// acc[32 (k) x 32 (c)] += sum over 32 rows of dlogits[row][k] * H[row][c0 + c]   (head weight gradient)
__device__ __forceinline__ void colred_head(f32x16& acc, const float* ls_r0, const float* Hs_r0_c0) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float* ap = ls_r0 + h * LSP + r;
    const float* bp = Hs_r0_c0 + h * LDT + r;
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) acc = mfma32(ap[2 * kk * LSP], bp[2 * kk * LDT], acc);
}

template <bool BF = false>
__device__ __forceinline__ void stage_rows(float* dst, const float* src, long row0, long nrows, long stride,
                                           int col0, int ncols) {
    // dst[r][k] = src[(row0+r)*stride + col0 + k]  for r < TM, k < KC; zero outside [nrows) x [ncols)
    // thread (r0 = tid/64, k = tid%64) walks rows r0, r0+4, ...: one pointer bump per element, no multiplies
    const int k = threadIdx.x & 63, r0 = threadIdx.x >> 6;
    const bool kok = k < ncols;
    const float* p = src + (row0 + r0) * stride + col0 + k;
    const long step = 4 * stride;
    float v[TM / 4];
#pragma unroll
    for (int i = 0; i < TM / 4; ++i) {
        v[i] = (kok && row0 + r0 + 4 * i < nrows) ? *p : 0.0f;
        p += step;
    }
#pragma unroll
    for (int i = 0; i < TM / 4; ++i) dst[(r0 + 4 * i) * LDT + k] = enc<BF>(v[i]);
}

// ---- register-staged tile prefetch (T14 "issue early / write late"): a 64x64 fp32 tile = 16 floats per thread.
// The loads are issued one phase (or one whole tile) ahead of the ds_write that consumes them, so HBM latency
// hides under the MFMA phases in between; plain global loads stay in flight across s_barrier.
struct Tile16 { float4 v[4]; };

template <bool VEC>
__device__ __forceinline__ void tile_load(Tile16& t, const float* src, long row0, long nrows, long stride, int col0, int ncols) {
    if (VEC) {  // 16-byte loads: rows 16-byte aligned, ncols % 4 == 0
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = threadIdx.x + NTHREADS * i;
            const int r = idx >> 4, c4 = (idx & 15) * 4;
            const long row = row0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < nrows && c4 < ncols) v = *reinterpret_cast<const float4*>(src + row * stride + col0 + c4);
            t.v[i] = v;
        }
    } else {
        float* f = reinterpret_cast<float*>(t.v);
        const int k = threadIdx.x & 63, r0 = threadIdx.x >> 6;
        const bool kok = k < ncols;
        const float* p = src + (row0 + r0) * stride + col0 + k;
        const long step = 4 * stride;
#pragma unroll
        for (int i = 0; i < 16; ++i) {  // element i = row r0 + 4i, column k: one pointer bump per load
            f[i] = (kok && row0 + r0 + 4 * i < nrows) ? *p : 0.0f;
            p += step;
        }
    }
}

template <bool VEC, bool BF = false>
__device__ __forceinline__ void tile_store(float* dst, const Tile16& t) {
    if (VEC) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = threadIdx.x + NTHREADS * i;
            const int r = idx >> 4, c4 = (idx & 15) * 4;
            if (BF) {
                const float4 v = t.v[i];
                *reinterpret_cast<float4*>(dst + r * LDT + c4) = make_float4(enc<true>(v.x), enc<true>(v.y), enc<true>(v.z), enc<true>(v.w));
            } else {  // keep this a DIRECT member-to-LDS store: routing it through a local float4 changed the register allocation
                      // of the fp32 kernels (a prefetch register got recycled as a zero constant -> s_waitcnt vmcnt(0) right
                      // after the prefetch was issued) and cost 2 % of the actor kernel
                *reinterpret_cast<float4*>(dst + r * LDT + c4) = t.v[i];
            }
        }
    } else {
        const float* f = reinterpret_cast<const float*>(t.v);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = threadIdx.x + NTHREADS * i;
            dst[(idx >> 6) * LDT + (idx & 63)] = enc<BF>(f[i]);
        }
    }
}

#ifdef CM_PHASE_PROF
#define PH_DECL unsigned long long ph_[16] = {0}; unsigned long long ph_t0 = __builtin_amdgcn_s_memtime();
#define PH(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ph_[i] += t_ - ph_t0; ph_t0 = t_; } while (0)
#define PH_FLUSH do { if (a.prof && threadIdx.x == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) a.prof[(size_t)blockIdx.x * 16 + i_] = ph_[i_]; } } while (0)
#else
#define PH_DECL
#define PH(i)
#define PH_FLUSH
#endif

// per-row inputs of the loss heads, fetched at the top of a tile so their HBM latency hides under the MFMA phases
template <int KJ> struct RowIn { int act; float lpo, adv, ret; int eplen, ag, e, t; unsigned char avb[KJ]; };

// VEC: 0 = 4-byte tile loads; 1 = 16-byte loads of the input rows and of streamed W0 chunks (can_vec)
// B-operand register image of W[n0 .. n0+31][k0 .. k0+63] for the 32x32x2 products: w[4j + i] = W[(n0 + r) * ld + k0 + 8j + 4h + i]
// (lane r = lane & 31, h = lane >> 5), zero outside [nrows) x [ncols); and the product on it, k order identical to rowpar_nt -- same bits.
__device__ __forceinline__ void w0_regs_load(float (&w)[32], const float* W, int n0, int nrows, long ld, int k0, int ncols) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int n = n0 + r;
    const float* p = W + (long)n * ld + k0 + 4 * h;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) w[4 * j + i] = (n < nrows && k0 + 8 * j + 4 * h + i < ncols) ? p[8 * j + i] : 0.0f;
}
__device__ __forceinline__ void rowpar_regb(f32x16& acc, const float* As, const float (&w)[32], int kb) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float4* ap = reinterpret_cast<const float4*>(As + r * LDT + 4 * h);
    float4 a = ap[0];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (j < kb) {
            float4 an = a;
            if (j + 1 < kb) an = ap[2 * (j + 1)];
            acc = mfma32(a.x, w[4 * j], acc);
            acc = mfma32(a.y, w[4 * j + 1], acc);
            acc = mfma32(a.z, w[4 * j + 2], acc);
            acc = mfma32(a.w, w[4 * j + 3], acc);
            a = an;
        }
    }
}

// NCH: > 0 = training kernels compiled for that many 64-column input chunks; 0 = forward kernels, any width, W0 chunks streamed through
// LDS per tile when the input is wider than one chunk; -2 = forward kernels for EXACTLY two chunks (65 .. 128 columns: config 4's 115) with
// W0 as B operands in registers for the whole launch (64 registers) -- the streamed form re-staged two 16 KB W0 chunks for every 64-row
// tile: 42 % of config 4's act pass (profiles/r03_phase_act.txt)
template <int NCH, int MODE, int VEC, int LCAP, int KJ, bool BF = false, bool HAND = false>
__global__ __launch_bounds__(NTHREADS, wgs_per_cu(NCH)) void k_mlp(const MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool TRAIN = (MODE >= M_ACTOR);
    constexpr int NC = (NCH > 0 ? NCH : 1);
    const Offsets off = make_offsets(a.din, a.H, a.L, a.dout);
    const Lds lds = make_lds(a.L, a.dout, (a.din + KC - 1) / KC);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, lc = lane & 31;
    const int H = a.H, L = a.L, dout = a.dout, din = a.din;
    const int nch = (din + KC - 1) / KC;
    // compile-time for the training instantiations so the streaming paths (and their registers) vanish
    constexpr bool W0REG = (NCH == -2);
    const bool w0_resident = (NCH > 0) ? (NCH == 1) : (nch == 1);
    const bool ws_resident = (LCAP == 1) || (L == 1);
    const float* W0g = a.w0p ? a.w0p : a.params + off.W0;  // streamed W0 chunks (see MlpArgs::w0p); the LDS-resident form reads params
    const int w0ld = a.w0p ? a.w0_ld : din;
    float* Xs = smem + lds.Xs;
    float* W0s = smem + lds.W0s;
    float* Ws = smem + lds.Ws;
    float* wouts = smem + lds.wout;
    float* ls = smem + lds.ls;
    float* red = smem + lds.red;
    constexpr int KP = KJ * 4;           // head outputs handled by this instantiation (8 or 32)
    constexpr int lstride = KP;
    constexpr int WR = (KJ == 2) ? 16 : KMAX;  // rows of the zero-padded head weight image

    // ---- one-time staging of small tensors (+ resident weights)
    for (int i = tid; i < WR * HP; i += NTHREADS) {
        const int k = i / HP, c = i % HP;
        wouts[k * WLD + c] = (c < H && k < dout) ? a.params[off.Wout + k * H + c] : 0.0f;
    }
    for (int i = tid; i < HP; i += NTHREADS) {
        smem[lds.b0 + i] = (i < H) ? a.params[off.b0 + i] : 0.0f;
        for (int l = 0; l < LCAP; ++l)
            if (l < L) smem[lds.bl(l) + i] = (i < H) ? a.params[off.bl(l) + i] : 0.0f;
    }
    for (int i = tid; i < KMAX; i += NTHREADS) smem[lds.bout + i] = (i < dout) ? a.params[off.bout + i] : 0.0f;
    if (w0_resident) stage_rows<BF>(W0s, a.params + off.W0, 0, H, din, 0, din);
    if (ws_resident && L >= 1) stage_rows<BF>(Ws, a.params + off.Wl(0), 0, H, H, 0, H);

    // ---- persistent accumulators (training)
    f32x16 accW0[NC];
    f32x16 accWl[LCAP];
    // wave-private head backward where its 12 extra accumulator registers fit (K <= 8 heads, one input chunk or the split schedule, <= 1
    // hidden->hidden layer: 208 -> 219 registers for the actor pass of config 3); K > 8 heads (+ 24 .. 32 registers) and two-chunk / deep
    // instantiations sit at the 256-register limit of two workgroups per CU and would spill (config 4's actor: 4 -> 64 spilled registers)
    constexpr bool WP = (CM_HEAD_WP != 0) && KJ == 2 && NCH <= 1 && LCAP == 1;
    // the three head products on the 4x4x1 MFMA: only in the hand-ordered instantiations, i.e. launches that have the GPU to themselves (launch_variant:
    // >= 2^21 rows).  Its 14-deep operand pipelines cost 26 registers (221 -> 247): beside the critic's kernels the two actor workgroups of a CU then
    // leave no register slot for the small launches of the other stream (fold + step, scan), which wait for a whole actor pass -- measured on the
    // 512-env share: 1.34 -> 1.475 ms with the new head, although the pass alone is 2 % faster (profiles/r06_head_4x4_ab.txt).  Forward-only modes
    // keep the 16x16x4 logits: their samplers are held bit-identical to the fused rollouts'.
    constexpr bool WP44 = WP && TRAIN && HAND && (CM_HEAD_44 != 0);
    f32x4 accWoW[WP ? WR / 16 : 1][4];  // WP: dWout partial of THIS wave's rows, [k (16 per q)][16 hidden cols per ct]; summed over the waves at the end
    float dboW[KJ];                     // WP: head bias gradient from this lane's dlogits (k = 4 j + hq), summed over lanes and waves at the end
    f32x4 accWo[WR / 16];  // !WP: dWout[k (16 per tile) x 16 hidden cols]: wave w owns hidden columns 16w..16w+15
    float dbo = 0.0f;
    float dbh[LCAP + 1];
    float st_pg = 0.f, st_ent = 0.f, st_kl = 0.f, st_clip = 0.f, st_vl = 0.f, st_cnt = 0.f;
    if (TRAIN) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int g = 0; g < 16; ++g) accW0[c][g] = 0.0f;
#pragma unroll
        for (int l = 0; l < LCAP; ++l)
#pragma unroll
            for (int g = 0; g < 16; ++g) accWl[l][g] = 0.0f;
#pragma unroll
        for (int q = 0; q < (WP ? WR / 16 : 1); ++q)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) accWoW[q][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < KJ; ++j) dboW[j] = 0.0f;
#pragma unroll
        for (int q = 0; q < WR / 16; ++q) accWo[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int l = 0; l <= LCAP; ++l) dbh[l] = 0.0f;
    }

    const long ntiles = (a.rows + TM - 1) / TM;
    PH_DECL
    // clock probe (scalar: a kernel argument): two counter reads at entry, four 8-byte stores at exit of every workgroup
    const bool clk_on = (MODE == M_ACTOR) && a.clk != nullptr;
    unsigned long long clk_c0 = 0, clk_r0 = 0;
    if (clk_on) { clk_c0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    const int hrow = tid >> 2, hq = tid & 3;  // head mapping: 4 lanes per row, 16 hidden columns per lane
    const int Aseq = (MODE == M_CRITIC && !a.per_agent) ? 1 : a.A;

    // ---- software pipeline prologue: first X chunk (and W0 chunk when it is streamed) of the first tile
    Tile16 px, pw;
    Tile16 pn;  // NCH == 1 training kernels: the NEXT tile's X, requested mid-tile (px is busy holding this tile's X for dW0)
    constexpr bool EARLY_NEXT = TRAIN && NCH == 1;
#ifndef CM_XMERGE
#define CM_XMERGE 1
#endif
    constexpr bool XMERGE = EARLY_NEXT && (CM_XMERGE != 0);  // X re-staged for dW0 under the last in-place phase (see there)
#ifndef CM_XKEEP
#define CM_XKEEP 1
#endif
    // single-chunk training kernels used to RE-LOAD the tile they had just stored to LDS into the same prefetch registers (an L2 hit, 1 GB per
    // launch at config 3) to have it back for the layer-0 weight gradient; tile_store does not consume the registers, so they simply keep it
    constexpr bool XKEEP = EARLY_NEXT && (CM_XKEEP != 0);
#ifndef CM_ROWIN_LATE
#define CM_ROWIN_LATE 1
#endif
    // measured (gpurun_out/r04k): with the hand-ordered product forms (actor passes above 2^21 rows) 1.729 -> 1.697 ms at config 3; with the
    // compiler-scheduled forms the late requests cost 2 % (512-env share 0.281 -> 0.288 ms per pass, config 2 0.212 -> 0.215): HAND only
    constexpr bool ROWIN_LATE = XKEEP && HAND && (CM_ROWIN_LATE != 0);
    float w0ra[W0REG ? 32 : 1], w0rb[W0REG ? 32 : 1];  // W0REG: this wave's 32 output columns of W0, both input chunks, for the whole launch
    if constexpr (W0REG) {
        w0_regs_load(w0ra, a.params + off.W0, 32 * wn, H, din, 0, din);
        w0_regs_load(w0rb, a.params + off.W0, 32 * wn, H, din, KC, din);
    }
    // Tile range of this workgroup.  Default: tiles blockIdx.x, + grid, ...  With a.split_tiles the two halves of the grid own two
    // CONSECUTIVE tile ranges of unequal size (see set_tile_split): still a static map (the partial sums of a workgroup, and with them the
    // folded gradient, do not depend on timing), but the half that the CU arbiter favours gets more tiles.
    long tile_begin = blockIdx.x, tile_stride = gridDim.x, tile_end = ntiles;
    if (a.split_tiles > 0) {
        const long half = gridDim.x >> 1;
        tile_stride = half;
        if ((long)blockIdx.x < half) tile_end = a.split_tiles; else tile_begin = a.split_tiles + ((long)blockIdx.x - half);
    }
    if (tile_begin < tile_end) {
        tile_load<(VEC != 0)>(px, a.x, tile_begin * TM, a.rows, a.x_stride, 0, min(KC, din));
        if (!w0_resident && !W0REG) tile_load<(VEC == 1)>(pw, W0g, 0, H, w0ld, 0, min(KC, din));
        // W0REG: the W0 staging registers carry the tile's SECOND X chunk instead -- both chunks of a tile are requested a whole tile
        // ahead (one chunk ahead, the second chunk's HBM round trip was exposed behind 2 k cycles of MFMAs)
        if (W0REG) tile_load<(VEC != 0)>(pw, a.x, tile_begin * TM, a.rows, a.x_stride, KC, din - KC);
    }

    for (long tile = tile_begin; tile < tile_end; tile += tile_stride) {
        const long row0 = tile * TM;
        const long next_row0 = (tile + tile_stride < tile_end) ? (tile + tile_stride) * TM : a.rows;  // last tile of the range: the prefetch loads predicate off
        // ================= forward, layer 0 (input chunks) =================
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
        RowIn<KJ> ri;
        ri.act = 0; ri.lpo = 0.f; ri.adv = 0.f; ri.ret = 0.f; ri.eplen = 0; ri.ag = 0; ri.e = 0; ri.t = 0;
#pragma unroll
        for (int j = 0; j < KJ; ++j) ri.avb[j] = 1;
        // COMA's factored critic input: the layer-0 addend of this tile, requested now, consumed in the layer-0 epilogue
        constexpr bool ADD_OK = (MODE == M_FWD || MODE == M_QCRITIC);
        float zadd[ADD_OK ? 16 : 1];
        const bool has_add = ADD_OK && a.z0_add != nullptr;
        if (ADD_OK) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const long row = row0 + 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                zadd[g] = (has_add && row < a.rows) ? a.z0_add[row * HP + 32 * wn + lc] : 0.0f;
            }
        }
        const int grow = (int)row0 + hrow;  // this lane-group's global row (host guarantees rows < 2^31)
        const bool rvalid = grow < (int)a.rows;
        // per-row head inputs of the tile (availability bytes, action / old log-prob / advantage / return, episode length): requested at the top
        // of the tile, consumed in the head phase.  The SAME statements in three places, chosen at compile time -- each placement was measured on
        // the instantiations it applies to (gpurun_out/r04k, r04l; a semantically neutral refactoring of this block moved the actor pass by 2 %
        // either way, see the end of docs/KERNEL_NOTES.md 8b):
        //   * training passes with the hand-ordered product forms and a kept tile (ROWIN_LATE: the actor pass above 2^21 rows): AFTER the
        //     barrier, under the layer-0 products, instead of in the zero-MFMA interval between the two barriers: 1.729 -> 1.697 ms at config 3;
        //   * the other training passes: inline before the tile prefetch (the compiler turns them into booleans right after the layer-0 loop;
        //     vmcnt waits are in issue order, so behind the prefetch they would drag it along) -- the late form cost them 2 %;
        //   * forward / act passes: the same place, through a lambda (config 4's act pass 1.75 -> 1.71 ms with the code the compiler makes of it)
#define CM_FETCH_ROW_INPUTS() do { \
                const bool use_avail = (MODE == M_ACTOR || MODE == M_COMA_ACTOR) || ((MODE == M_ACT || MODE == M_FWD) && a.avail != nullptr); \
                if (rvalid && use_avail) { \
                    const uint8_t* ap = a.avail + (long)grow * a.avail_stride; \
_Pragma("unroll") \
                    for (int j = 0; j < KJ; ++j) \
                        if (4 * j + hq < dout) ri.avb[j] = ap[4 * j + hq]; \
                } \
                if (TRAIN && rvalid) { \
                    const int seq = grow / a.T; \
                    ri.t = grow - seq * a.T; \
                    ri.e = seq / Aseq; \
                    ri.ag = seq - ri.e * Aseq; \
                    ri.eplen = a.ep_len[ri.e]; \
                    if (MODE == M_ACTOR) { ri.act = a.action[grow]; ri.lpo = a.logp_old[grow]; ri.adv = a.adv[grow]; } \
                    else if (MODE == M_COMA_ACTOR) { ri.act = a.action[grow]; ri.adv = a.adv[grow]; } \
                    else if (MODE == M_QCRITIC) { ri.act = a.action[grow]; ri.ret = a.ret[grow]; } \
                    else if (a.per_agent) ri.ret = a.ret[grow]; \
                } \
        } while (0)
        auto fetch_row_inputs = [&]() { CM_FETCH_ROW_INPUTS(); };
        for (int c = 0; c < nch; ++c) {
            __syncthreads();  // previous readers of Xs / W0s are done
            if (W0REG && c == 1) tile_store<(VEC != 0), BF>(Xs, pw); else tile_store<(VEC != 0), BF>(Xs, px);
            if (!w0_resident && !W0REG) tile_store<(VEC == 1), BF>(W0s, pw);
            if (c == 0) {
                if constexpr (!TRAIN) fetch_row_inputs();
                else if constexpr (!ROWIN_LATE) CM_FETCH_ROW_INPUTS();
                else if (L < 1) CM_FETCH_ROW_INPUTS();
            }
            // issue the next chunk's loads now; they land while the MFMAs below (and, for the last chunk,
            // the whole rest of the tile) execute
            {
                const bool last = (c + 1 == nch);
                const int cn = last ? 0 : c + 1;
                const bool again = TRAIN && NCH > 0 && (NCH > 1 || L >= 1);  // backward re-reads X of THIS tile
                const long r0n = last ? (again ? row0 : next_row0) : row0;
                const int wn_ = min(KC, din - cn * KC);
                if (W0REG) {  // chunk c of the NEXT tile into the buffer that was just emptied
                    if (c == 0) tile_load<(VEC != 0)>(px, a.x, next_row0, a.rows, a.x_stride, 0, KC);
                    else tile_load<(VEC != 0)>(pw, a.x, next_row0, a.rows, a.x_stride, KC, din - KC);
                } else if (!(XKEEP && L >= 1))  // XKEEP: px already holds this tile and keeps it until the layer-0 weight gradient re-stages it
                tile_load<(VEC != 0)>(px, a.x, r0n, a.rows, a.x_stride, cn * KC, wn_);
                if (!w0_resident && !W0REG && !(last && TRAIN && NCH > 0)) tile_load<(VEC == 1)>(pw, W0g, 0, H, w0ld, cn * KC, wn_);
            }
            __syncthreads();
            PH(0);
            if constexpr (ROWIN_LATE) { if (c == 0 && L >= 1) CM_FETCH_ROW_INPUTS(); }
            const int w = min(KC, din - c * KC);
            if constexpr (W0REG) { if (c == 0) rowpar_regb(acc, Xs + 32 * wm * LDT, w0ra, (w + 7) >> 3); else rowpar_regb(acc, Xs + 32 * wm * LDT, w0rb, (w + 7) >> 3); }
            else if (BF) rowpar_nt_bf<HAND>(acc, Xs + 32 * wm * LDT, W0s + 32 * wn * LDT, (w + 15) >> 4);  // BF instantiations: HAND = single-pass bf16
            else rowpar_nt_sel<HAND>(acc, Xs + 32 * wm * LDT, W0s + 32 * wn * LDT, (w + 7) >> 3);
        }
        {
            float* H0 = smem + lds.Hs(0);
            const float bias = smem[lds.b0 + 32 * wn + lc];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                float z = acc[g] + bias;
                if (ADD_OK) z += zadd[g];
                const float hv = fmaxf(z, 0.0f);
                H0[row * LDT + 32 * wn + lc] = enc<BF>(hv);
                // forward passes may hand the layer-0 activations to a later pass with the same parameters (cm_value_pass_keep_h0_ld -> the first critic
                // epoch, cm_critic_fused.h): [rows][HP] row-major through MlpArgs::dz0, which no forward kernel uses otherwise
                if constexpr (MODE == M_FWD) { if (a.dz0 != nullptr && row0 + row < a.rows) a.dz0[(row0 + row) * HP + 32 * wn + lc] = hv; }
            }
        }
        __syncthreads();
        PH(1);
        // ================= forward, hidden layers =================
#pragma unroll
        for (int l = 1; l <= LCAP; ++l) {
            if (l <= L) {
                if (!ws_resident) {
                    stage_rows<BF>(Ws, a.params + off.Wl(l - 1), 0, H, H, 0, H);
                    __syncthreads();
                }
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
                if (BF) rowpar_nt_bf<HAND>(acc, smem + lds.Hs(l - 1) + 32 * wm * LDT, Ws + 32 * wn * LDT, HP / 16);
                else rowpar_nt_sel<HAND>(acc, smem + lds.Hs(l - 1) + 32 * wm * LDT, Ws + 32 * wn * LDT, HP / 8);
                float* Hl = smem + lds.Hs(l);
                const float bias = smem[lds.bl(l - 1) + 32 * wn + lc];
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                    Hl[row * LDT + 32 * wn + lc] = enc<BF>(fmaxf(acc[g] + bias, 0.0f));
                }
                __syncthreads();
            }
        }
        PH(2);
        if (EARLY_NEXT && L >= 1) tile_load<(VEC != 0)>(pn, a.x, next_row0, a.rows, a.x_stride, 0, min(KC, din));
        // ================= head forward: 16x16x4 MFMA, wave w owns rows 16w..16w+15 (= the rows of its quad lanes) ====
        float* HL = smem + lds.Hs(L);
        if constexpr (WP44) {
            head_logits44<BF>(ls + 16 * wave * lstride, HL + 16 * wave * LDT, wouts, smem + lds.bout);
        } else {
            const int n = lane & 15, g4 = lane >> 4;
#pragma unroll
            for (int ct = 0; ct < WR / 16; ++ct) {
                const f32x4 lg = head_logits_mfma<BF>(HL + 16 * wave * LDT, wouts + 16 * ct * WLD);
                const int k = 16 * ct + n;
                if (k < KP) {
                    const float bias = smem[lds.bout + k];
#pragma unroll
                    for (int q = 0; q < 4; ++q) ls[(16 * wave + 4 * g4 + q) * lstride + k] = lg[q] + bias;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();  // same-wave LDS hand-off (DS ops of one wave execute in order)
        float zreg[KJ];
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            zreg[j] = -1e9f;
            if (4 * j + hq < dout && ri.avb[j]) zreg[j] = ls[hrow * lstride + 4 * j + hq];  // masked_fill(~avail, -1e9)
        }
        if (MODE == M_FWD) {
#pragma unroll
            for (int j = 0; j < KJ; ++j)
                if (rvalid && 4 * j + hq < dout) a.y[(long)grow * dout + 4 * j + hq] = zreg[j];
        }
        // K > 8 heads (KJ == 8) sample on the four lanes of the row, straight from registers; the one-lane samplers (K <= 8: bit-identical to
        // the fused rollouts; COMA's mixture; greedy) read the masked logits back from LDS
        const bool quad_sampler = (MODE == M_ACT) && KJ == 8 && a.act_eps == 0.0f;
        if (MODE == M_ACT && !quad_sampler) {
#pragma unroll
            for (int j = 0; j < KJ; ++j)
                if (4 * j + hq < dout) ls[hrow * lstride + 4 * j + hq] = zreg[j];
        }
        if (MODE == M_FWD) continue;  // next tile (the loop-top barrier protects LDS reuse)
        if (MODE == M_ACT && !quad_sampler) __syncthreads();
        PH(3);

        // ================= per-row head math =================
        if (MODE == M_ACT) {
            const int seq = a.t_decode > 0 ? grow / a.t_decode : grow;
            const int tt = a.t_decode > 0 ? grow - seq * a.t_decode : a.t;
            const unsigned long long gr = (unsigned long long)(a.row_offset + seq);
            if (quad_sampler) {
                if constexpr (KJ == 8) {
                    // every lane of the quad draws the row's uniform (same instruction count as one lane of four)
                    const cm_u4 rnd = cm_philox4x32((uint32_t)gr, (uint32_t)(gr >> 32), (uint32_t)tt, CM_STREAM_ACT,
                                                    (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
                    int chosen; float lp;
                    quad_categorical_sample<KJ>(zreg, dout, hq, cm_u01(rnd.x), &chosen, &lp);
                    if (hq == 0 && rvalid) {
                        a.action_out[(long)grow * a.out_stride] = chosen;
                        a.logp_out[(long)grow * a.out_stride] = lp;
                    }
                }
            } else if (hq == 0 && rvalid) {
                const cm_u4 rnd = cm_philox4x32((uint32_t)gr, (uint32_t)(gr >> 32), (uint32_t)tt, CM_STREAM_ACT,
                                                (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
                int chosen; float lp;
                if (a.act_eps < 0.0f) cm_categorical_greedy(ls + hrow * lstride, dout, &chosen, &lp);
                else if (a.act_eps > 0.0f) cm_categorical_sample_eps(ls + hrow * lstride, dout, cm_u01(rnd.x), a.act_eps, &chosen, &lp);
                else cm_categorical_sample(ls + hrow * lstride, dout, cm_u01(rnd.x), &chosen, &lp);
                a.action_out[(long)grow * a.out_stride] = chosen;
                a.logp_out[(long)grow * a.out_stride] = lp;
            }
            PH(4);
            continue;
        }

        if (TRAIN) {
            const bool valid = rvalid && (ri.t < ri.eplen);
            const float invA = 1.0f / (float)a.A;
            if (MODE == M_ACTOR) {
                // Categorical(logits) statistics with the row's K logits spread over its 4 lanes (k = 4j + hq)
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < KJ; ++j) if (4 * j + hq < dout) m = fmaxf(m, zreg[j]);
                m = quad_max(m);
                float s = 0.0f;
                float pj[KJ];
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    pj[j] = 0.0f;
                    if (4 * j + hq < dout) { pj[j] = head_exp(zreg[j] - m); s += pj[j]; }
                }
                s = quad_sum(s);
                const float lse = m + head_log(s);
                const float rs = head_rcp(s);
                float ent = 0.0f, lpa = 0.0f;
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    if (4 * j + hq < dout) {
                        const float lp = zreg[j] - lse;
                        const float p = pj[j] * rs;  // softmax probability (exp(z - m) / sum)
                        pj[j] = p;
                        ent -= p * lp;
                        if (4 * j + hq == ri.act) lpa = lp;
                    }
                }
                ent = quad_sum(ent);
                lpa = quad_sum(lpa);
                const float log_ratio = lpa - ri.lpo;
                const float ratio = head_exp(log_ratio);
                const float advv = ri.adv;
                const float pg1 = advv * ratio;
                const float pg2 = advv * fminf(fmaxf(ratio, a.clip_lo), a.clip_hi);
                const bool inr = (ratio >= a.clip_lo) && (ratio <= a.clip_hi);
                // d min(pg1,pg2)/d ratio with torch's tie rule (grad/2 to each operand)
                float g;
                if (pg1 < pg2) g = advv;
                else if (pg1 > pg2) g = inr ? advv : 0.0f;
                else g = 0.5f * advv + (inr ? 0.5f * advv : 0.0f);
                if (valid && hq == 0) {
                    st_pg += invA * fminf(pg1, pg2);
                    st_ent += invA * ent;
                    st_kl += invA * ((ratio - 1.0f) - log_ratio);
                    st_clip += (fabsf(ratio - 1.0f) > a.clip_eps) ? invA : 0.0f;
                    if (ri.ag == 0) st_cnt += 1.0f;
                }
                const float gr = g * ratio;
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    const int k = 4 * j + hq;
                    if (k < dout) {
                        const float lp = zreg[j] - lse;
                        float d = invA * (-gr * ((k == ri.act ? 1.0f : 0.0f) - pj[j]) + a.ent_coef * pj[j] * (lp + ent));
                        if (!valid || zreg[j] <= -5e8f) d = 0.0f;  // padded rows; masked_fill blocks the gradient
                        ls[hrow * lstride + k] = d;
                        if constexpr (WP) dboW[j] += d;  // head bias gradient straight from the register (no LDS read-back on the wave's critical path)
                    }
                }
            } else if (MODE == M_COMA_ACTOR) {
                // pi = softmax(masked logits) (eps = 0 in the update, coma_multienvs.py:650); log_pi = log(pi + 1e-8);
                // row loss = -log_pi[a] * adv - c * (-(pi * log_pi).mean_k); rows of all agents are SUMMED (no agent-mean)
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < KJ; ++j) if (4 * j + hq < dout) m = fmaxf(m, zreg[j]);
                m = quad_max(m);
                float s = 0.0f;
                float pj[KJ], lq[KJ];
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    pj[j] = 0.0f;
                    if (4 * j + hq < dout) { pj[j] = expf(zreg[j] - m); s += pj[j]; }
                }
                s = quad_sum(s);
                const float rs = 1.0f / s;
                const float invK = 1.0f / (float)dout;
                float ent = 0.0f, lpa = 0.0f, pa = 0.0f;
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    lq[j] = 0.0f;
                    if (4 * j + hq < dout) {
                        const float p = pj[j] * rs;
                        pj[j] = p;
                        lq[j] = logf(p + 1e-8f);
                        ent -= p * lq[j];
                        if (4 * j + hq == ri.act) { lpa = lq[j]; pa = p; }
                    }
                }
                ent = quad_sum(ent) * invK;
                lpa = quad_sum(lpa);
                pa = quad_sum(pa);
                const float advv = ri.adv;
                if (valid && hq == 0) {
                    st_pg += lpa * advv;
                    st_ent += ent;
                    if (ri.ag == 0) st_cnt += 1.0f;
                }
                // dL/dpi_k, then through the softmax: dz_j = pi_j * (g_j - sum_k g_k pi_k)
                float gj[KJ], gbar = 0.0f;
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    gj[j] = 0.0f;
                    const int k = 4 * j + hq;
                    if (k < dout) {
                        gj[j] = a.ent_coef * invK * (lq[j] + pj[j] / (pj[j] + 1e-8f));
                        if (k == ri.act) gj[j] -= advv / (pa + 1e-8f);
                        gbar += gj[j] * pj[j];
                    }
                }
                gbar = quad_sum(gbar);
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    const int k = 4 * j + hq;
                    if (k < dout) {
                        float d = pj[j] * (gj[j] - gbar);
                        if (!valid || zreg[j] <= -5e8f) d = 0.0f;
                        ls[hrow * lstride + k] = d;
                    }
                }
            } else if (MODE == M_QCRITIC) {
                // Q of the taken action vs the target (F.mse_loss over alive envs x agents, x n_alive: agent-MEAN, env-SUM)
                float qa = 0.0f;
#pragma unroll
                for (int j = 0; j < KJ; ++j) if (4 * j + hq == ri.act) qa = zreg[j];
                qa = quad_sum(qa);
                const float df = qa - ri.ret;
                if (valid && hq == 0) {
                    st_vl += invA * df * df;
                    if (ri.ag == 0) st_cnt += 1.0f;
                }
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    const int k = 4 * j + hq;
                    if (k < dout) ls[hrow * lstride + k] = (valid && k == ri.act) ? 2.0f * invA * df : 0.0f;
                }
            } else if (hq == 0) {  // M_CRITIC: one output per row, owned by lane hq == 0
                float d = 0.0f;
                if (valid) {
                    const float v = zreg[0];
                    if (a.per_agent) {
                        const float df = v - ri.ret;
                        st_vl += invA * df * df;
                        d = 2.0f * invA * df;
                        if (ri.ag == 0) st_cnt += 1.0f;
                    } else {
                        float sd = 0.0f, sq = 0.0f;
                        for (int q = 0; q < a.A; ++q) {
                            const float df = v - a.ret[((long)ri.e * a.A + q) * a.T + ri.t];
                            sd += df; sq += df * df;
                        }
                        st_vl += invA * sq;
                        d = 2.0f * invA * sd;
                        st_cnt += 1.0f;
                    }
                }
                ls[hrow * lstride] = d;
            }
            if constexpr (WP) {
            // head bias gradient from the dlogits this lane just wrote (k = 4 j + hq; M_CRITIC: only lane hq == 0 wrote, slot 0)
            __builtin_amdgcn_wave_barrier();
            if constexpr (MODE != M_ACTOR) {
#pragma unroll
            for (int j = 0; j < KJ; ++j)
                if (4 * j + hq < dout) dboW[j] += ls[hrow * lstride + 4 * j + hq];
            }
            PH(4);
            // dWout, dZ_L and the in-place relu' write for this wave's 16 rows (same-wave LDS hand-off: DS ops of a wave execute in order)
            if constexpr (WP44) head_bwd_wave44<BF>(accWoW[0], ls + 16 * wave * KP, HL + 16 * wave * LDT, wouts);
            else head_bwd_wave<KP, (WP ? WR / 16 : 1), BF>(accWoW, ls + 16 * wave * KP, HL + 16 * wave * LDT, wouts);
            PH(5);
            __syncthreads();
            PH(6);
            } else {
            __syncthreads();
            PH(4);
            // ---- dWout, dbout (contraction over the tile's rows; reads HL before it is overwritten)
            // wave (wm, wn): rows 32wm..32wm+31 of the tile, hidden columns 32wn..32wn+31, all 32 (padded) head rows
#pragma unroll
            for (int q = 0; q < WR / 16; ++q) colred_head16<KP, BF>(accWo[q], ls, 16 * q, HL + 16 * wave);
            {
                const int k = tid & 31, part = tid >> 5;
                float sb = 0.0f;
                if (k < KP) {
#pragma unroll
                    for (int r = 0; r < 8; ++r) sb += ls[(part * 8 + r) * KP + k];
                }
                dbo += sb;
            }
            // ---- dZ_L = (dlogits * Wout) .* relu'(H_L): MFMA into registers now, in-place write after the barrier
#pragma unroll
            for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
            head_bwd_mfma<KP>(acc, ls + 32 * wm * KP, wouts + 32 * wn);
            __syncthreads();
            PH(5);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                float* p = HL + row * LDT + 32 * wn + lc;
                *p = enc<BF>((dec<BF>(*p) > 0.0f) ? acc[g] : 0.0f);
            }
            __syncthreads();
            PH(6);
            }
            // ================= backward through hidden layers =================
#pragma unroll
            for (int l = LCAP; l >= 1; --l) {
                if (l <= L) {
                    float* Zl = smem + lds.Hs(l);       // holds dZ_l
                    float* Hm = smem + lds.Hs(l - 1);   // holds H_{l-1}
                    if (!ws_resident) {
                        stage_rows<BF>(Ws, a.params + off.Wl(l - 1), 0, H, H, 0, H);
                        __syncthreads();
                    }
                    {   // bias gradient: column sums
                        const int c = tid & 63, part = tid >> 6;
                        float s = 0.0f;
#pragma unroll
                        for (int r = 0; r < TM / 4; ++r) s += dec<BF>(Zl[(part * (TM / 4) + r) * LDT + c]);
                        dbh[l] += s;
                    }
                    if (BF) colred_bf<HAND>(accWl[l - 1], Zl + 32 * wm, Hm + 32 * wn);
                    else colred_sel<HAND>(accWl[l - 1], Zl + 32 * wm, Hm + 32 * wn);
#pragma unroll
                    for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
                    if (BF) rowpar_tn_bf<HAND>(acc, Zl + 32 * wm * LDT, Ws + 32 * wn);
                    else rowpar_tn_sel<HAND>(acc, Zl + 32 * wm * LDT, Ws + 32 * wn);
                    __syncthreads();
                    PH(7);
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                        float* p = Hm + row * LDT + 32 * wn + lc;
                        *p = enc<BF>((dec<BF>(*p) > 0.0f) ? acc[g] : 0.0f);
                    }
                    if (XMERGE && l == 1) {
                        // the tile's X goes back into its buffer HERE (single-chunk inputs): Xs held dZ_1, whose last readers (this
                        // layer's two products) finished at the barrier above -- the separate barrier pair of the layer-0 weight gradient
                        // (wait for those readers, store, publish) folds into the one that publishes dZ_0: two barriers fewer per tile
                        tile_store<(VEC != 0), BF>(Xs, px);
                        px = pn;  // the next tile's X, in flight since the head phase (EARLY_NEXT)
                    }
                    __syncthreads();
                    PH(8);
                }
            }
            // ================= layer 0 backward: bias + dW0 chunks =================
            {
                float* Z0 = smem + lds.Hs(0);
                {
                    const int c = tid & 63, part = tid >> 6;
                    float s = 0.0f;
#pragma unroll
                    for (int r = 0; r < TM / 4; ++r) s += dec<BF>(Z0[(part * (TM / 4) + r) * LDT + c]);
                    dbh[0] += s;
                }
                if (NCH == 0 || (MODE == M_QCRITIC && a.dz0 != nullptr)) {  // external layer-0 weight gradient: hand dZ0 to k_dw0_stream (coalesced 16-byte stores)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int idx = tid + NTHREADS * i;
                        const int r = idx >> 4, c4 = (idx & 15) * 4;
                        if (row0 + r < a.rows)
                        {
                            if (BF) {
                                const float4 v = *reinterpret_cast<const float4*>(Z0 + r * LDT + c4);
                                *reinterpret_cast<float4*>(a.dz0 + (row0 + r) * HP + c4) = make_float4(dec<true>(v.x), dec<true>(v.y), dec<true>(v.z), dec<true>(v.w));
                            } else {
                                *reinterpret_cast<float4*>(a.dz0 + (row0 + r) * HP + c4) = *reinterpret_cast<const float4*>(Z0 + r * LDT + c4);
                            }
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < (NCH > 0 ? NCH : 0); ++c) {
                    if (NCH > 1 || (L >= 1 && !XMERGE)) {  // X chunks are re-streamed (L2 / MALL hits) through the same prefetch registers
                        __syncthreads();
                        tile_store<(VEC != 0), BF>(Xs, px);
                        const bool last = (c + 1 == NCH);
                        const int cn = last ? 0 : c + 1;
                        if (EARLY_NEXT) px = pn;  // already in flight since the head phase
                        else tile_load<(VEC != 0)>(px, a.x, last ? next_row0 : row0, a.rows, a.x_stride, cn * KC, min(KC, din - cn * KC));
                        if (last && !w0_resident) tile_load<(VEC == 1)>(pw, W0g, 0, H, w0ld, 0, min(KC, din));
                        __syncthreads();
                    }
                    if (BF) colred_bf<HAND>(accW0[c], Z0 + 32 * wm, Xs + 32 * wn);
                    else colred_sel<HAND>(accW0[c], Z0 + 32 * wm, Xs + 32 * wn);
                }
            }
            PH(9);
        }
    }
    PH_FLUSH;
#undef CM_FETCH_ROW_INPUTS
    if (clk_on && tid == 0) {  // the tile loop of this workgroup: what the launch's duration is made of (the partial write below is ~1 % of it)
        unsigned long long* c = a.clk + 4 * (size_t)blockIdx.x;
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11)), hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));  // HW_REG_XCC_ID, HW_REG_HW_ID
        c[0] = __builtin_amdgcn_s_memtime() - clk_c0; c[1] = ((unsigned long long)xcc << 32) | hw; c[2] = clk_r0; c[3] = __builtin_amdgcn_s_memrealtime();
    }

    // ================= write this workgroup's partial gradient + stats =================
    if (TRAIN) {
        float* out = a.partial + (size_t)blockIdx.x * a.PS;
        // dW0: wave (wm, wn) holds rows n = 32wm + i, cols k = 64c + 32wn + j
#pragma unroll
        for (int c = 0; c < (NCH > 0 ? NCH : 0); ++c) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int n = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                const int k = KC * c + 32 * wn + lc;
                if (n < H && k < din) out[off.W0 + n * din + k] = accW0[c][g];
            }
        }
#pragma unroll
        for (int l = 0; l < LCAP; ++l) {
            if (l < L) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int n = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                    const int k = 32 * wn + lc;
                    if (n < H && k < H) out[off.Wl(l) + n * H + k] = accWl[l][g];
                }
            }
        }
        if constexpr (WP) {   // dWout: every wave holds a [WR x 64] partial over ITS rows -- lane (n, g): k = 16q + 4g + r, column 16ct + n; summed over the
            // four waves in wave order through LDS (the tile buffers are dead), one 16-row block of k at a time
            const int n = lane & 15, g4 = lane >> 4;
            float* wsum = smem + lds.Xs;  // [4 waves][16 k][64 c] = 16 KB of the (dead) X | W0 tile buffers
            if constexpr (WP44) {   // eight partials (wave, row half) of [8 k][64 c]: lane 4 b + x, b = 8 rh + 4 kg + cg, register i of accWoW[0][m] = k 4 kg + i, column 16 cg + 4 x + m
                const int x = lane & 3, b = lane >> 2, cg = b & 3, kg = (b >> 2) & 1, rh = b >> 3;
                __syncthreads();
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float4*>(wsum + ((2 * wave + rh) * 8 + 4 * kg + i) * HP + 16 * cg + 4 * x) =
                        make_float4(accWoW[0][0][i], accWoW[0][1][i], accWoW[0][2][i], accWoW[0][3][i]);
                __syncthreads();
                for (int i = tid; i < 8 * HP; i += NTHREADS) {
                    const int k = i / HP, c = i % HP;
                    float sw = wsum[i];
#pragma unroll
                    for (int part = 1; part < 8; ++part) sw += wsum[part * 8 * HP + i];
                    if (k < dout && c < H) out[off.Wout + k * H + c] = sw;
                }
            } else
#pragma unroll
            for (int q = 0; q < WR / 16; ++q) {
                __syncthreads();
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) wsum[(wave * 16 + 4 * g4 + r) * HP + 16 * ct + n] = accWoW[q][ct][r];
                __syncthreads();
                for (int i = tid; i < 16 * HP; i += NTHREADS) {
                    const int k = 16 * q + i / HP, c = i % HP;
                    if (k < dout && c < H) out[off.Wout + k * H + c] = ((wsum[i] + wsum[16 * HP + i]) + wsum[2 * 16 * HP + i]) + wsum[3 * 16 * HP + i];
                }
            }
            // dbout: lanes with the same hq over the wave's 16 rows (xor 4 .. 32), then the four waves
            __syncthreads();
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                float v = dboW[j];
                v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
                if (lane < 4) red[wave * KMAX + 4 * j + lane] = v;
            }
            __syncthreads();
            if (tid < dout) out[off.bout + tid] = ((red[tid] + red[KMAX + tid]) + red[2 * KMAX + tid]) + red[3 * KMAX + tid];
        } else
        {   // dWout: lane (n = lane & 15, g = lane >> 4) of wave w holds k = 16q + 4g + r (r = 0..3), column 16w + n
            const int n = lane & 15, g4 = lane >> 4;
#pragma unroll
            for (int q = 0; q < WR / 16; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * q + 4 * g4 + r, c = 16 * wave + n;
                    if (k < dout && c < H) out[off.Wout + k * H + c] = accWo[q][r];
                }
            __syncthreads();
            red[tid] = dbo;  // [8 parts][32 k]
            __syncthreads();
            if (tid < dout) {
                float sb = 0.0f;
#pragma unroll
                for (int q = 0; q < 8; ++q) sb += red[q * 32 + tid];
                out[off.bout + tid] = sb;
            }
        }
        // bias grads: 4 row-parts per column -> LDS -> sum
#pragma unroll
        for (int l = 0; l <= LCAP; ++l) {
            if (l <= L) {
                __syncthreads();
                red[(tid >> 6) * HP + (tid & 63)] = dbh[l];
                __syncthreads();
                if (tid < H) {
                    const float s = red[tid] + red[HP + tid] + red[2 * HP + tid] + red[3 * HP + tid];
                    out[(l == 0 ? off.b0 : off.bl(l - 1)) + tid] = s;
                }
            }
        }
        // stats
        float sv[6] = {st_pg, st_ent, st_kl, st_clip, st_vl, st_cnt};
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            const float v = cm_wave_sum(sv[s]);
            if (lane == 0) red[s * 4 + wave] = v;
        }
        __syncthreads();
        if (tid < CM_NUM_STATS) {
            float v = 0.0f;
            if (tid < 6) v = red[tid * 4] + red[tid * 4 + 1] + red[tid * 4 + 2] + red[tid * 4 + 3];
            out[off.P + tid] = v;
        }
    }
}

// sum per-workgroup partials: out[i] = sum_w partial[w][i] for i in the window [i0, n).
// 1024 threads = 64 columns x 16 row groups: every group streams its share of the partial rows with 256-byte
// coalesced reads, the 16 partial sums meet in LDS in a fixed order (deterministic).
constexpr int RED_COLS = 64, RED_GROUPS = 16;
__global__ __launch_bounds__(RED_COLS * RED_GROUPS) void k_reduce_partials(const float* __restrict__ partial, int nparts, int PS,
                                                                           int i0, int n, float* __restrict__ out) {
    __shared__ float sh[RED_GROUPS][RED_COLS];
    const int c = threadIdx.x & (RED_COLS - 1), g = threadIdx.x / RED_COLS;
    const int i = i0 + blockIdx.x * RED_COLS + c;
    float s0 = 0.f, s1 = 0.f;
    if (i < n) {
        int w = g;
        for (; w + RED_GROUPS < nparts; w += 2 * RED_GROUPS) {
            s0 += partial[(size_t)w * PS + i];
            s1 += partial[(size_t)(w + RED_GROUPS) * PS + i];
        }
        if (w < nparts) s0 += partial[(size_t)w * PS + i];
    }
    sh[g][c] = s0 + s1;
    __syncthreads();
    if (g == 0 && i < n) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < RED_GROUPS; ++q) t += sh[q][c];
        out[i] = t;
    }
}

constexpr int MAX_GRID = 512;  // persistent workgroups: up to two per CU

inline int check_shapes(const char* who, int din, int H, int L, int dout) {
    CM_REQUIRE(din > 0 && H > 0 && L >= 0 && dout > 0, "%s: bad dims din=%d H=%d L=%d dout=%d", who, din, H, L, dout);
    CM_REQUIRE(H <= HP, "%s: hidden_dim=%d > %d is not supported by this build", who, H, HP);
    CM_REQUIRE(L <= LMAX, "%s: num_layers=%d > %d is not supported by this build", who, L, LMAX);
    CM_REQUIRE(dout <= KMAX, "%s: output width %d > %d is not supported by this build", who, dout, KMAX);
    return 0;
}
inline int check_rows(const char* who, long rows) {
    CM_REQUIRE(rows < (1L << 31), "%s: %ld rows exceed the 2^31 row limit of one launch", who, rows);
    return 0;
}

// Unequal static split of a full persistent grid (two workgroups per CU).  Measured with cm_clock_probe on the actor pass of config 3
// (tools/probes/wg_span.py, round 5): the 512 workgroups finish in TWO groups -- the first 256 (dispatched first, one per CU: the older
// waves of every SIMD, which the issue arbiter favours) after 1.45 ms, the second 256 after 1.67 ms, the last 0.2 ms with one workgroup per
// CU.  Handing the favoured half `pct` % of the tiles (a function of the launch's shape only: results stay run-to-run identical) lets
// both finish together: 1.713 -> 1.691 ms at 55 - 57 % (gpurun_out/r05e; less than the 0.2 ms tail suggests -- a workgroup alone on
// its CU runs its tiles 1.7 x faster than beside a partner, so the tail was mostly useful work).  option "tile_split": auto | 50 (equal) | 52 .. 60.
inline void set_tile_split(MlpArgs& a, int grid, int nch, bool alone = false) {
    a.split_tiles = 0;
    const long nt = (a.rows + TM - 1) / TM;
    if (wgs_per_cu(nch) != 2 || grid != 512 || nt < 4L * grid) return;  // only a full two-per-CU grid with >= 4 tiles per workgroup
    const int opt = cm_option(CM_OPTION_TILE_SPLIT);
    // auto: only launches that have the GPU to themselves (>= 2^21 rows: learner.overlap_critic's one-stream schedule) -- beside the critic's
    // kernels the favoured half is not the first half of the grid, and a split costs 15 - 25 % (512-env share: 0.28 -> 0.33 - 0.35 ms)
    // (alone: launches that are ordered behind everything else by construction -- the value pass waits for the critic stream first)
    const int pct = opt ? opt : ((alone || a.rows >= (1L << 21)) ? CM_TILE_SPLIT_DEFAULT : 50);
    if (pct <= 50) return;
    a.split_tiles = (nt * pct + 50) / 100;
}

inline int grid_for(long rows, int nch = 0) {
    const long nt = (rows + TM - 1) / TM;
    const long cap = 256L * wgs_per_cu(nch);
    return (int)(nt < cap ? nt : cap);
}

// 16-byte tile loads need 16-byte aligned rows of the input (leading dimension a multiple of 4 floats, >= din rounded up: the quad that
// straddles din reads zero padding) AND of W0 wherever W0 is tile-loaded, i.e. streamed: its own rows when din % 4 == 0, else the
// padded image of prep_w0_image; a single-chunk input (din <= 64) keeps W0 in LDS and never tile-loads it.
__host__ __device__ inline bool x_rows_vec(const MlpArgs& a) {
    return (a.x_stride % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.x) & 15) == 0) && a.x_stride >= (a.din + 3) / 4 * 4;
}
inline bool can_vec(const MlpArgs& a) {
    const bool w_ok = (a.din % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.params) & 15) == 0);
    return x_rows_vec(a) && (w_ok || (a.din + KC - 1) / KC == 1 || a.w0p != nullptr);
}
inline size_t w0_image_floats(int din, int H) { return (din % 4 != 0 && din > KC) ? (size_t)H * ((din + 3) / 4 * 4) : 0; }
__global__ void k_pad_w0(const float* __restrict__ w0, int H, int din, int ld, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < H * ld) { const int n = i / ld, k = i - n * ld; out[i] = (k < din) ? w0[n * din + k] : 0.0f; }
}
// builds the image into `scratch` (>= w0_image_floats floats, 16-byte aligned) when it pays; otherwise leaves a.w0p = NULL
inline void prep_w0_image(MlpArgs& a, float* scratch, size_t scratch_floats, hipStream_t s) {
    a.w0p = nullptr; a.w0_ld = 0;
    const size_t need = w0_image_floats(a.din, a.H);
    if (!need || !x_rows_vec(a) || !scratch || scratch_floats < need || (reinterpret_cast<uintptr_t>(scratch) & 15)) return;
    const int ld = (a.din + 3) / 4 * 4;
    const Offsets off = make_offsets(a.din, a.H, a.L, a.dout);
    hipLaunchKernelGGL(k_pad_w0, dim3((a.H * ld + 255) / 256), dim3(256), 0, s, a.params + off.W0, a.H, a.din, ld, scratch);
    a.w0p = scratch; a.w0_ld = ld;
}

template <int NCH, int MODE, int VEC, int LCAP, int KJ, bool BF = false, bool HAND = false>
inline void launch_one(const MlpArgs& a, int grid, size_t lds_bytes, hipStream_t s) {
#ifdef CM_PHASE_PROF
    // profiling build only (tools/phase_prof.py): CM_PROF_ONE_WG=1 pads LDS so that ONE workgroup fits a CU and halves the grid --
    // timing experiment for the occupancy argument of docs/KERNEL_NOTES.md section 10 (results of such a launch are not meaningful)
    if (const char* e = getenv("CM_PROF_ONE_WG")) {
        if (e[0] == '1') { if (lds_bytes < 100 * 1024) lds_bytes = 100 * 1024; if (grid > 256) grid = 256; }
    }
#endif
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp<NCH, MODE, VEC, LCAP, KJ, BF, HAND>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((k_mlp<NCH, MODE, VEC, LCAP, KJ, BF, HAND>), dim3(grid), dim3(NTHREADS), lds_bytes, s, a);
}

// option mfma = bf16x3 / bf16 opts the PPO training passes into the compensated / single-pass bf16 GEMM loops (default: exact fp32 MFMA)
inline bool mfma_bf16x3() { return cm_mfma_mode() != 0; }   // any bf16 mode: the split-word kernels
inline bool mfma_bf16_one() { return cm_mfma_mode() == 2; } // single pass

// runtime (vec, L <= 1, dout <= 8) -> compile-time (VEC, LCAP, KJ)
template <int NCH, int MODE>
inline void launch_variant(const MlpArgs& a, int grid, size_t lds_bytes, hipStream_t s) {
    const bool vec = can_vec(a), l1 = a.L <= 1, k8 = a.dout <= 8;
    if constexpr ((MODE == M_ACTOR || MODE == M_CRITIC) && NCH <= 2) {
        if (vec && l1 && k8 && mfma_bf16x3()) {
            if (mfma_bf16_one()) launch_one<NCH, MODE, 1, 1, 2, true, true>(a, grid, lds_bytes, s);
            else launch_one<NCH, MODE, 1, 1, 2, true>(a, grid, lds_bytes, s);
            return;
        }
    }
    // hand-ordered LDS reads (see rowpar_nt): the single-chunk actor pass of a batch too large to share the GPU with the critic's
    // epochs (learner.overlap_critic's 2^21-row limit), and every forward pass (the value pass follows the rollout and the join with
    // the critic stream: nothing runs beside it).  cm_set_option("mlp_forms", "hand"|"loop") forces one form wherever both are compiled (A/B runs, tests).
    if constexpr (((MODE == M_ACTOR || MODE == M_COMA_ACTOR || MODE == M_QCRITIC) && NCH == 1) || (MODE == M_FWD && NCH <= 0)) {
        const int f = cm_option(CM_OPTION_MLP_FORMS);  // 0 auto, 1 hand, 2 loop
        const bool big = MODE == M_FWD || a.rows >= (1L << 21);  // >=: the learner's one-stream schedule starts AT 2^21 rows (learner.overlap_critic) -- nothing runs beside the pass there
        const bool hand = f ? (f == 1) : big;
        if (vec && l1 && k8 && hand) { launch_one<NCH, MODE, 1, 1, 2, false, true>(a, grid, lds_bytes, s); return; }
    }
    if (vec) {
        if (l1) { if (k8) launch_one<NCH, MODE, 1, 1, 2>(a, grid, lds_bytes, s); else launch_one<NCH, MODE, 1, 1, 8>(a, grid, lds_bytes, s); }
        else    { if (k8) launch_one<NCH, MODE, 1, 2, 2>(a, grid, lds_bytes, s); else launch_one<NCH, MODE, 1, 2, 8>(a, grid, lds_bytes, s); }
    } else {
        if (l1) { if (k8) launch_one<NCH, MODE, 0, 1, 2>(a, grid, lds_bytes, s); else launch_one<NCH, MODE, 0, 1, 8>(a, grid, lds_bytes, s); }
        else    { if (k8) launch_one<NCH, MODE, 0, 2, 2>(a, grid, lds_bytes, s); else launch_one<NCH, MODE, 0, 2, 8>(a, grid, lds_bytes, s); }
    }
}

template <int MODE>
inline int launch_infer(const MlpArgs& a, int grid, size_t lds_bytes, hipStream_t s) {
    // exactly two input chunks: W0 in registers (155 .. 242 registers per lane, two workgroups per CU either way)
    if ((a.din + KC - 1) / KC == 2) launch_variant<-2, MODE>(a, grid, lds_bytes, s);
    else launch_variant<0, MODE>(a, grid, lds_bytes, s);
    return 0;
}

template <int MODE>
inline int launch_train(const MlpArgs& a, int grid, size_t lds_bytes, hipStream_t s) {
    const int nch = (a.din + KC - 1) / KC;
    switch (nch) {
        case 1: launch_variant<1, MODE>(a, grid, lds_bytes, s); break;
        case 2: launch_variant<2, MODE>(a, grid, lds_bytes, s); break;
        case 3: launch_variant<3, MODE>(a, grid, lds_bytes, s); break;
        case 4: launch_variant<4, MODE>(a, grid, lds_bytes, s); break;
        case 5: launch_variant<5, MODE>(a, grid, lds_bytes, s); break;
        case 6: launch_variant<6, MODE>(a, grid, lds_bytes, s); break;
        case 7: launch_variant<7, MODE>(a, grid, lds_bytes, s); break;
        case 8: launch_variant<8, MODE>(a, grid, lds_bytes, s); break;
        default: CM_FAIL(-1, "input width %d > %d is not supported by the fused training kernels", a.din, 8 * KC);
    }
    return 0;
}

// COMA modes: fused kernels for din <= 128 only (wider inputs take the split schedule of cm_mlp_split.h)
template <int MODE>
inline int launch_train_small(const MlpArgs& a, int grid, size_t lds_bytes, hipStream_t s) {
    const int nch = (a.din + KC - 1) / KC;
    switch (nch) {
        case 1: launch_variant<1, MODE>(a, grid, lds_bytes, s); break;
        case 2: launch_variant<2, MODE>(a, grid, lds_bytes, s); break;
        default: CM_FAIL(-1, "input width %d > %d needs the split schedule", a.din, 2 * KC);
    }
    return 0;
}

}  // namespace
