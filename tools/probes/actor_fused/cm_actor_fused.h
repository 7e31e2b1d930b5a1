// cm_actor_fused.h -- PPO actor forward + backward on the design of cm_critic_fused.h (included by cm_mlp_actor.hip).
//
// cleanmarl/mappo_multienvs.py:527-551, :561-582 (clipped-surrogate loss body, backward) for a single-chunk observation (Do <= 64),
// one hidden->hidden layer, H <= 64, K <= 16 actions.  Wave w owns hidden columns 16w .. 16w+15 of both layers; W0, W1 (both
// orientations) and the zero-padded head (both orientations) are MFMA B operands in REGISTERS, as are the weight-gradient accumulators
// of the wave's slices, for the whole launch.  Activations travel through LDS tiles; every LDS read that feeds an MFMA is issued by
// hand ahead of its use (cm_common.h) -- straight-line pipelines, checked by tools/lint_lds_hazards.py.
//   * head without cross-wave sums: wave w computes the COMPLETE 16 x 16 logit tile of rows 16w .. 16w+15 (A = its rows of H1 from
//     LDS, B = Wout in registers); the softmax / PPO math runs with lane n = action n, four rows per lane, DPP row reductions;
//   * dWout from the wave's OWN dlogits registers as the A operand (contraction slot g <-> row 4g + t), B = the same rows of H1 read
//     16 bytes at a time: a [16 actions x 64 hidden] partial per wave, folded over the four waves once per launch;
//   * dH1 = dlogits Wout on a 64 x 20 dlogits tile in LDS; the rest (dW1, dH0, dW0, bias sums) as in cm_critic_fused.h.
// 75 KB of LDS and <= 256 registers per lane: TWO workgroups per CU, whose barrier / loss-math phases overlap the other's products.
// Same un-normalised sums and statistic slots as k_mlp<.., M_ACTOR>; summation order differs (tolerance-tested, 1e-4).
#pragma once
#include "cm_critic_fused.h"

namespace {

constexpr int DLS = 20;  // row stride of the dlogits tile: 16-byte reads down 16 rows touch all 64 banks

__device__ __forceinline__ float af_row16_max(float v) {
#define CM_ROR(ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    v = fmaxf(v, CM_ROR(0x128)); v = fmaxf(v, CM_ROR(0x124)); v = fmaxf(v, CM_ROR(0x122)); v = fmaxf(v, CM_ROR(0x121));
#undef CM_ROR
    return v;
}

__global__ __launch_bounds__(NTHREADS, 2) void k_actor_fused(const MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* XS = smem;                       // [TM][LDT] the tile's observations
    float* H0s = XS + TM * LDT;             // [TM][LDT] h0 (forward / dW1), then dZ0
    float* H1s = H0s + TM * LDT;            // [TM][LDT] h1 (head, dWout)
    float* DZ1 = H1s + TM * LDT;            // [TM][LDT]
    float* DL = DZ1 + TM * LDT;             // [TM][DLS] dLoss/dlogits
    float* red = DL + TM * DLS;             // 2 * NTHREADS
    const Offsets off = make_offsets(a.din, a.H, 1, a.dout);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const int c0 = 16 * wave, col = c0 + n, H = a.H, din = a.din, K = a.dout;
    const bool cok = col < H, kok = n < K;
    // ---- weights -> registers
    float w0[16], w1n[16], w1t[16], won[16], wot[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 16 * j + 4 * g + i;
            w0[4 * j + i] = (cok && k < din) ? a.params[off.W0 + (long)col * din + k] : 0.0f;
            w1n[4 * j + i] = (cok && k < H) ? a.params[off.Wl(0) + col * H + k] : 0.0f;   // nt image: W1[col][k]
            w1t[4 * j + i] = (cok && k < H) ? a.params[off.Wl(0) + k * H + col] : 0.0f;   // tn image: W1[k][col]
            won[4 * j + i] = (kok && k < H) ? a.params[off.Wout + n * H + k] : 0.0f;      // logits: Wout[action n][hidden k]
        }
#pragma unroll
    for (int i = 0; i < 4; ++i) wot[i] = (cok && 4 * g + i < K) ? a.params[off.Wout + (4 * g + i) * H + col] : 0.0f;  // dH1: Wout[action 4g+i][col]
    const float b0r = cok ? a.params[off.b0 + col] : 0.0f, b1r = cok ? a.params[off.bl(0) + col] : 0.0f;
    const float bor = kok ? a.params[off.bout + n] : 0.0f;
    // ---- gradient accumulators (whole launch)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 dw0[4] = {zero4, zero4, zero4, zero4}, dw1[4] = {zero4, zero4, zero4, zero4}, dwo[4] = {zero4, zero4, zero4, zero4};
    float db0 = 0.f, db1 = 0.f, dbo = 0.f, st_pg = 0.f, st_ent = 0.f, st_kl = 0.f, st_clip = 0.f, st_cnt = 0.f;

    const long ntiles = (a.rows + TM - 1) / TM;
    const float invA = 1.0f / (float)a.A;
    Tile16 pa;
    auto xload = [&](Tile16& t, long tile) {
        const int ncols = min(KC, din);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + NTHREADS * i;
            const int r = idx >> 4, c4 = (idx & 15) * 4;
            const long row = min(tile * TM + r, a.rows - 1);
            const f32x4 q = *reinterpret_cast<const f32x4*>(a.x + row * a.x_stride + (c4 < ncols ? c4 : 0));
            t.v[i] = make_float4(q[0], q[1], q[2], q[3]);
        }
    };
    xload(pa, blockIdx.x);

    PH_DECL
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * TM, ntile = tile + gridDim.x;
        tile_store<true>(XS, pa);
        xload(pa, min(ntile, ntiles - 1));
        __syncthreads();
        // ================= forward, layer 0 =================
        f32x4 xr[2][2];
        f32x4 z0[4] = {zero4, zero4, zero4, zero4};
        {
            const unsigned xb = cf_lds_addr(XS + n * LDT + 4 * g);
            CF_LD2(xr, xb, 0);
            CF_STEP(xr, xb, 0, z0, w0); CF_STEP(xr, xb, 1, z0, w0); CF_STEP(xr, xb, 2, z0, w0); CF_STEP(xr, xb, 3, z0, w0);
            CF_STEP(xr, xb, 4, z0, w0); CF_STEP(xr, xb, 5, z0, w0); CF_STEP(xr, xb, 6, z0, w0); CF_LAST(xr, z0, w0);
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) H0s[(16 * rb + 4 * g + q) * LDT + col] = fmaxf(z0[rb][q] + b0r, 0.0f);
        __syncthreads();
        PH(0);
        // ---- per-row inputs of the head: lane (n, g) of wave w owns rows 16w + 4g + q, action n.  Unconditional clamped loads.
        // (ep_len is kept as loaded and compared in the loss phase: a boolean made here would wait for each load in place)
        int act[4], epl[4], tq[4]; float lpo[4], adv[4]; bool inb[4], first[4]; unsigned char avb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long row = row0 + 16 * wave + 4 * g + q;
            const int rr = (int)min(row, a.rows - 1);  // rows < 2^31 per launch (checked by the entry point)
            const int seq = rr / a.T, t = rr - seq * a.T;
            const int e = seq / a.A;
            act[q] = a.action[rr]; lpo[q] = a.logp_old[rr]; adv[q] = a.adv[rr];
            avb[q] = a.avail[(long)rr * a.avail_stride + (kok ? n : 0)];
            epl[q] = a.ep_len[e]; tq[q] = t; inb[q] = row < a.rows;
            first[q] = (seq - e * a.A) == 0;
        }
        // ================= hidden layer =================
        f32x4 z1[4] = {zero4, zero4, zero4, zero4};
        {
            const unsigned hb = cf_lds_addr(H0s + n * LDT + 4 * g);
            CF_LD2(xr, hb, 0);
            CF_STEP(xr, hb, 0, z1, w1n); CF_STEP(xr, hb, 1, z1, w1n); CF_STEP(xr, hb, 2, z1, w1n); CF_STEP(xr, hb, 3, z1, w1n);
            CF_STEP(xr, hb, 4, z1, w1n); CF_STEP(xr, hb, 5, z1, w1n); CF_STEP(xr, hb, 6, z1, w1n); CF_LAST(xr, z1, w1n);
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) H1s[(16 * rb + 4 * g + q) * LDT + col] = fmaxf(z1[rb][q] + b1r, 0.0f);
        __syncthreads();
        PH(1);
        // ================= head: logits of rows 16w .. 16w+15 (lane (n, g): rows 16w + 4g + q, action n) =================
        f32x4 lg = zero4;
        {
            const unsigned ab = cf_lds_addr(H1s + (16 * wave + n) * LDT + 4 * g);
            f32x4 aq[2];
            aq[0] = cf_lds128<0>(ab);
            aq[1] = cf_lds128<64>(ab); cf_wait<1>(aq[0]);
            lg = mfma16(aq[0][0], won[0], lg); lg = mfma16(aq[0][1], won[1], lg); lg = mfma16(aq[0][2], won[2], lg); lg = mfma16(aq[0][3], won[3], lg);
            aq[0] = cf_lds128<128>(ab); cf_wait<1>(aq[1]);
            lg = mfma16(aq[1][0], won[4], lg); lg = mfma16(aq[1][1], won[5], lg); lg = mfma16(aq[1][2], won[6], lg); lg = mfma16(aq[1][3], won[7], lg);
            aq[1] = cf_lds128<192>(ab); cf_wait<1>(aq[0]);
            lg = mfma16(aq[0][0], won[8], lg); lg = mfma16(aq[0][1], won[9], lg); lg = mfma16(aq[0][2], won[10], lg); lg = mfma16(aq[0][3], won[11], lg);
            cf_wait<0>(aq[1]);
            lg = mfma16(aq[1][0], won[12], lg); lg = mfma16(aq[1][1], won[13], lg); lg = mfma16(aq[1][2], won[14], lg); lg = mfma16(aq[1][3], won[15], lg);
        }
        PH(2);
        // ================= Categorical statistics, clipped surrogate, dLoss/dlogits (the arithmetic of k_mlp's M_ACTOR epilogue) =========
        f32x4 dl;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool valid_q = inb[q] && tq[q] < epl[q];
            const float z = (kok && avb[q]) ? lg[q] + bor : -1e9f;            // masked_fill(~avail, -1e9); lanes n >= K stay out below
            const float m = af_row16_max(kok ? z : -INFINITY);
            const float ex = kok ? expf(z - m) : 0.0f;
            const float s = cf_row16_sum(ex);
            const float lse = m + logf(s);
            const float p = ex * (1.0f / s);
            const float lp = z - lse;
            const float ent = cf_row16_sum(kok ? -(p * lp) : 0.0f);
            const float lpa = cf_row16_sum((kok && n == act[q]) ? lp : 0.0f);
            const float log_ratio = lpa - lpo[q];
            const float ratio = expf(log_ratio);
            const float pg1 = adv[q] * ratio;
            const float pg2 = adv[q] * fminf(fmaxf(ratio, a.clip_lo), a.clip_hi);
            const bool inr = (ratio >= a.clip_lo) && (ratio <= a.clip_hi);
            float gg;  // d min(pg1, pg2) / d ratio with torch's tie rule (grad / 2 to each operand)
            if (pg1 < pg2) gg = adv[q];
            else if (pg1 > pg2) gg = inr ? adv[q] : 0.0f;
            else gg = 0.5f * adv[q] + (inr ? 0.5f * adv[q] : 0.0f);
            if (valid_q && n == 0) {
                st_pg += invA * fminf(pg1, pg2);
                st_ent += invA * ent;
                st_kl += invA * ((ratio - 1.0f) - log_ratio);
                st_clip += (fabsf(ratio - 1.0f) > a.clip_eps) ? invA : 0.0f;
                if (first[q]) st_cnt += 1.0f;
            }
            const float gr = gg * ratio;
            float d = invA * (-gr * ((n == act[q] ? 1.0f : 0.0f) - p) + a.ent_coef * p * (lp + ent));
            if (!kok || !valid_q || z <= -5e8f) d = 0.0f;                     // padded rows / actions; masked_fill blocks the gradient
            dl[q] = d;
            dbo += d;
            DL[(16 * wave + 4 * g + q) * DLS + n] = d;
        }
        __syncthreads();
        PH(3);
        // ================= dWout partial over this wave's rows: A = own dlogits (slot g <-> row 4g + t), B = H1 rows, 16 bytes at a time ====
        {
            const unsigned hb = cf_lds_addr(H1s + (16 * wave + 4 * g) * LDT + 4 * n);
            f32x4 bq[2];
            bq[0] = cf_lds128<0>(hb);
            bq[1] = cf_lds128<LDT * 4>(hb); cf_wait<1>(bq[0]);
            dwo[0] = mfma16(dl[0], bq[0][0], dwo[0]); dwo[1] = mfma16(dl[0], bq[0][1], dwo[1]); dwo[2] = mfma16(dl[0], bq[0][2], dwo[2]); dwo[3] = mfma16(dl[0], bq[0][3], dwo[3]);
            bq[0] = cf_lds128<2 * LDT * 4>(hb); cf_wait<1>(bq[1]);
            dwo[0] = mfma16(dl[1], bq[1][0], dwo[0]); dwo[1] = mfma16(dl[1], bq[1][1], dwo[1]); dwo[2] = mfma16(dl[1], bq[1][2], dwo[2]); dwo[3] = mfma16(dl[1], bq[1][3], dwo[3]);
            bq[1] = cf_lds128<3 * LDT * 4>(hb); cf_wait<1>(bq[0]);
            dwo[0] = mfma16(dl[2], bq[0][0], dwo[0]); dwo[1] = mfma16(dl[2], bq[0][1], dwo[1]); dwo[2] = mfma16(dl[2], bq[0][2], dwo[2]); dwo[3] = mfma16(dl[2], bq[0][3], dwo[3]);
            cf_wait<0>(bq[1]);
            dwo[0] = mfma16(dl[3], bq[1][0], dwo[0]); dwo[1] = mfma16(dl[3], bq[1][1], dwo[1]); dwo[2] = mfma16(dl[3], bq[1][2], dwo[2]); dwo[3] = mfma16(dl[3], bq[1][3], dwo[3]);
        }
        // ================= dH1 = dlogits Wout (own 16 columns, all 64 rows) -> dZ1 =================
        {
            const unsigned db_ = cf_lds_addr(DL + n * DLS + 4 * g);
            f32x4 dq[2];
            f32x4 dh1[4] = {zero4, zero4, zero4, zero4};
            dq[0] = cf_lds128<0>(db_);
            dq[1] = cf_lds128<16 * DLS * 4>(db_); cf_wait<1>(dq[0]);
            dh1[0] = mfma16(dq[0][0], wot[0], dh1[0]); dh1[0] = mfma16(dq[0][1], wot[1], dh1[0]); dh1[0] = mfma16(dq[0][2], wot[2], dh1[0]); dh1[0] = mfma16(dq[0][3], wot[3], dh1[0]);
            dq[0] = cf_lds128<32 * DLS * 4>(db_); cf_wait<1>(dq[1]);
            dh1[1] = mfma16(dq[1][0], wot[0], dh1[1]); dh1[1] = mfma16(dq[1][1], wot[1], dh1[1]); dh1[1] = mfma16(dq[1][2], wot[2], dh1[1]); dh1[1] = mfma16(dq[1][3], wot[3], dh1[1]);
            dq[1] = cf_lds128<48 * DLS * 4>(db_); cf_wait<1>(dq[0]);
            dh1[2] = mfma16(dq[0][0], wot[0], dh1[2]); dh1[2] = mfma16(dq[0][1], wot[1], dh1[2]); dh1[2] = mfma16(dq[0][2], wot[2], dh1[2]); dh1[2] = mfma16(dq[0][3], wot[3], dh1[2]);
            cf_wait<0>(dq[1]);
            dh1[3] = mfma16(dq[1][0], wot[0], dh1[3]); dh1[3] = mfma16(dq[1][1], wot[1], dh1[3]); dh1[3] = mfma16(dq[1][2], wot[2], dh1[3]); dh1[3] = mfma16(dq[1][3], wot[3], dh1[3]);
            float hv[16];  // relu' from the lane's own h1 words: all 16 reads first (one LDS round trip, not sixteen)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) hv[4 * rb + q] = H1s[(16 * rb + 4 * g + q) * LDT + col];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float dz = (hv[4 * rb + q] > 0.0f) ? dh1[rb][q] : 0.0f;
                    db1 += dz;
                    DZ1[(16 * rb + 4 * g + q) * LDT + col] = dz;
                }
        }
        __syncthreads();
        PH(4);
        // ================= dW1 (own columns) and dH0 = dZ1 W1 -> dZ0 =================
        f32x4 dh[4] = {zero4, zero4, zero4, zero4};
        {
            float a1[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) a1[t] = DZ1[(4 * t + g) * LDT + col];
            const unsigned hq = cf_lds_addr(H0s + g * LDT + 4 * n), db = cf_lds_addr(DZ1 + n * LDT + 4 * g);
            f32x4 xq[3];
#define CM_LDQ(t_) xq[(t_) % 3] = cf_lds128<(4 * (t_) * LDT) * 4>(hq)
#define CM_DW1_MM(t_) do { const f32x4 x = xq[(t_) % 3]; \
                dw1[0] = mfma16(a1[t_], x[0], dw1[0]); dw1[1] = mfma16(a1[t_], x[1], dw1[1]); \
                dw1[2] = mfma16(a1[t_], x[2], dw1[2]); dw1[3] = mfma16(a1[t_], x[3], dw1[3]); } while (0)
#define CM_DW1_STEP(t_) do { CM_LDQ((t_) + 2); cf_wait<2>(xq[(t_) % 3]); CM_DW1_MM(t_); } while (0)
            CM_LDQ(0); CM_LDQ(1);
            CM_DW1_STEP(0); CM_DW1_STEP(1); CM_DW1_STEP(2); CM_DW1_STEP(3); CM_DW1_STEP(4); CM_DW1_STEP(5); CM_DW1_STEP(6);
            CM_DW1_STEP(7); CM_DW1_STEP(8); CM_DW1_STEP(9); CM_DW1_STEP(10); CM_DW1_STEP(11); CM_DW1_STEP(12); CM_DW1_STEP(13);
            CF_LD2(xr, db, 0);
            cf_wait<3>(xq[14 % 3]); CM_DW1_MM(14);
            cf_wait<2>(xq[15 % 3]); CM_DW1_MM(15);
#undef CM_DW1_STEP
#undef CM_DW1_MM
#undef CM_LDQ
            CF_STEP(xr, db, 0, dh, w1t); CF_STEP(xr, db, 1, dh, w1t); CF_STEP(xr, db, 2, dh, w1t); CF_STEP(xr, db, 3, dh, w1t);
            CF_STEP(xr, db, 4, dh, w1t); CF_STEP(xr, db, 5, dh, w1t); CF_STEP(xr, db, 6, dh, w1t); CF_LAST(xr, dh, w1t);
        }
        __syncthreads();  // every wave is done with H0 (dW1) and dZ1 (dH0): H0's buffer becomes the dZ0 tile
        {
            float hv[16];  // relu' from the lane's own h0 words (all reads first), overwritten in place by dZ0
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) hv[4 * rb + q] = H0s[(16 * rb + 4 * g + q) * LDT + col];
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float dz = (hv[4 * rb + q] > 0.0f) ? dh[rb][q] : 0.0f;
                    db0 += dz;
                    H0s[(16 * rb + 4 * g + q) * LDT + col] = dz;
                }
        }
        __syncthreads();
        PH(5);
        // ================= dW0 (own columns) =================
        {
            float a0[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) a0[t] = H0s[(4 * t + g) * LDT + col];
            const unsigned xb = cf_lds_addr(XS + g * LDT + 4 * n);
            f32x4 xq[3];
#define CM_LDB(t_) xq[(t_) % 3] = cf_lds128<(4 * (t_) * LDT) * 4>(xb)
#define CM_DW0_MM(t_) do { const f32x4 x = xq[(t_) % 3]; \
                dw0[0] = mfma16(a0[t_], x[0], dw0[0]); dw0[1] = mfma16(a0[t_], x[1], dw0[1]); \
                dw0[2] = mfma16(a0[t_], x[2], dw0[2]); dw0[3] = mfma16(a0[t_], x[3], dw0[3]); } while (0)
#define CM_DW0_STEP(t_) do { CM_LDB((t_) + 2); cf_wait<2>(xq[(t_) % 3]); CM_DW0_MM(t_); } while (0)
            CM_LDB(0); CM_LDB(1);
            CM_DW0_STEP(0); CM_DW0_STEP(1); CM_DW0_STEP(2); CM_DW0_STEP(3); CM_DW0_STEP(4); CM_DW0_STEP(5); CM_DW0_STEP(6);
            CM_DW0_STEP(7); CM_DW0_STEP(8); CM_DW0_STEP(9); CM_DW0_STEP(10); CM_DW0_STEP(11); CM_DW0_STEP(12); CM_DW0_STEP(13);
            cf_wait<1>(xq[14 % 3]); CM_DW0_MM(14);
            cf_wait<0>(xq[15 % 3]); CM_DW0_MM(15);
#undef CM_DW0_STEP
#undef CM_DW0_MM
#undef CM_LDB
        }
        __syncthreads();  // X tile, dZ0 and the dlogits tile consumed: the next tile may overwrite them
        PH(6);
    }
    PH_FLUSH;
    // ================================ this workgroup's partial row [P + 8] (torch parameter order)
    float* out = a.partial + (size_t)blockIdx.x * a.PS;
    // dW0 / dW1: lane (n, g) holds rows m = 4g + q of the 16 x 16 tile -> hidden unit c0 + 4g + q; accumulator jt = input column 4n + jt
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int hu = c0 + 4 * g + q;
        if (hu < H) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const int k = 4 * n + jt;
                if (k < din) out[off.W0 + (long)hu * din + k] = dw0[jt][q];
                if (k < H) out[off.Wl(0) + hu * H + k] = dw1[jt][q];
            }
        }
    }
    // dWout: every wave holds a [16 actions x 64 hidden] partial over its rows (lane (n, g): action 4g + q, hidden column 4n + jt)
    {
        float* sc = smem;  // 4 waves x 16 x 64 floats: the tiles are dead
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) sc[wave * 1024 + (4 * g + q) * 64 + 4 * n + jt] = dwo[jt][q];
        __syncthreads();
        for (int i = tid; i < 16 * 64; i += NTHREADS) {
            const int k = i >> 6, c = i & 63;
            if (k < K && c < H) out[off.Wout + k * H + c] = (sc[i] + sc[1024 + i]) + (sc[2048 + i] + sc[3072 + i]);
        }
    }
    // column sums held per lane: b0 / b1 (hidden column col, this lane group's rows), bout (action n, this lane's rows)
    {
        float vals[3] = {db0, db1, dbo};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            __syncthreads();
            red[tid] = vals[k];
            __syncthreads();
            if (k < 2) {
                if (g == 0 && cok) {
                    const float s = (red[64 * wave + n] + red[64 * wave + 16 + n]) + (red[64 * wave + 32 + n] + red[64 * wave + 48 + n]);
                    if (k == 0) out[off.b0 + col] = s; else out[off.bl(0) + col] = s;
                }
            } else if (tid < 16 && tid < K) {
                float s = 0.0f;
                for (int w = 0; w < 4; ++w) s += (red[64 * w + tid] + red[64 * w + 16 + tid]) + (red[64 * w + 32 + tid] + red[64 * w + 48 + tid]);
                out[off.bout + tid] = s;
            }
        }
    }
    // statistics: held by lanes n == 0
    {
        const float s0 = cm_wave_sum(st_pg), s1 = cm_wave_sum(st_ent), s2 = cm_wave_sum(st_kl), s3 = cm_wave_sum(st_clip), s4 = cm_wave_sum(st_cnt);
        __syncthreads();
        if (lane == 0) { red[wave] = s0; red[4 + wave] = s1; red[8 + wave] = s2; red[12 + wave] = s3; red[16 + wave] = s4; }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int k = 0; k < CM_NUM_STATS; ++k) out[off.P + k] = 0.0f;
            out[off.P + CM_STAT_PG] = (red[0] + red[1]) + (red[2] + red[3]);
            out[off.P + CM_STAT_ENT] = (red[4] + red[5]) + (red[6] + red[7]);
            out[off.P + CM_STAT_KL] = (red[8] + red[9]) + (red[10] + red[11]);
            out[off.P + CM_STAT_CLIP] = (red[12] + red[13]) + (red[14] + red[15]);
            out[off.P + CM_STAT_COUNT] = (red[16] + red[17]) + (red[18] + red[19]);
        }
    }
}

inline bool actor_fused_shape(const MlpArgs& a) {
    return a.din <= KC && a.H <= HP && a.L == 1 && a.dout <= 16 && x_rows_vec(a) && a.avail != nullptr && !mfma_bf16x3();
}
inline size_t actor_fused_lds_bytes() { return (size_t)(4 * TM * LDT + TM * DLS + 2 * NTHREADS) * sizeof(float); }

inline int run_actor_fused(MlpArgs a, float* grad_and_stats, void* ws, size_t ws_bytes, hipStream_t s, const char* who) {
    const int64_t P = cm_mlp_param_count(a.din, a.H, a.L, a.dout);
    const size_t need = train_ws_bytes(a.din, a.H, a.L, a.dout);
    CM_REQUIRE(ws && ws_bytes >= need, "%s: workspace too small (%zu < %zu)", who, ws_bytes, need);
    a.partial = (float*)ws; a.PS = (int)((P + CM_NUM_STATS + 63) / 64 * 64);
#ifdef CM_PHASE_PROF
    a.prof = g_prof;
#endif
    const long ntiles = (a.rows + TM - 1) / TM;
    const int grid = (int)(ntiles < MAX_GRID ? ntiles : MAX_GRID);  // two workgroups per CU
    size_t lds = actor_fused_lds_bytes();
    int grid_ = grid;
#ifdef CM_PHASE_PROF
    if (const char* e = getenv("CM_PROF_ONE_WG")) if (e[0] == '1') { lds = 100 * 1024; if (grid_ > 256) grid_ = 256; }  // clean per-phase costs
#endif
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_actor_fused), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k_actor_fused, dim3(grid_), dim3(NTHREADS), lds, s, a);
    CM_CHECK_LAUNCH(who);
    return finish_train(a, grid_, P, grad_and_stats, s, who);
}

}  // namespace
