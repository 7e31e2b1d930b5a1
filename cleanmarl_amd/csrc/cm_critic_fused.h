// cm_critic_fused.h -- value-critic forward + backward for WIDE inputs in ONE pass over the input (included by cm_mlp_critic.hip).
//
// cleanmarl/mappo_multienvs.py:554-558, :582 (critic MSE, backward) for the MAPPO central state (384 wide at BASELINE config 3).
// The split schedule of cm_mlp_split.h reads X twice from HBM (forward, then k_dw0_stream) and round-trips dZ0 [rows][64] through
// HBM: 2.4 x the algorithmic bytes, and its forward re-streams W0 (96 KB) through LDS for every 64-row tile.  Here
//   * W0 lives in REGISTERS: wave w owns hidden columns 16w .. 16w+15, its [16 x Din] slice is the B operand of
//     v_mfma_f32_16x16x4_f32 (16 registers per 64-column chunk) -- no W0 traffic at all after the prologue;
//   * the X tile (64 rows x Din) stays in LDS from the forward pass to the layer-0 weight gradient of the same tile, whose
//     accumulators (dW0 slice [16 x Din], 16 registers per chunk) also live in registers for the whole launch: X is read from HBM
//     once, dZ0 never leaves the chip;
//   * hidden activations / value head / loss are produced from the accumulator layout in place (h0, h1 stay in registers for the
//     relu' masks and the head's dot product); LDS carries only what other waves need (H0, dZ1, dZ0 tiles).
//   * every LDS read that feeds an MFMA is issued by hand, a step or two ahead of its use (cm_common.h: cf_lds128 / cf_wait; the kernel
//     runs ONE wave per SIMD, so nothing else hides LDS latency, and the compiler keeps a read next to its use).  The generated code is
//     checked for touched in-flight registers by tools/lint_lds_hazards.py; all pipelines are straight-line code for that reason.
// One workgroup per CU (registers: 2 x 16 x NC + ~230, W0 parked in the accumulation-register file), the next tile's first two X
// chunks in flight in registers during the backward phases.  Same un-normalised sums and statistic slots as k_mlp<.., M_CRITIC>;
// summation order differs (tolerance-tested, 1e-4).  Measured at config 3 (524288 rows x 384): 0.576 ms = 71 % of the fp32 MFMA peak,
// 0.85 GB of HBM traffic for 0.82 GB algorithmic (split schedule: 0.772 ms, 1.95 GB).  docs/KERNEL_NOTES.md section 3.1b.
// Shapes: Din = 65 .. 448 (NC = 2 .. 7 chunks of 64) on 16-byte aligned rows, H <= 64, ONE hidden->hidden layer, scalar output, at
// least CM_FUSED_MIN_ROWS rows (whole CUs per workgroup: small batches and batches overlapped with other kernels keep the split schedule).
#pragma once
#include "cm_mlp_train.h"

namespace {

// sum over the 16 lanes of a DPP row (row_ror:8,4,2,1), result on every lane
__device__ __forceinline__ float cf_row16_sum(float v) {
#define CM_ROR(ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    v += CM_ROR(0x128); v += CM_ROR(0x124); v += CM_ROR(0x122); v += CM_ROR(0x121);
#undef CM_ROR
    return v;
}

// [64 rows x 64 k] (LDS, A operand) x [64 k x 16 own columns] (registers, B operand) in 8 steps of 8 MFMAs: step s = 16 k (s >> 1) x 32
// rows (s & 1), two 16-byte reads per step into the ring xr[2][2], issued one step ahead.  base = cf_lds_addr(tile + n * LDT + 4 * g).
#define CF_LD2(xr, base, s_) do { (xr)[(s_) & 1][0] = cf_lds128<(32 * ((s_) & 1) * LDT + 16 * ((s_) >> 1)) * 4>(base); \
                                  (xr)[(s_) & 1][1] = cf_lds128<((32 * ((s_) & 1) + 16) * LDT + 16 * ((s_) >> 1)) * 4>(base); } while (0)
#define CF_MM8(xr, s_, acc, w) do { \
        const f32x4 x0_ = (xr)[(s_) & 1][0], x1_ = (xr)[(s_) & 1][1]; \
        f32x4& za_ = (acc)[2 * ((s_) & 1)]; f32x4& zb_ = (acc)[2 * ((s_) & 1) + 1]; \
        za_ = mfma16(x0_[0], (w)[4 * ((s_) >> 1)], za_);     zb_ = mfma16(x1_[0], (w)[4 * ((s_) >> 1)], zb_); \
        za_ = mfma16(x0_[1], (w)[4 * ((s_) >> 1) + 1], za_); zb_ = mfma16(x1_[1], (w)[4 * ((s_) >> 1) + 1], zb_); \
        za_ = mfma16(x0_[2], (w)[4 * ((s_) >> 1) + 2], za_); zb_ = mfma16(x1_[2], (w)[4 * ((s_) >> 1) + 2], zb_); \
        za_ = mfma16(x0_[3], (w)[4 * ((s_) >> 1) + 3], za_); zb_ = mfma16(x1_[3], (w)[4 * ((s_) >> 1) + 3], zb_); } while (0)
#define CF_STEP(xr, base, s_, acc, w) do { CF_LD2(xr, base, (s_) + 1); cf_wait<2>((xr)[(s_) & 1][0], (xr)[(s_) & 1][1]); CF_MM8(xr, s_, acc, w); } while (0)
#define CF_LAST(xr, acc, w) do { cf_wait<0>((xr)[1][0], (xr)[1][1]); CF_MM8(xr, 7, acc, w); } while (0)

// SAVED (round 6): the layer-0 activations h0 = relu(x W0^T + b0) of every row come from memory (MlpArgs::dz0, [rows][HP]) instead of the NC-chunk
// product -- the value pass at the head of the update computed them with the SAME parameters the first critic epoch uses (cm_value_pass_keep_h0_ld /
// cm_critic_fwd_bwd_h0_ld).  The X tile still goes to LDS (the layer-0 weight gradient contracts it), W0 is not loaded at all.  At config 3 (384-wide
// state) the layer-0 product is 39 % of the kernel's time (profiles/r06_phase_critic.txt) against 64 of 448 floats per row of extra traffic.
template <int NC, bool SAVED = false>
__global__ __launch_bounds__(NTHREADS, 1) void k_critic_fused(const MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* XS = smem;                       // [NC][TM][LDT] the tile's input, chunk by chunk
    float* H0s = XS + NC * TM * LDT;        // [TM][LDT] h0 (forward / dW1), then dZ0
    float* DZ1 = H0s + TM * LDT;            // [TM][LDT]
    float* vpart = DZ1 + TM * LDT;          // [4][TM] partial values per wave
    float* dv = vpart + 4 * TM;             // [TM] dLoss/dv per row
    float* red = dv + TM;                   // 2 * NTHREADS
    const Offsets off = make_offsets(a.din, a.H, 1, 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const int c0 = 16 * wave, col = c0 + n, H = a.H, din = a.din;
    const bool cok = col < H;
    const int lrow = tid >> 2, part = tid & 3;  // loss phase: four lanes per row
    // ---- weights of this wave's 16 hidden columns -> registers
    float w0[SAVED ? 1 : NC][16], w1n[16], w1t[16];
    if constexpr (!SAVED) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 64 * c + 16 * j + 4 * g + i;
                w0[c][4 * j + i] = (cok && k < din) ? a.params[off.W0 + (long)col * din + k] : 0.0f;
            }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 16 * j + 4 * g + i;  // nt image: W1[col][k]; tn image: W1[k][col]
            w1n[4 * j + i] = (cok && k < H) ? a.params[off.Wl(0) + col * H + k] : 0.0f;
            w1t[4 * j + i] = (cok && k < H) ? a.params[off.Wl(0) + k * H + col] : 0.0f;
        }
    // park the layer-0 slice in accumulation registers (the MFMA B operand may come from either file): left to itself the allocator
    // keeps all 16 NC of them in the 256 architectural registers and spills the X prefetch sets to scratch
    if constexpr (!SAVED) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" : "+a"(w0[c][k]));
    }
    const float b0r = cok ? a.params[off.b0 + col] : 0.0f, b1r = cok ? a.params[off.bl(0) + col] : 0.0f;
    const float wo = cok ? a.params[off.Wout + col] : 0.0f, bout = a.params[off.bout];
    // ---- gradient accumulators (whole launch)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 dw0[NC][4], dw1[4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) dw0[c][jt] = zero4;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) dw1[jt] = zero4;
    float db0 = 0.f, db1 = 0.f, dwo = 0.f, dbo = 0.f, st_vl = 0.f, st_cnt = 0.f;

    const long ntiles = (a.rows + TM - 1) / TM;
    // X chunk of 64 rows x 64 columns = 16 floats per thread, register-staged (two chunks in flight).  Branch-free 16-byte loads (the
    // launcher takes this kernel only for 16-byte aligned rows): rows past the end re-read the last row (their loss derivative is 0,
    // so they add nothing), column blocks past the input width re-read block 0 (they meet zero weights in the forward pass and only feed
    // gradient columns that are never written) -- unconditional loads let the compiler count the requests in flight (vmcnt) exactly
    Tile16 pa, pb;
    auto xload = [&](Tile16& t, long tile, int c) {
        const int ncols = min(KC, din - c * KC);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + NTHREADS * i;
            const int r = idx >> 4, c4 = (idx & 15) * 4;
            const long row = min(tile * TM + r, a.rows - 1);
            // a true 16-byte vector load: a struct-to-struct float4 assignment becomes a memcpy into a private-memory copy of the set
            const f32x4 q = *reinterpret_cast<const f32x4*>(a.x + row * a.x_stride + c * KC + (c4 < ncols ? c4 : 0));
            t.v[i] = make_float4(q[0], q[1], q[2], q[3]);
        }
    };
    auto xstore = [&](float* dst, const Tile16& t) { tile_store<true>(dst, t); };
    // SAVED: the tile's h0 rows, requested a tile ahead like X (same clamped, branch-free 16-byte loads; leading dimension HP)
    Tile16 ph;
    auto hload = [&](Tile16& t, long tile) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + NTHREADS * i;
            const int r = idx >> 4, c4 = (idx & 15) * 4;
            const long row = min(tile * TM + r, a.rows - 1);
            const f32x4 q = *reinterpret_cast<const f32x4*>(a.dz0 + row * HP + c4);
            t.v[i] = make_float4(q[0], q[1], q[2], q[3]);
        }
    };
    if ((long)blockIdx.x < ntiles) { xload(pa, blockIdx.x, 0); xload(pb, blockIdx.x, 1); if constexpr (SAVED) hload(ph, blockIdx.x); }

    PH_DECL
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * TM, ntile = tile + gridDim.x;
        // the loss's targets for this tile are requested now and consumed after the forward pass (central critic: the row's env,
        // agents part and part + 4; more than 8 agents are fetched in the loss phase itself).  Unconditional, clamped indices.
        const long lr = min(row0 + lrow, a.rows - 1);
        const long le = lr / a.T; const int lt = (int)(lr - le * a.T);  // central critic: env and step of the row
        const float rp0 = a.ret[a.per_agent ? lr : (le * a.A + min(part, a.A - 1)) * a.T + lt];
        const float rp1 = a.ret[a.per_agent ? lr : (le * a.A + min(part + 4, a.A - 1)) * a.T + lt];
        // ================= forward, layer 0: chunk c + 1 is written to LDS BEFORE chunk c is multiplied, so the barrier after the
        // products finds every wave's stores long finished (one exposed store + barrier per tile instead of one per chunk)
        f32x4 z0[4] = {zero4, zero4, zero4, zero4};
        f32x4 xr[2][2];
        float h0[16];
        if constexpr (SAVED) {
            // no layer-0 product: the X chunks go to LDS in the order (and through the register sets) of the product loop below, h0 comes from memory
            xstore(XS, pa);
            if (NC > 2) xload(pa, tile, 2); else xload(pa, min(ntile, ntiles - 1), 0);
#pragma unroll
            for (int c = 0; c + 1 < NC; ++c) {
                Tile16& set = ((c + 1) & 1) ? pb : pa;
                xstore(XS + (c + 1) * TM * LDT, set);
                if (c + 3 < NC) xload(set, tile, c + 3);
                else xload(set, min(ntile, ntiles - 1), c + 3 - NC);
            }
            xstore(H0s, ph);
            hload(ph, min(ntile, ntiles - 1));
            PH(0);
            if (NC & 1) { const Tile16 t = pa; pa = pb; pb = t; }
            __syncthreads();
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) h0[4 * rb + q] = H0s[(16 * rb + 4 * g + q) * LDT + col];
        } else {
        xstore(XS, pa);
        if (NC > 2) xload(pa, tile, 2); else xload(pa, min(ntile, ntiles - 1), 0);
        __syncthreads();
        const unsigned xa_f = cf_lds_addr(XS + n * LDT + 4 * g);  // forward A operand: rows 16 rb + n, columns 16 j + 4 g ..
        CF_LD2(xr, xa_f, 0);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const unsigned xc = xa_f + c * TM * LDT * 4;
            CF_STEP(xr, xc, 0, z0, w0[c]);
            CF_STEP(xr, xc, 1, z0, w0[c]);
            if (c + 1 < NC) {
                // under the first products: the next chunk goes to LDS and its register set requests the chunk two ahead (the next
                // tile's first two chunks at the end; the last tile re-reads itself)
                Tile16& set = ((c + 1) & 1) ? pb : pa;
                xstore(XS + (c + 1) * TM * LDT, set);
                if (c + 3 < NC) xload(set, tile, c + 3);
                else xload(set, min(ntile, ntiles - 1), c + 3 - NC);
            }
            CF_STEP(xr, xc, 2, z0, w0[c]);
            CF_STEP(xr, xc, 3, z0, w0[c]);
            CF_STEP(xr, xc, 4, z0, w0[c]);
            CF_STEP(xr, xc, 5, z0, w0[c]);
            CF_STEP(xr, xc, 6, z0, w0[c]);
            if (c + 1 < NC) {
                // the barrier for chunk c + 1 (stored five steps ago) goes BEFORE the last step, whose products then cover the first
                // reads of the next chunk: no LDS latency bubble at the chunk boundary
                __syncthreads();
                CF_LD2(xr, xc + TM * LDT * 4, 0);
                cf_wait<2>(xr[1][0], xr[1][1]);
                CF_MM8(xr, 7, z0, w0[c]);
            } else {
                CF_LAST(xr, z0, w0[c]);
            }
        }
        PH(0);
        if (NC & 1) { const Tile16 t = pa; pa = pb; pb = t; }  // odd chunk count: the next tile's chunk 0 was requested into pb
        // h0 = relu(z0 + b0): kept in registers (relu' mask), written to LDS for the other waves
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                h0[4 * rb + q] = fmaxf(z0[rb][q] + b0r, 0.0f);
                H0s[(16 * rb + 4 * g + q) * LDT + col] = h0[4 * rb + q];
            }
        __syncthreads();
        }
        PH(1);
        // ================= hidden layer + value head =================
        f32x4 z1[4] = {zero4, zero4, zero4, zero4};
        {
            const unsigned hb = cf_lds_addr(H0s + n * LDT + 4 * g);
            CF_LD2(xr, hb, 0);
            CF_STEP(xr, hb, 0, z1, w1n); CF_STEP(xr, hb, 1, z1, w1n); CF_STEP(xr, hb, 2, z1, w1n); CF_STEP(xr, hb, 3, z1, w1n);
            CF_STEP(xr, hb, 4, z1, w1n); CF_STEP(xr, hb, 5, z1, w1n); CF_STEP(xr, hb, 6, z1, w1n); CF_LAST(xr, z1, w1n);
        }
        float h1[16];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                h1[4 * rb + q] = fmaxf(z1[rb][q] + b1r, 0.0f);
                const float vp = cf_row16_sum(h1[4 * rb + q] * wo);  // this wave's 16 columns of the row's dot product
                if (n == 0) vpart[wave * TM + 16 * rb + 4 * g + q] = vp;
            }
        __syncthreads();
        PH(2);
        // ================= loss: four lanes per row (lane & 3 = share of the agents), row = tid >> 2 =================
        {
            const long row = row0 + lrow;
            float sd = 0.0f, sq = 0.0f;
            bool live = false;
            if (row < a.rows) {
                const float v = ((vpart[lrow] + vpart[TM + lrow]) + (vpart[2 * TM + lrow] + vpart[3 * TM + lrow])) + bout;
                if (a.per_agent) {
                    const long seq = row / a.T; const int t = (int)(row - seq * a.T);
                    const long e = seq / a.A;
                    live = t < a.ep_len[e];
                    if (live && part == 0) {
                        const float df = v - rp0;
                        sd = df; sq = df * df;
                        if (seq - e * a.A == 0) st_cnt += 1.0f;
                    }
                } else {
                    live = lt < a.ep_len[le];
                    if (live) {
                        if (part < a.A) { const float df = v - rp0; sd += df; sq += df * df; }
                        if (part + 4 < a.A) { const float df = v - rp1; sd += df; sq += df * df; }
                        for (int q = part + 8; q < a.A; q += 4) {
                            const float df = v - a.ret[(le * a.A + q) * a.T + lt];
                            sd += df; sq += df * df;
                        }
                        if (part == 0) st_cnt += 1.0f;
                    }
                }
            }
            // (share 0 + share 1) + (share 2 + share 3) on every lane of the quad: quad_perm [1,0,3,2] then [2,3,0,1]
#define CM_QP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
            sd += CM_QP(sd, 0xB1); sq += CM_QP(sq, 0xB1);
            sd += CM_QP(sd, 0x4E); sq += CM_QP(sq, 0x4E);
#undef CM_QP
            if (part == 0) {
                const float invA = 1.0f / (float)a.A;
                const float d = live ? 2.0f * invA * sd : 0.0f;
                if (live) st_vl += invA * sq;
                dv[lrow] = d;
                dbo += d;
            }
        }
        __syncthreads();
        PH(3);
        // ================= backward: head -> dZ1 (own columns, registers -> LDS) =================
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 16 * rb + 4 * g + q;
                const float d = dv[r];
                dwo = fmaf(d, h1[4 * rb + q], dwo);
                const float dz = (h1[4 * rb + q] > 0.0f) ? d * wo : 0.0f;
                db1 += dz;
                DZ1[r * LDT + col] = dz;
            }
        __syncthreads();
        PH(4);
        // dW1[own n][k] += sum_rows dZ1[row][n] h0[row][k]: A operand (this wave's columns of dZ1) loaded once, contraction over rows.
        // B operand: lane (n, g) reads h0[row 4t+g][4n .. 4n+3] with ONE 16-byte LDS read (conflict-free per 16-lane group) and feeds
        // four accumulators, so accumulator i holds input columns 4n + i (a free choice of the MFMA's N index -> column map)
        f32x4 dh[4] = {zero4, zero4, zero4, zero4};
        {
            float a1[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) a1[t] = DZ1[(4 * t + g) * LDT + col];
            // 16 steps (rows 4 t + g), reads two steps ahead; the first reads of the dH0 product below join the queue at the end
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
            PH(5);
            // dH0 = dZ1 W1 (own 16 columns), through relu' -> dZ0 in registers
            CF_STEP(xr, db, 0, dh, w1t); CF_STEP(xr, db, 1, dh, w1t); CF_STEP(xr, db, 2, dh, w1t); CF_STEP(xr, db, 3, dh, w1t);
            CF_STEP(xr, db, 4, dh, w1t); CF_STEP(xr, db, 5, dh, w1t); CF_STEP(xr, db, 6, dh, w1t); CF_LAST(xr, dh, w1t);
        }
        __syncthreads();  // every wave is done with H0 (dW1) and dZ1 (dH0): H0's buffer becomes the dZ0 tile
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float dz = (h0[4 * rb + q] > 0.0f) ? dh[rb][q] : 0.0f;
                db0 += dz;
                H0s[(16 * rb + 4 * g + q) * LDT + col] = dz;
            }
        __syncthreads();
        PH(6);
        // dW0[own n][k] += sum_rows dZ0[row][n] X[row][k]: the X tile is still in LDS
        {
            float a0[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) a0[t] = H0s[(4 * t + g) * LDT + col];
            const unsigned xb = cf_lds_addr(XS + g * LDT + 4 * n);  // B operand: rows 4 t + g, columns 4 n ..
            f32x4 xq[3];
            // step (c, t): rows 4 t + g of chunk c, one 16-byte read, 4 MFMAs; reads two steps (8 MFMAs) ahead of their use
#define CM_LDB(t_, i_) xq[(i_) % 3] = cf_lds128<(4 * ((t_) & 15) * LDT) * 4>(xb + ((i_) >> 4) * (TM * LDT * 4))
#define CM_DW0_STEP(t_) do { \
                const int idx = 16 * c + (t_); \
                if (idx + 2 < 16 * NC) { CM_LDB((t_) + 2, idx + 2); cf_wait<2>(xq[idx % 3]); } \
                else if (idx + 1 < 16 * NC) cf_wait<1>(xq[idx % 3]); \
                else cf_wait<0>(xq[idx % 3]); \
                const f32x4 x = xq[idx % 3]; \
                dw0[c][0] = mfma16(a0[t_], x[0], dw0[c][0]); dw0[c][1] = mfma16(a0[t_], x[1], dw0[c][1]); \
                dw0[c][2] = mfma16(a0[t_], x[2], dw0[c][2]); dw0[c][3] = mfma16(a0[t_], x[3], dw0[c][3]); } while (0)
            CM_LDB(0, 0); CM_LDB(1, 1);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                CM_DW0_STEP(0); CM_DW0_STEP(1); CM_DW0_STEP(2); CM_DW0_STEP(3); CM_DW0_STEP(4); CM_DW0_STEP(5); CM_DW0_STEP(6); CM_DW0_STEP(7);
                CM_DW0_STEP(8); CM_DW0_STEP(9); CM_DW0_STEP(10); CM_DW0_STEP(11); CM_DW0_STEP(12); CM_DW0_STEP(13); CM_DW0_STEP(14); CM_DW0_STEP(15);
            }
#undef CM_DW0_STEP
#undef CM_LDB
        }
        PH(7);
        __syncthreads();  // X tile and dZ0 consumed: the next tile may overwrite them
        PH(8);
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
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    const int k = 64 * c + 4 * n + jt;
                    if (k < din) out[off.W0 + (long)hu * din + k] = dw0[c][jt][q];
                }
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const int k = 4 * n + jt;
                if (k < H) out[off.Wl(0) + hu * H + k] = dw1[jt][q];
            }
        }
    }
    // column sums held per lane (column col, this lane group's rows): fold the four lane groups
    {
        float vals[4] = {db0, db1, dwo, 0.0f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            __syncthreads();
            red[tid] = vals[k];
            __syncthreads();
            if (g == 0 && cok) {
                const float s = (red[64 * wave + n] + red[64 * wave + 16 + n]) + (red[64 * wave + 32 + n] + red[64 * wave + 48 + n]);
                if (k == 0) out[off.b0 + col] = s; else if (k == 1) out[off.bl(0) + col] = s; else out[off.Wout + col] = s;
            }
        }
    }
    // scalars held by the first lane of every quad in all four waves: dbout and the two statistics
    {
        const float s0 = cm_wave_sum(dbo), s1 = cm_wave_sum(st_vl), s2 = cm_wave_sum(st_cnt);
        __syncthreads();
        if (lane == 0) { red[wave] = s0; red[4 + wave] = s1; red[8 + wave] = s2; }
        __syncthreads();
        if (tid == 0) {
            out[off.bout] = (red[0] + red[1]) + (red[2] + red[3]);
#pragma unroll
            for (int k = 0; k < CM_NUM_STATS; ++k) out[off.P + k] = 0.0f;
            out[off.P + CM_STAT_VLOSS] = (red[4] + red[5]) + (red[6] + red[7]);
            out[off.P + CM_STAT_COUNT] = (red[8] + red[9]) + (red[10] + red[11]);
        }
    }
}

// ---- TWO row tiles per iteration (round 6, OPT-IN: critic_schedule = "fused2" -- measured 5 % slower than the kernel above at config 4; NC <= 2: 2 x (NC + 2) tiles of LDS = 139 KB at NC = 2 -- config 4's per-agent critic on 115-wide
// observations).  The kernel above runs ONE wave per SIMD: nothing hides the latency of its five short phases (h0 -> LDS, hidden layer +
// value, loss, dZ1 -> LDS, dW1 / dH0 + dZ0 -> LDS: 12 k of the ~22 k cycles of a two-chunk tile for 6 k cycles of MFMA issue) except more
// independent work between the same barriers.  Two workgroups per CU do not fit the register file (300 registers: measured with spills,
// 2.93 -> 3.01 ms, docs/HISTORY.md); two TILES per workgroup do: every phase handles tile 2p and tile 2p + 1 back to back, the barriers
// (8 per iteration) are paid once per 128 rows, and the VALU / LDS phases of the two tiles interleave in the instruction stream.  Same
// arithmetic per tile; a workgroup's partial row sums other tiles than the one-tile kernel's (pairs instead of a grid stride), so the
// results agree with it to rounding, not bit for bit.
template <int NC>
__global__ __launch_bounds__(NTHREADS, 1) void k_critic_fused2(const MlpArgs a) {
    constexpr int TT = 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* XS = smem;                            // [TT][NC][TM][LDT] the tiles' input, chunk by chunk
    float* H0s = XS + TT * NC * TM * LDT;        // [TT][TM][LDT] h0 (forward / dW1), then dZ0
    float* DZ1 = H0s + TT * TM * LDT;            // [TT][TM][LDT]
    float* vpart = DZ1 + TT * TM * LDT;          // [TT][4][TM] partial values per wave
    float* dv = vpart + TT * 4 * TM;             // [TT][TM] dLoss/dv per row
    float* red = dv + TT * TM;                   // 2 * NTHREADS
    const Offsets off = make_offsets(a.din, a.H, 1, 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
    const int c0 = 16 * wave, col = c0 + n, H = a.H, din = a.din;
    const bool cok = col < H;
    const int lrow = tid >> 2, part = tid & 3;  // loss phase: four lanes per row
    // ---- weights of this wave's 16 hidden columns -> registers
    float w0[NC][16], w1n[16], w1t[16];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 64 * c + 16 * j + 4 * g + i;
                w0[c][4 * j + i] = (cok && k < din) ? a.params[off.W0 + (long)col * din + k] : 0.0f;
            }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 16 * j + 4 * g + i;  // nt image: W1[col][k]; tn image: W1[k][col]
            w1n[4 * j + i] = (cok && k < H) ? a.params[off.Wl(0) + col * H + k] : 0.0f;
            w1t[4 * j + i] = (cok && k < H) ? a.params[off.Wl(0) + k * H + col] : 0.0f;
        }
    // park the layer-0 slice in accumulation registers (the MFMA B operand may come from either file): left to itself the allocator
    // keeps all 16 NC of them in the 256 architectural registers and spills the X prefetch sets to scratch
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("" : "+a"(w0[c][k]));
    const float b0r = cok ? a.params[off.b0 + col] : 0.0f, b1r = cok ? a.params[off.bl(0) + col] : 0.0f;
    const float wo = cok ? a.params[off.Wout + col] : 0.0f, bout = a.params[off.bout];
    // ---- gradient accumulators (whole launch)
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 dw0[NC][4], dw1[4];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) dw0[c][jt] = zero4;
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) dw1[jt] = zero4;
    float db0 = 0.f, db1 = 0.f, dwo = 0.f, dbo = 0.f, st_vl = 0.f, st_cnt = 0.f;

    const long ntiles = (a.rows + TM - 1) / TM;
    // X chunk of 64 rows x 64 columns = 16 floats per thread, register-staged (two chunks in flight).  Branch-free 16-byte loads (the
    // launcher takes this kernel only for 16-byte aligned rows): rows past the end re-read the last row (their loss derivative is 0,
    // so they add nothing), column blocks past the input width re-read block 0 (they meet zero weights in the forward pass and only feed
    // gradient columns that are never written) -- unconditional loads let the compiler count the requests in flight (vmcnt) exactly
    Tile16 pa[TT], pb[TT];
    auto xload = [&](Tile16& t, long tile, int c) {
        const int ncols = min(KC, din - c * KC);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid + NTHREADS * i;
            const int r = idx >> 4, c4 = (idx & 15) * 4;
            const long row = min(tile * TM + r, a.rows - 1);
            // a true 16-byte vector load: a struct-to-struct float4 assignment becomes a memcpy into a private-memory copy of the set
            const f32x4 q = *reinterpret_cast<const f32x4*>(a.x + row * a.x_stride + c * KC + (c4 < ncols ? c4 : 0));
            t.v[i] = make_float4(q[0], q[1], q[2], q[3]);
        }
    };
    auto xstore = [&](float* dst, const Tile16& t) { tile_store<true>(dst, t); };
    static_assert(NC == 2, "the two-tile kernel is written for two input chunks");
    const long npairs = (ntiles + TT - 1) / TT;
    if ((long)blockIdx.x < npairs) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) { xload(pa[tt], min(TT * (long)blockIdx.x + tt, ntiles - 1), 0); xload(pb[tt], min(TT * (long)blockIdx.x + tt, ntiles - 1), 1); }
    }

    PH_DECL
    for (long pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
        // tile tt of the pair; a pair's second tile may lie past the end: its rows are all invalid (loss derivative 0, loads clamped to the last row)
        const long npair = pair + gridDim.x;
        long row0[TT];
        float rp0[TT], rp1[TT];
        long le[TT]; int lt[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            row0[tt] = (TT * pair + tt) * TM;
            // the loss's targets of the tile, requested now and consumed after the forward pass (see k_critic_fused)
            const long lr = min(row0[tt] + lrow, a.rows - 1);
            le[tt] = lr / a.T; lt[tt] = (int)(lr - le[tt] * a.T);
            rp0[tt] = a.ret[a.per_agent ? lr : (le[tt] * a.A + min(part, a.A - 1)) * a.T + lt[tt]];
            rp1[tt] = a.ret[a.per_agent ? lr : (le[tt] * a.A + min(part + 4, a.A - 1)) * a.T + lt[tt]];
        }
        // ================= forward, layer 0: products (tile 0, chunk 0), (tile 1, chunk 0), (tile 0, chunk 1), (tile 1, chunk 1); chunk 1 of
        // both tiles goes to LDS under the first product, its barrier sits a whole product later
        f32x4 z0[TT][4];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) z0[tt][i] = zero4;
            xstore(XS + tt * NC * TM * LDT, pa[tt]);
            xload(pa[tt], min(TT * npair + tt, ntiles - 1), 0);
        }
        __syncthreads();
        f32x4 xr[2][2];
        {
            const unsigned xa0 = cf_lds_addr(XS + n * LDT + 4 * g), xa1 = xa0 + NC * TM * LDT * 4;  // forward A operands of tile 0 / tile 1
            CF_LD2(xr, xa0, 0);
            CF_STEP(xr, xa0, 0, z0[0], w0[0]); CF_STEP(xr, xa0, 1, z0[0], w0[0]);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                xstore(XS + (tt * NC + 1) * TM * LDT, pb[tt]);
                xload(pb[tt], min(TT * npair + tt, ntiles - 1), 1);
            }
            CF_STEP(xr, xa0, 2, z0[0], w0[0]); CF_STEP(xr, xa0, 3, z0[0], w0[0]); CF_STEP(xr, xa0, 4, z0[0], w0[0]);
            CF_STEP(xr, xa0, 5, z0[0], w0[0]); CF_STEP(xr, xa0, 6, z0[0], w0[0]);
            CF_LD2(xr, xa1, 0); cf_wait<2>(xr[1][0], xr[1][1]); CF_MM8(xr, 7, z0[0], w0[0]);
            CF_STEP(xr, xa1, 0, z0[1], w0[0]); CF_STEP(xr, xa1, 1, z0[1], w0[0]); CF_STEP(xr, xa1, 2, z0[1], w0[0]); CF_STEP(xr, xa1, 3, z0[1], w0[0]);
            CF_STEP(xr, xa1, 4, z0[1], w0[0]); CF_STEP(xr, xa1, 5, z0[1], w0[0]); CF_STEP(xr, xa1, 6, z0[1], w0[0]);
            __syncthreads();  // chunk 1 of both tiles (stored a product and a half ago)
            const unsigned xb0 = xa0 + TM * LDT * 4, xb1 = xa1 + TM * LDT * 4;
            CF_LD2(xr, xb0, 0); cf_wait<2>(xr[1][0], xr[1][1]); CF_MM8(xr, 7, z0[1], w0[0]);
            CF_STEP(xr, xb0, 0, z0[0], w0[1]); CF_STEP(xr, xb0, 1, z0[0], w0[1]); CF_STEP(xr, xb0, 2, z0[0], w0[1]); CF_STEP(xr, xb0, 3, z0[0], w0[1]);
            CF_STEP(xr, xb0, 4, z0[0], w0[1]); CF_STEP(xr, xb0, 5, z0[0], w0[1]); CF_STEP(xr, xb0, 6, z0[0], w0[1]);
            CF_LD2(xr, xb1, 0); cf_wait<2>(xr[1][0], xr[1][1]); CF_MM8(xr, 7, z0[0], w0[1]);
            CF_STEP(xr, xb1, 0, z0[1], w0[1]); CF_STEP(xr, xb1, 1, z0[1], w0[1]); CF_STEP(xr, xb1, 2, z0[1], w0[1]); CF_STEP(xr, xb1, 3, z0[1], w0[1]);
            CF_STEP(xr, xb1, 4, z0[1], w0[1]); CF_STEP(xr, xb1, 5, z0[1], w0[1]); CF_STEP(xr, xb1, 6, z0[1], w0[1]); CF_LAST(xr, z0[1], w0[1]);
        }
        PH(0);
        // h0 = relu(z0 + b0): kept in registers (relu' mask), written to LDS for the other waves
        float h0[TT][16];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h0[tt][4 * rb + q] = fmaxf(z0[tt][rb][q] + b0r, 0.0f);
                    H0s[tt * TM * LDT + (16 * rb + 4 * g + q) * LDT + col] = h0[tt][4 * rb + q];
                }
        __syncthreads();
        PH(1);
        // ================= hidden layer + value head =================
        f32x4 z1[TT][4];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int i = 0; i < 4; ++i) z1[tt][i] = zero4;
        {
            const unsigned hb0 = cf_lds_addr(H0s + n * LDT + 4 * g), hb1 = hb0 + TM * LDT * 4;
            CF_LD2(xr, hb0, 0);
            CF_STEP(xr, hb0, 0, z1[0], w1n); CF_STEP(xr, hb0, 1, z1[0], w1n); CF_STEP(xr, hb0, 2, z1[0], w1n); CF_STEP(xr, hb0, 3, z1[0], w1n);
            CF_STEP(xr, hb0, 4, z1[0], w1n); CF_STEP(xr, hb0, 5, z1[0], w1n); CF_STEP(xr, hb0, 6, z1[0], w1n);
            CF_LD2(xr, hb1, 0); cf_wait<2>(xr[1][0], xr[1][1]); CF_MM8(xr, 7, z1[0], w1n);
            CF_STEP(xr, hb1, 0, z1[1], w1n); CF_STEP(xr, hb1, 1, z1[1], w1n); CF_STEP(xr, hb1, 2, z1[1], w1n); CF_STEP(xr, hb1, 3, z1[1], w1n);
            CF_STEP(xr, hb1, 4, z1[1], w1n); CF_STEP(xr, hb1, 5, z1[1], w1n); CF_STEP(xr, hb1, 6, z1[1], w1n); CF_LAST(xr, z1[1], w1n);
        }
        float h1[TT][16];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    h1[tt][4 * rb + q] = fmaxf(z1[tt][rb][q] + b1r, 0.0f);
                    const float vp = cf_row16_sum(h1[tt][4 * rb + q] * wo);  // this wave's 16 columns of the row's dot product
                    if (n == 0) vpart[(tt * 4 + wave) * TM + 16 * rb + 4 * g + q] = vp;
                }
        __syncthreads();
        PH(2);
        // ================= loss: four lanes per row (lane & 3 = share of the agents), row = tid >> 2, one tile after the other =================
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            const long row = row0[tt] + lrow;
            const float* vp = vpart + tt * 4 * TM;
            float sd = 0.0f, sq = 0.0f;
            bool live = false;
            if (row < a.rows) {
                const float v = ((vp[lrow] + vp[TM + lrow]) + (vp[2 * TM + lrow] + vp[3 * TM + lrow])) + bout;
                if (a.per_agent) {
                    const long seq = row / a.T; const int t = (int)(row - seq * a.T);
                    const long e = seq / a.A;
                    live = t < a.ep_len[e];
                    if (live && part == 0) {
                        const float df = v - rp0[tt];
                        sd = df; sq = df * df;
                        if (seq - e * a.A == 0) st_cnt += 1.0f;
                    }
                } else {
                    live = lt[tt] < a.ep_len[le[tt]];
                    if (live) {
                        if (part < a.A) { const float df = v - rp0[tt]; sd += df; sq += df * df; }
                        if (part + 4 < a.A) { const float df = v - rp1[tt]; sd += df; sq += df * df; }
                        for (int q = part + 8; q < a.A; q += 4) {
                            const float df = v - a.ret[(le[tt] * a.A + q) * a.T + lt[tt]];
                            sd += df; sq += df * df;
                        }
                        if (part == 0) st_cnt += 1.0f;
                    }
                }
            }
#define CM_QP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
            sd += CM_QP(sd, 0xB1); sq += CM_QP(sq, 0xB1);
            sd += CM_QP(sd, 0x4E); sq += CM_QP(sq, 0x4E);
#undef CM_QP
            if (part == 0) {
                const float invA = 1.0f / (float)a.A;
                const float d = live ? 2.0f * invA * sd : 0.0f;
                if (live) st_vl += invA * sq;
                dv[tt * TM + lrow] = d;
                dbo += d;
            }
        }
        __syncthreads();
        PH(3);
        // ================= backward: head -> dZ1 (own columns, registers -> LDS) =================
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int r = 16 * rb + 4 * g + q;
                    const float d = dv[tt * TM + r];
                    dwo = fmaf(d, h1[tt][4 * rb + q], dwo);
                    const float dz = (h1[tt][4 * rb + q] > 0.0f) ? d * wo : 0.0f;
                    db1 += dz;
                    DZ1[tt * TM * LDT + r * LDT + col] = dz;
                }
        __syncthreads();
        PH(4);
        // dW1 += dZ1^T h0 (both tiles into the same accumulators), then dH0 = dZ1 W1 of both tiles
        f32x4 dh[TT][4];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dh[tt][i] = zero4;
            const float* DZt = DZ1 + tt * TM * LDT;
            float a1[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) a1[t] = DZt[(4 * t + g) * LDT + col];
            const unsigned hq = cf_lds_addr(H0s + tt * TM * LDT + g * LDT + 4 * n), db = cf_lds_addr(DZt + n * LDT + 4 * g);
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
            CF_STEP(xr, db, 0, dh[tt], w1t); CF_STEP(xr, db, 1, dh[tt], w1t); CF_STEP(xr, db, 2, dh[tt], w1t); CF_STEP(xr, db, 3, dh[tt], w1t);
            CF_STEP(xr, db, 4, dh[tt], w1t); CF_STEP(xr, db, 5, dh[tt], w1t); CF_STEP(xr, db, 6, dh[tt], w1t); CF_LAST(xr, dh[tt], w1t);
        }
        PH(5);
        __syncthreads();  // every wave is done with H0 (dW1) and dZ1 (dH0): H0's buffers become the dZ0 tiles
#pragma unroll
        for (int tt = 0; tt < TT; ++tt)
#pragma unroll
            for (int rb = 0; rb < 4; ++rb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float dz = (h0[tt][4 * rb + q] > 0.0f) ? dh[tt][rb][q] : 0.0f;
                    db0 += dz;
                    H0s[tt * TM * LDT + (16 * rb + 4 * g + q) * LDT + col] = dz;
                }
        __syncthreads();
        PH(6);
        // dW0[own n][k] += sum_rows dZ0[row][n] X[row][k]: both X tiles are still in LDS
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            float a0[16];
#pragma unroll
            for (int t = 0; t < 16; ++t) a0[t] = H0s[tt * TM * LDT + (4 * t + g) * LDT + col];
            const unsigned xb = cf_lds_addr(XS + tt * NC * TM * LDT + g * LDT + 4 * n);  // B operand: rows 4 t + g, columns 4 n ..
            f32x4 xq[3];
#define CM_LDB(t_, i_) xq[(i_) % 3] = cf_lds128<(4 * ((t_) & 15) * LDT) * 4>(xb + ((i_) >> 4) * (TM * LDT * 4))
#define CM_DW0_STEP(t_) do { \
                const int idx = 16 * c + (t_); \
                if (idx + 2 < 16 * NC) { CM_LDB((t_) + 2, idx + 2); cf_wait<2>(xq[idx % 3]); } \
                else if (idx + 1 < 16 * NC) cf_wait<1>(xq[idx % 3]); \
                else cf_wait<0>(xq[idx % 3]); \
                const f32x4 x = xq[idx % 3]; \
                dw0[c][0] = mfma16(a0[t_], x[0], dw0[c][0]); dw0[c][1] = mfma16(a0[t_], x[1], dw0[c][1]); \
                dw0[c][2] = mfma16(a0[t_], x[2], dw0[c][2]); dw0[c][3] = mfma16(a0[t_], x[3], dw0[c][3]); } while (0)
            CM_LDB(0, 0); CM_LDB(1, 1);
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                CM_DW0_STEP(0); CM_DW0_STEP(1); CM_DW0_STEP(2); CM_DW0_STEP(3); CM_DW0_STEP(4); CM_DW0_STEP(5); CM_DW0_STEP(6); CM_DW0_STEP(7);
                CM_DW0_STEP(8); CM_DW0_STEP(9); CM_DW0_STEP(10); CM_DW0_STEP(11); CM_DW0_STEP(12); CM_DW0_STEP(13); CM_DW0_STEP(14); CM_DW0_STEP(15);
            }
#undef CM_DW0_STEP
#undef CM_LDB
        }
        PH(7);
        __syncthreads();  // X tiles and dZ0 consumed: the next pair may overwrite them
        PH(8);
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
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int jt = 0; jt < 4; ++jt) {
                    const int k = 64 * c + 4 * n + jt;
                    if (k < din) out[off.W0 + (long)hu * din + k] = dw0[c][jt][q];
                }
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const int k = 4 * n + jt;
                if (k < H) out[off.Wl(0) + hu * H + k] = dw1[jt][q];
            }
        }
    }
    // column sums held per lane (column col, this lane group's rows): fold the four lane groups
    {
        float vals[4] = {db0, db1, dwo, 0.0f};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            __syncthreads();
            red[tid] = vals[k];
            __syncthreads();
            if (g == 0 && cok) {
                const float s = (red[64 * wave + n] + red[64 * wave + 16 + n]) + (red[64 * wave + 32 + n] + red[64 * wave + 48 + n]);
                if (k == 0) out[off.b0 + col] = s; else if (k == 1) out[off.bl(0) + col] = s; else out[off.Wout + col] = s;
            }
        }
    }
    // scalars held by the first lane of every quad in all four waves: dbout and the two statistics
    {
        const float s0 = cm_wave_sum(dbo), s1 = cm_wave_sum(st_vl), s2 = cm_wave_sum(st_cnt);
        __syncthreads();
        if (lane == 0) { red[wave] = s0; red[4 + wave] = s1; red[8 + wave] = s2; }
        __syncthreads();
        if (tid == 0) {
            out[off.bout] = (red[0] + red[1]) + (red[2] + red[3]);
#pragma unroll
            for (int k = 0; k < CM_NUM_STATS; ++k) out[off.P + k] = 0.0f;
            out[off.P + CM_STAT_VLOSS] = (red[4] + red[5]) + (red[6] + red[7]);
            out[off.P + CM_STAT_COUNT] = (red[8] + red[9]) + (red[10] + red[11]);
        }
    }
}

constexpr long CM_FUSED_MIN_ROWS = 131072;  // 8 row tiles per CU; below, the split schedule of cm_mlp_split.h is as fast or faster (docs/KERNEL_NOTES.md 3.1)
inline bool critic_fused_shape(const MlpArgs& a) {
    const int nc = (a.din + KC - 1) / KC;
    return nc >= 2 && nc <= 7 && a.H <= HP && a.L == 1 && a.dout == 1 && x_rows_vec(a) && !mfma_bf16x3();  // 8 chunks: 177 KB of LDS
}
inline size_t critic_fused_lds_bytes(int nc) { return (size_t)(nc * TM * LDT + 2 * TM * LDT + 5 * TM + 2 * NTHREADS) * sizeof(float); }
inline size_t critic_fused2_lds_bytes(int nc) { return (size_t)(2 * (nc * TM * LDT + 2 * TM * LDT + 5 * TM) + 2 * NTHREADS) * sizeof(float); }

// launches k_critic_fused + the partial-row reduction; same workspace layout as the fused k_mlp passes (MAX_GRID partial rows)
inline int run_critic_fused(MlpArgs a, float* grad_and_stats, void* ws, size_t ws_bytes, hipStream_t s, const char* who, const cm_opt_step_t* opt = nullptr) {
    const int64_t P = cm_mlp_param_count(a.din, a.H, a.L, a.dout);
    const size_t need = train_ws_bytes(a.din, a.H, a.L, a.dout);
    CM_REQUIRE(ws && ws_bytes >= need, "%s: workspace too small (%zu < %zu)", who, ws_bytes, need);
    a.partial = (float*)ws; a.PS = (int)((P + CM_NUM_STATS + 63) / 64 * 64);
#ifdef CM_PHASE_PROF
    a.prof = g_prof;
#endif
    const int nc = (a.din + KC - 1) / KC;
    const long ntiles = (a.rows + TM - 1) / TM;
    // critic_schedule = "fused2" (opt-in, round 6): two row tiles per iteration for two-chunk inputs (k_critic_fused2).  Measured at config 4
    // (5.2 M rows x 115 columns, profiles/r06_critic_two_tile_ab.txt): 3.08 ms against 2.93 ms for the one-tile kernel -- half the barriers per
    // row, but 410 registers (values shuttled between the two register files, 86 spilled scalars) and sequential hand-issued pipelines: not the default
    if (nc == 2 && cm_option(CM_OPTION_CRITIC_SCHEDULE) == 3 && ntiles >= 2) {
        const long npairs = (ntiles + 1) / 2;
        const int grid2 = (int)(npairs < 256 ? npairs : 256);
        const size_t lds2 = critic_fused2_lds_bytes(nc);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_critic_fused2<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
        hipLaunchKernelGGL(k_critic_fused2<2>, dim3(grid2), dim3(NTHREADS), lds2, s, a);
        CM_CHECK_LAUNCH(who);
        return finish_train(a, grid2, P, grad_and_stats, s, who, 0, opt);
    }
    const int grid = (int)(ntiles < 256 ? ntiles : 256);  // one workgroup per CU
    const size_t lds = critic_fused_lds_bytes(nc);
    // a.dz0 != NULL: the rows' layer-0 activations for THESE parameters are in memory (cm_critic_fwd_bwd_h0_ld): the kernel without the layer-0 product
#define CM_CF(NC_) do { \
        if (a.dz0) { \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_critic_fused<NC_, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL((k_critic_fused<NC_, true>), dim3(grid), dim3(NTHREADS), lds, s, a); \
        } else { \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_critic_fused<NC_>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
            hipLaunchKernelGGL(k_critic_fused<NC_>, dim3(grid), dim3(NTHREADS), lds, s, a); } } while (0)
    switch (nc) {
        case 2: CM_CF(2); break; case 3: CM_CF(3); break; case 4: CM_CF(4); break; case 5: CM_CF(5); break;
        case 6: CM_CF(6); break; default: CM_CF(7); break;
    }
#undef CM_CF
    CM_CHECK_LAUNCH(who);
    return finish_train(a, grid, P, grad_and_stats, s, who, 0, opt);
}

}  // namespace
