// cm_gru_step2.h -- the recurrent GRU step of the second-generation kernels (included by cm_gru.hip inside its anonymous namespace,
// after sigmoidf_ / tanhf_; used by k_gru2_fwd in cm_gru_v2.h and by the fused rollout k_gru32_rollout).
#pragma once

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for this wave's outstanding global stores
// (s_waitcnt vmcnt(0)); the per-step workspace stores are never read inside the step loops, and waiting for their
// acknowledgement at the next barrier cost ~1400 cycles per step (profiles/r02_phase_gru_v2.txt, "barrier_top").
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---------------------------------------------------------------------------------------------------------------------------
// The recurrent step, "column-quarter" form: wave w owns hidden columns 16w .. 16w+15 of ALL three gates for the tile's 32 rows
// (two 16-row blocks on v_mfma_f32_16x16x4_f32).  r, z, n and h' of a (row, column) are then produced by ONE lane from its own
// accumulators -- no hand-over of gates between waves, two LDS barriers per step (x1 complete, h' complete) -- and the matrix work is
// the same on every wave (fc1 + six 32 x 16 x 64 products = 6.7 k cycles, the balanced minimum for a 32-row tile on one CU).
// The six gate blocks + fc1 are 112 registers per lane: w[4j + i] = W[(16w + n) * ld + 16j + 4g + i] for lane (n = lane & 15,
// g = lane >> 4), the B operand of MFMA (j, i); the A operand is one 16-byte LDS read per four MFMAs and is shared by the three
// products that multiply the same activations.
struct G2W { float w1[16], xr[16], xz[16], xn[16], hr[16], hz[16], hn[16]; float b1, br, bz, bin, bhn; };
template <bool VEC>
__device__ __forceinline__ void load_nt16_regs(float (&w)[16], const float* W, int c0, int nrows, int ld, int ncols) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const int c = c0 + n;
    const float* p = W + (long)c * ld + 4 * g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (VEC) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (c < nrows && 16 * j + 4 * g < ncols) v = *reinterpret_cast<const float4*>(p + 16 * j);
            w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) w[4 * j + i] = (c < nrows && 16 * j + 4 * g + i < ncols) ? p[16 * j + i] : 0.0f;
        }
    }
}
template <bool WV>
__device__ __forceinline__ void g2_load_weights(G2W& w, const float* params, const GruOff& off, int din, int H) {
    const int wave = threadIdx.x >> 6, c0 = 16 * wave, c = c0 + (threadIdx.x & 15);
    load_nt16_regs<false>(w.w1, params + off.W1, c0, H, din, din);
    load_nt16_regs<WV>(w.xr, params + off.Wih, c0, H, H, H);
    load_nt16_regs<WV>(w.xz, params + off.Wih + H * H, c0, H, H, H);
    load_nt16_regs<WV>(w.xn, params + off.Wih + 2 * H * H, c0, H, H, H);
    load_nt16_regs<WV>(w.hr, params + off.Whh, c0, H, H, H);
    load_nt16_regs<WV>(w.hz, params + off.Whh + H * H, c0, H, H, H);
    load_nt16_regs<WV>(w.hn, params + off.Whh + 2 * H * H, c0, H, H, H);
    const bool ok = c < H;
    w.b1 = ok ? params[off.b1 + c] : 0.0f;
    w.br = ok ? params[off.bih + c] + params[off.bhh + c] : 0.0f;            // same association as the first-generation kernels
    w.bz = ok ? params[off.bih + H + c] + params[off.bhh + H + c] : 0.0f;
    w.bin = ok ? params[off.bih + 2 * H + c] : 0.0f;
    w.bhn = ok ? params[off.bhh + 2 * H + c] : 0.0f;
}
// The two halves of g2_load_weights for the split forward sweep (k_gru2_pre / k_gru2_fwdx<.., PRE>): the blocks that multiply the
// observation and x1 (h-independent: fc1, W_ir, W_iz, W_in) and the blocks that multiply h_{t-1} (the recurrence: W_hr, W_hz, W_hn).
// The biases keep g2_load_weights' association (b_ih + b_hh for r and z).
template <bool WV>
__device__ __forceinline__ void g2_load_weights_x(G2W& w, const float* params, const GruOff& off, int din, int H) {
    const int wave = threadIdx.x >> 6, c0 = 16 * wave, c = c0 + (threadIdx.x & 15);
    load_nt16_regs<false>(w.w1, params + off.W1, c0, H, din, din);
    load_nt16_regs<WV>(w.xr, params + off.Wih, c0, H, H, H);
    load_nt16_regs<WV>(w.xz, params + off.Wih + H * H, c0, H, H, H);
    load_nt16_regs<WV>(w.xn, params + off.Wih + 2 * H * H, c0, H, H, H);
    w.b1 = (c < H) ? params[off.b1 + c] : 0.0f;
}
template <bool WV>
__device__ __forceinline__ void g2_load_weights_h(G2W& w, const float* params, const GruOff& off, int H) {
    const int wave = threadIdx.x >> 6, c0 = 16 * wave, c = c0 + (threadIdx.x & 15);
    load_nt16_regs<WV>(w.hr, params + off.Whh, c0, H, H, H);
    load_nt16_regs<WV>(w.hz, params + off.Whh + H * H, c0, H, H, H);
    load_nt16_regs<WV>(w.hn, params + off.Whh + 2 * H * H, c0, H, H, H);
    const bool ok = c < H;
    w.br = ok ? params[off.bih + c] + params[off.bhh + c] : 0.0f;
    w.bz = ok ? params[off.bih + H + c] + params[off.bhh + H + c] : 0.0f;
    w.bin = ok ? params[off.bih + 2 * H + c] : 0.0f;
    w.bhn = ok ? params[off.bhh + 2 * H + c] : 0.0f;
}
// three products that share their A operand: acc_p[rb] += A[16 rb + ..][16 kb] * W_p^T, p = 0..2, rb = 0..1 (six independent chains).
// The sweeps run ONE wave per SIMD, so the A-operand reads of k block j + 1 are issued (inline asm, cm_common.h) BEFORE the 24 MFMAs of
// block j: left to the compiler they followed them, and every block started with an exposed LDS round trip (~100 of its 770 cycles).
#define CM_G2_LD(j_) do { xa[(j_) & 1] = cf_lds128<16 * (j_) * 4>(ab); ya[(j_) & 1] = cf_lds128<(16 * LDT + 16 * (j_)) * 4>(ab); } while (0)
__device__ __forceinline__ void g2_prod3(f32x4 (&a0)[2], f32x4 (&a1)[2], f32x4 (&a2)[2], const float* As,
                                         const float (&w0)[16], const float (&w1)[16], const float (&w2)[16]) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const unsigned ab = cf_lds_addr(As + n * LDT + 4 * g);
    f32x4 xa[2], ya[2];
#define CM_G2_K(j_, i) do { \
        a0[0] = mfma16(x[i], w0[4 * (j_) + i], a0[0]); a1[0] = mfma16(x[i], w1[4 * (j_) + i], a1[0]); a2[0] = mfma16(x[i], w2[4 * (j_) + i], a2[0]); \
        a0[1] = mfma16(y[i], w0[4 * (j_) + i], a0[1]); a1[1] = mfma16(y[i], w1[4 * (j_) + i], a1[1]); a2[1] = mfma16(y[i], w2[4 * (j_) + i], a2[1]); } while (0)
#define CM_G2_BLOCK(j_, NEXT_) do { NEXT_; const f32x4 x = xa[(j_) & 1], y = ya[(j_) & 1]; \
        CM_G2_K(j_, 0); CM_G2_K(j_, 1); CM_G2_K(j_, 2); CM_G2_K(j_, 3); } while (0)
    CM_G2_LD(0);
    CM_G2_BLOCK(0, CM_G2_LD(1); cf_wait<2>(xa[0], ya[0]));
    CM_G2_BLOCK(1, CM_G2_LD(2); cf_wait<2>(xa[1], ya[1]));
    CM_G2_BLOCK(2, CM_G2_LD(3); cf_wait<2>(xa[0], ya[0]));
    CM_G2_BLOCK(3, cf_wait<0>(xa[1], ya[1]));
#undef CM_G2_BLOCK
#undef CM_G2_K
}
__device__ __forceinline__ void g2_prod1(f32x4 (&a0)[2], const float* As, const float (&w0)[16], int kb16) {
    const int lane = threadIdx.x & 63, n = lane & 15, g = lane >> 4;
    const unsigned ab = cf_lds_addr(As + n * LDT + 4 * g);
    f32x4 xa[2], ya[2];
#define CM_G2_BLOCK1(j_, NEXT_) do { NEXT_; const f32x4 x = xa[(j_) & 1], y = ya[(j_) & 1]; \
        a0[0] = mfma16(x[0], w0[4 * (j_)], a0[0]);     a0[1] = mfma16(y[0], w0[4 * (j_)], a0[1]); \
        a0[0] = mfma16(x[1], w0[4 * (j_) + 1], a0[0]); a0[1] = mfma16(y[1], w0[4 * (j_) + 1], a0[1]); \
        a0[0] = mfma16(x[2], w0[4 * (j_) + 2], a0[0]); a0[1] = mfma16(y[2], w0[4 * (j_) + 2], a0[1]); \
        a0[0] = mfma16(x[3], w0[4 * (j_) + 3], a0[0]); a0[1] = mfma16(y[3], w0[4 * (j_) + 3], a0[1]); } while (0)
    // kb16 is uniform: blocks past it are skipped (their weights are zero anyway).  Every case is ONE straight-line pipeline with its
    // first read inside: a read issued ahead of a branch would be copied by the compiler (phi) while still in flight -- the linter
    // tools/lint_lds_hazards.py checks the generated code for exactly that
    switch (kb16) {
        case 1:
            CM_G2_LD(0); CM_G2_BLOCK1(0, cf_wait<0>(xa[0], ya[0]));
            break;
        case 2:
            CM_G2_LD(0); CM_G2_BLOCK1(0, CM_G2_LD(1); cf_wait<2>(xa[0], ya[0])); CM_G2_BLOCK1(1, cf_wait<0>(xa[1], ya[1]));
            break;
        case 3:
            CM_G2_LD(0); CM_G2_BLOCK1(0, CM_G2_LD(1); cf_wait<2>(xa[0], ya[0])); CM_G2_BLOCK1(1, CM_G2_LD(2); cf_wait<2>(xa[1], ya[1]));
            CM_G2_BLOCK1(2, cf_wait<0>(xa[0], ya[0]));
            break;
        default:
            CM_G2_LD(0); CM_G2_BLOCK1(0, CM_G2_LD(1); cf_wait<2>(xa[0], ya[0])); CM_G2_BLOCK1(1, CM_G2_LD(2); cf_wait<2>(xa[1], ya[1]));
            CM_G2_BLOCK1(2, CM_G2_LD(3); cf_wait<2>(xa[0], ya[0])); CM_G2_BLOCK1(3, cf_wait<0>(xa[1], ya[1]));
            break;
    }
#undef CM_G2_BLOCK1
}
#undef CM_G2_LD
// One step.  In: X0 = obs tile [32][LDT] (zero padded to 64 columns), hp = h_{t-1}.  Out: X1 = x1, hn = h'; with SAVE also the
// tiles r, z, n, W_hn h + b_hn (SR, SZ, SN, SG) that the backward sweep needs.  Barriers: the caller's barrier BEFORE the call must
// cover X0 / hp (and the previous readers of X1 / S* / hn); the function ends with the barrier that completes hn and the tiles.
template <bool SAVE>
__device__ __forceinline__ void gru2_step(const G2W& w, const float* X0, float* X1, const float* hp, float* hn,
                                          float* SR, float* SZ, float* SN, float* SG, int din, int H) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const int col = 16 * wave + n;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 a1[2] = {zero, zero};
    g2_prod1(a1, X0, w.w1, (din + 15) >> 4);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int q = 0; q < 4; ++q) X1[(16 * rb + 4 * g + q) * LDT + col] = fmaxf(a1[rb][q] + w.b1, 0.0f);
    f32x4 hr[2] = {zero, zero}, hz[2] = {zero, zero}, hnn[2] = {zero, zero};
    g2_prod3(hr, hz, hnn, hp, w.hr, w.hz, w.hn);   // does not depend on x1: in flight while the other waves finish theirs
    lds_barrier();                                  // x1 complete
    f32x4 xr[2] = {zero, zero}, xz[2] = {zero, zero}, xn[2] = {zero, zero};
    g2_prod3(xr, xz, xn, X1, w.xr, w.xz, w.xn);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = (16 * rb + 4 * g + q) * LDT + col;
            const float r = sigmoidf_((xr[rb][q] + hr[rb][q]) + w.br);
            const float z = sigmoidf_((xz[rb][q] + hz[rb][q]) + w.bz);
            const float ghn = hnn[rb][q] + w.bhn;
            const float nn = tanhf_(xn[rb][q] + w.bin + r * ghn);
            hn[o] = (col < H) ? (1.0f - z) * nn + z * hp[o] : 0.0f;
            if (SAVE) { SR[o] = r; SZ[o] = z; SN[o] = nn; SG[o] = ghn; }
        }
    lds_barrier();                                  // h' (and the saved tiles) complete
}

// ---------------------------------------------------------------------------------------------------------------------------
// The step SPLIT at its dependence on h (round 6; cleanmarl/mappo_lstm_multienvs.py:170-176: x1 = relu(fc1(obs_t)) and gi = W_ih x1 + b_ih
// do not depend on h_{t-1}, only gh = W_hh h_{t-1} does).  gru2_step runs fc1 + six 32 x 16 x 64 products = 6.7 k MFMA cycles per step on
// the chain; four of those seven products are h-independent and the whole chunk's observations are known when the sweep starts.
//   * gru2_pre_unit: x1 and the three W_ih products of ONE (tile, step) -- the first half of gru2_step, same MFMA sequences on the same
//     operands, so the accumulators are the bits gru2_step would have held -- for a throughput launch over all (tile, step) units of the
//     chunk on ALL compute units (k_gru2_pre).  The accumulators leave in lane order (one 16-byte store per lane and 16-row block: fully
//     coalesced) and come back into the same lanes of the chain.
//   * gru2_step_h: the rest of the step with those accumulators as inputs: three W_hh products + the gate math; ONE barrier per step
//     (the caller's, at the top of its loop) instead of three, 3.07 k MFMA cycles instead of 6.9 k.
// GI_UNIT floats per (tile, step): [wave 4][k = 2 gate + rb : 6][lane 64][4].
constexpr int GI_UNIT = 4 * 6 * 64 * 4;
__device__ __forceinline__ void gru2_pre_unit(const G2W& w, const float* X0, float* X1, float* gi_unit, int din) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const int col = 16 * wave + n;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 a1[2] = {zero, zero};
    g2_prod1(a1, X0, w.w1, (din + 15) >> 4);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int q = 0; q < 4; ++q) X1[(16 * rb + 4 * g + q) * LDT + col] = fmaxf(a1[rb][q] + w.b1, 0.0f);
    lds_barrier();                                  // x1 complete
    f32x4 xr[2] = {zero, zero}, xz[2] = {zero, zero}, xn[2] = {zero, zero};
    g2_prod3(xr, xz, xn, X1, w.xr, w.xz, w.xn);
    f32x4* out = reinterpret_cast<f32x4*>(gi_unit + (size_t)wave * (6 * 64 * 4)) + lane;
    out[0] = xr[0]; out[64] = xr[1]; out[128] = xz[0]; out[192] = xz[1]; out[256] = xn[0]; out[320] = xn[1];
}
__device__ __forceinline__ void gru2_gi_load(f32x4 (&gi)[6], const float* gi_unit) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f32x4* in = reinterpret_cast<const f32x4*>(gi_unit + (size_t)wave * (6 * 64 * 4)) + lane;
#pragma unroll
    for (int k = 0; k < 6; ++k) gi[k] = in[64 * k];
}
// hp = h_{t-1} (complete: the caller's barrier), gi = this lane's W_ih accumulators of the step.  Writes hn = h' and, with SAVE, the tiles
// the backward sweep needs; NO trailing barrier (the caller's next loop-top barrier completes them).
template <bool SAVE>
__device__ __forceinline__ void gru2_step_h(const G2W& w, const f32x4 (&gi)[6], const float* hp, float* hn,
                                            float* SR, float* SZ, float* SN, float* SG, int H) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, g = lane >> 4;
    const int col = 16 * wave + n;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 hr[2] = {zero, zero}, hz[2] = {zero, zero}, hnn[2] = {zero, zero};
    g2_prod3(hr, hz, hnn, hp, w.hr, w.hz, w.hn);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = (16 * rb + 4 * g + q) * LDT + col;
            const float r = sigmoidf_((gi[rb][q] + hr[rb][q]) + w.br);
            const float z = sigmoidf_((gi[2 + rb][q] + hz[rb][q]) + w.bz);
            const float ghn = hnn[rb][q] + w.bhn;
            const float nn = tanhf_(gi[4 + rb][q] + w.bin + r * ghn);
            hn[o] = (col < H) ? (1.0f - z) * nn + z * hp[o] : 0.0f;
            if (SAVE) { SR[o] = r; SZ[o] = z; SN[o] = nn; SG[o] = ghn; }
        }
}
