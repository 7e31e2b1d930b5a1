// cm_mlp_fused128.h -- ONE-launch forward / training pass of a 65..128-unit MLP with one hidden->hidden layer: the reference's COMA
// critic default (cleanmarl/coma_multienvs.py:35, critic_hidden_dim = 128) and every --actor_hidden_dim / --critic_hidden_dim up to
// 128 of the MAPPO / IPPO scripts (cleanmarl/mappo_multienvs.py:127-160).  Included by cm_mlp_wide.h, which falls back to its layered
// schedule for everything else (wider, deeper, no hidden->hidden layer).
//
// One workgroup of EIGHT waves per CU (two per SIMD, <= 256 registers each), 64-row tiles, v_mfma_f32_16x16x4_f32 (exact fp32 products,
// fp32 accumulation).  Wave w owns hidden units 16w .. 16w+15 of BOTH hidden layers for all 64 rows of a tile:
//   * its slices of W1 (forward B operand) and W1^T (backward B operand) live in REGISTERS for the whole launch; W0 and Wout are
//     zero-padded LDS images written once per launch -- no weight traffic per tile, the rest of LDS holds activations only (X double
//     buffered, H0, H1 -> dH1, logits -> dlogits);
//   * the C layout of a 16x16x4 product (lane (c, g) holds rows 4g .. 4g+3 of column c) IS the A layout of the transposed operand, so
//     the three weight-gradient products take dZ straight from the registers the backward products left it in (dW1 = dH1^T H0,
//     dW0 = dH0^T X, dWout = dOut^T H1 with H1 as the register-resident B operand) -- no transposed tiles, no extra LDS round trip;
//   * rows are contracted in the order {16j + 4g + i}: the scalar operand reads of those products walk DOWN a column of rows with the
//     four lane groups 16 banks apart (row strides 132 / 68 / 36 floats = 4 mod 64): conflict-free, as are the b128 row reads;
//   * weight-gradient accumulators (32 + 16 + 8 registers per lane) persist across the tiles of a workgroup and leave as ONE partial row
//     per workgroup, folded in a fixed order by k_reduce_partials (deterministic, same pattern as the 64-wide kernels).
// Five barriers per tile (layer 0 | layer 1 | head | loss | dH1).  The per-row loss heads are wide_loss_row<MODE> (cm_mlp_wide.h), one
// thread per row.  Inputs wider than 64 columns (MAPPO's central state) keep layer 0 OUTSIDE: z0 = X W0^T by k_wide_gemm, dW0 = dZ0^T X
// by k_dw0_stream -- the fused kernel then starts from z0 and ends at dZ0 (EXT0).  COMA's factored critic input (cm_coma.hip) enters as
// the z0 addend and leaves as dZ0 in the same way.
#pragma once

namespace {

constexpr int F_NT = 512;    // threads per workgroup (8 waves)
constexpr int F_TM = 64;     // rows per tile
constexpr int F_HP = 128;    // padded hidden width
constexpr int F_LDH = 132;   // LDS row stride of H0 / H1 / the head weight image
constexpr int F_LDX = 68;    // ... of the X tiles
constexpr int F_LDO = 36;    // ... of the logits / dlogits tile
constexpr int F_GRID = 256;  // persistent workgroups: one per CU
// scheduling fence between product groups: without it the compiler hoists every LDS operand read of a phase above its first MFMA
// (128 live registers in the dW1 phase alone) and spills the register-resident weights
#define F_FENCE() __builtin_amdgcn_sched_barrier(0)
// The 32 contraction indices lane group kq takes of a 128-wide row start at koff = {0, 64, 32, 96}[kq], NOT at 32 kq: a ds_read_b128 is
// served in four groups of 16 lanes, and each group mixes eight lanes of one kq with the eight COMPLEMENTARY row lanes of the next
// (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, ...).  With the row stride at 4 banks (132 floats) the 16 rows of a group sit on 16
// distinct 16-byte slots only if both kq of the group start on the same bank -- offsets 0 / 32 floats put them half a bank row apart and
// every operand read of a 128-wide row was a 2-way conflict (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.46 in the forward kernel).
// Any permutation of the contraction order is valid as long as both operands use it (the register-resident weights are loaded with it).
#define F128_KOFF const int koff = 64 * (kq & 1) + 32 * (kq >> 1);

struct F128X {
    const float* z0; long ldz0;  // optional layer-0 pre-activation addend [rows][ldz0] (EXT0: the whole X W0^T product)
    float* dz0; long lddz0;      // optional output: layer-0 pre-activation gradient
    float* y; long ldy; int ncols;  // M_FWD: head output (columns dout .. ncols-1 written as zeros)
    int vecx;                    // X rows are 16-byte aligned
};

__host__ __device__ inline int f128_lds_floats(bool ext0) {
    return (ext0 ? 0 : 2 * F_TM * F_LDX + F_HP * F_LDX) + 2 * F_TM * F_LDH + KMAX * F_LDH + F_TM * F_LDO;  // 163 328 bytes of the 160 KB with layer 0 inside
}

__device__ __forceinline__ float f128_kq_sum(float v) {  // sum over the four lane groups of a column (fixed order)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

template <int MODE, bool EXT0, int KCAP>
__global__ __launch_bounds__(F_NT) void k_mlp128(const MlpArgs a, const F128X e) {
    constexpr bool TRAIN = (MODE >= M_ACTOR);
    constexpr bool MASKED = (MODE == M_FWD || MODE == M_ACTOR || MODE == M_COMA_ACTOR);
    // the Q-critic's head (MSE on the taken action, cleanmarl/coma_multienvs.py:620-631) is element-wise in the logits tile: it runs in the
    // epilogue of the head product (no loss phase, one barrier less); the per-row inputs wait in the padding columns of the logits tile
    constexpr bool EPI_LOSS = (MODE == M_QCRITIC);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* W0s = smem + 2 * F_TM * F_LDX;  // [128][68] zero-padded image of W0 (!EXT0)
    float* H0s = smem + (EXT0 ? 0 : 2 * F_TM * F_LDX + F_HP * F_LDX);
    float* H1s = H0s + F_TM * F_LDH;
    float* Wos = H1s + F_TM * F_LDH;
    float* outs = Wos + KMAX * F_LDH;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lc = lane & 15, kq = lane >> 4;
    F128_KOFF
    const int H = a.H, din = a.din, K = a.dout;
    const Offsets off = make_offsets(din, H, 1, K);
    const float* __restrict__ P = a.params;
    const int n = 16 * w + lc;  // the hidden unit of this lane (both layers)
    const bool nok = n < H;
    const long ntiles = (a.rows + F_TM - 1) / F_TM;
    const float invA = 1.0f / (float)a.A;

    // ---- register-resident weights
    float w1f[32], w1b[32];
    if constexpr (!EXT0) {
        for (int i = tid; i < F_HP * 64; i += F_NT) {
            const int m = i >> 6, k = i & 63;
            W0s[m * F_LDX + k] = (m < H && k < din) ? P[off.W0 + m * din + k] : 0.0f;
        }
    }
#pragma unroll
    for (int s = 0; s < 32; ++s) {  // unconditional loads from clamped indices, then a select: the 32 requests are in flight together
        const int k = koff + s;
        const float v = P[off.Wl(0) + min(n, H - 1) * H + min(k, H - 1)];
        w1f[s] = (nok && k < H) ? v : 0.0f;
    }
    if constexpr (TRAIN) {
#pragma unroll
        for (int s = 0; s < 32; ++s) {
            const int m = koff + s;
            const float v = P[off.Wl(0) + min(m, H - 1) * H + min(n, H - 1)];
            w1b[s] = (nok && m < H) ? v : 0.0f;
        }
    }
    const float b0v = nok ? P[off.b0 + n] : 0.0f, b1v = nok ? P[off.bl(0) + n] : 0.0f;
    for (int i = tid; i < KMAX * F_HP; i += F_NT) {
        const int j = i >> 7, k = i & 127;
        Wos[j * F_LDH + k] = (j < K && k < H) ? P[off.Wout + j * H + k] : 0.0f;
    }
    const int hrb = w & 3, hcb = w >> 2;          // head: row block / column block of this wave
    const int hj = 16 * hcb + lc;                 // head output of this lane
    const float boutv = hj < K ? P[off.bout + hj] : 0.0f;

    // ---- X tile loader: thread -> (row tid / 8, columns 8 (tid % 8) .. + 7)
    const int xr = tid >> 3, xc = (tid & 7) * 8;
    float4 xa = make_float4(0.f, 0.f, 0.f, 0.f), xb = xa;
    auto load_x = [&](long tile) {
        xa = make_float4(0.f, 0.f, 0.f, 0.f); xb = xa;
        const long row = tile * F_TM + xr;
        if (tile < ntiles && row < a.rows) {
            const float* p = a.x + row * a.x_stride + xc;
            if (e.vecx) {
                if (xc < din) xa = *reinterpret_cast<const float4*>(p);
                if (xc + 4 < din) xb = *reinterpret_cast<const float4*>(p + 4);
            } else {
                if (xc < din) xa.x = p[0];
                if (xc + 1 < din) xa.y = p[1];
                if (xc + 2 < din) xa.z = p[2];
                if (xc + 3 < din) xa.w = p[3];
                if (xc + 4 < din) xb.x = p[4];
                if (xc + 5 < din) xb.y = p[5];
                if (xc + 6 < din) xb.z = p[6];
                if (xc + 7 < din) xb.w = p[7];
            }
            if (xc + 1 >= din) xa.y = 0.f;
            if (xc + 2 >= din) xa.z = 0.f;
            if (xc + 3 >= din) xa.w = 0.f;
            if (xc + 5 >= din) xb.y = 0.f;
            if (xc + 6 >= din) xb.z = 0.f;
            if (xc + 7 >= din) xb.w = 0.f;
        }
    };
    auto store_x = [&](int buf) {
        float* d = Xs + buf * (F_TM * F_LDX) + xr * F_LDX + xc;
        *reinterpret_cast<float4*>(d) = xa;
        *reinterpret_cast<float4*>(d + 4) = xb;
    };

    // ---- persistent accumulators (training)
    f32x4 accW1[8], accW0[4], accWo[2];
    float db0 = 0.f, db1 = 0.f, dbo[2] = {0.f, 0.f};
    float st[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (TRAIN) {
#pragma unroll
        for (int i = 0; i < 8; ++i) accW1[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i) accW0[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        accWo[0] = accWo[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // per-row inputs of the element-wise head (EPI_LOSS): threads 0..63 request the NEXT tile's {action, target, valid | first agent} at
    // the top of the backward phases and park them in the padding columns 32..34 of the logits tile at the end of the tile -- a whole
    // backward pass between request and use (requested at the tile top they cost wave 0 a memory round trip per tile)
    int rin_act = 0, rin_fl = 0;
    float rin_tg = 0.0f;
    auto load_rin = [&](long tile) {
        rin_act = 0; rin_fl = 0; rin_tg = 0.0f;
        const long row = tile * F_TM + tid;
        if (tid < F_TM && tile < ntiles && row < a.rows) {
            const unsigned r32 = (unsigned)row, seq = r32 / (unsigned)a.T;  // rows < 2^31 (check_rows)
            const int t = (int)(r32 - seq * (unsigned)a.T);
            const int en = (int)(seq / (unsigned)a.A), ag = (int)(seq - (unsigned)en * (unsigned)a.A);
            rin_act = a.action[row];
            rin_tg = a.ret[row];
            rin_fl = (t < a.ep_len[en] ? 1 : 0) | (ag == 0 ? 2 : 0);
        }
    };
    auto store_rin = [&]() {
        if (tid < F_TM) {
            float* rp = outs + tid * F_LDO + KMAX;  // no product reads these columns
            rp[0] = __int_as_float(rin_act); rp[1] = rin_tg; rp[2] = __int_as_float(rin_fl);
        }
    };
    if constexpr (EPI_LOSS) { load_rin(blockIdx.x); store_rin(); }
    if constexpr (!EXT0) { load_x(blockIdx.x); store_x(0); }
    __syncthreads();

    int buf = 0;
    PH_DECL
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x, buf ^= 1) {
        const long row0 = tile * F_TM;
        if constexpr (!EXT0 && !TRAIN) load_x(tile + gridDim.x);  // in flight under the forward products
        const float* Xc = Xs + buf * (F_TM * F_LDX);

        // ---- layer 0: H0 = relu(X W0^T + b0 + z0)
        f32x4 acc[4];
        float zv[4][4];
        unsigned mask0 = 0;
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // z0 is requested here and added behind the products (its round trip hides under them)
                const long row = row0 + 16 * rb + 4 * kq + i;
                zv[rb][i] = (e.z0 && nok && row < a.rows) ? e.z0[row * e.ldz0 + n] : 0.0f;
                acc[rb][i] = b0v;
            }
        if constexpr (!EXT0) {
            float4 bw[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bw[q] = *reinterpret_cast<const float4*>(W0s + n * F_LDX + 16 * kq + 4 * q);
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const float* ap = Xc + (16 * rb + lc) * F_LDX + 16 * kq;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 av = *reinterpret_cast<const float4*>(ap + 4 * q);
                    acc[rb] = mfma16(av.x, bw[q].x, acc[rb]);
                    acc[rb] = mfma16(av.y, bw[q].y, acc[rb]);
                    acc[rb] = mfma16(av.z, bw[q].z, acc[rb]);
                    acc[rb] = mfma16(av.w, bw[q].w, acc[rb]);
                }
                F_FENCE();
            }
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float h = fmaxf(acc[rb][i] + zv[rb][i], 0.0f);
                if (h > 0.0f) mask0 |= 1u << (4 * rb + i);
                H0s[(16 * rb + 4 * kq + i) * F_LDH + n] = h;
            }
        PH(0);
        __syncthreads();  // B1: H0 complete
        PH(1);

        if constexpr (!EXT0 && TRAIN) load_x(tile + gridDim.x);  // training: requested a whole layer before it is parked in LDS (behind the loss barrier)
        // ---- layer 1: H1 = relu(H0 W1^T + b1).  The row-block loops from here on are ROLLED (runtime rb): an unrolled phase lets the
        // compiler hoist every operand read of the phase above its first MFMA and spill the persistent accumulators to scratch
        // (measured: 127 spilled registers, dW1 / dW0 phases 30 % / 100 % slower); one row block per iteration bounds the live set.
#pragma unroll 2
        for (int rb = 0; rb < 4; ++rb) {
            f32x4 c1 = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* ap = H0s + (16 * rb + lc) * F_LDH + koff;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 av = *reinterpret_cast<const float4*>(ap + 4 * q);
                c1 = mfma16(av.x, w1f[4 * q], c1);
                c1 = mfma16(av.y, w1f[4 * q + 1], c1);
                c1 = mfma16(av.z, w1f[4 * q + 2], c1);
                c1 = mfma16(av.w, w1f[4 * q + 3], c1);
            }
            float* hp = H1s + (16 * rb + 4 * kq) * F_LDH + n;
#pragma unroll
            for (int i = 0; i < 4; ++i) hp[i * F_LDH] = fmaxf(c1[i] + b1v, 0.0f);
        }
        if constexpr (!EXT0 && !TRAIN) store_x(buf ^ 1);  // the next tile's X (its previous readers finished before B1)
        PH(2);
        __syncthreads();  // B2: H1 complete
        PH(3);

        // ---- head: wave (hrb, hcb) -> 16 rows x 16 outputs
        {
            f32x4 hacc = f32x4{0.f, 0.f, 0.f, 0.f};
            if (16 * hcb < K) {
                const float* ap = H1s + (16 * hrb + lc) * F_LDH + koff;
                const float* bp = Wos + (16 * hcb + lc) * F_LDH + koff;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 av = *reinterpret_cast<const float4*>(ap + 4 * q);
                    const float4 bv = *reinterpret_cast<const float4*>(bp + 4 * q);
                    hacc = mfma16(av.x, bv.x, hacc);
                    hacc = mfma16(av.y, bv.y, hacc);
                    hacc = mfma16(av.z, bv.z, hacc);
                    hacc = mfma16(av.w, bv.w, hacc);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int rl = 16 * hrb + 4 * kq + i;
                const long row = row0 + rl;
                float v = 0.0f;
                if (hj < K) {
                    v = hacc[i] + boutv;
                    if (MASKED && a.avail && row < a.rows && !a.avail[row * a.avail_stride + hj]) v = -1e9f;  // masked_fill(~avail, -1e9)
                }
                if constexpr (EPI_LOSS) {  // d(loss)/dq = 2/A (q_taken - target) on the taken action of a valid row, 0 elsewhere
                    const float* rp = outs + rl * F_LDO + KMAX;
                    const int act = __float_as_int(rp[0]), fl = __float_as_int(rp[2]);
                    const float df = v - rp[1];
                    const bool hit = (fl & 1) && hj == act && hj < K;
                    v = hit ? 2.0f * invA * df : 0.0f;
                    if (hit) st[4] += invA * df * df;
                    if ((fl & 3) == 3 && hj == 0) st[5] += 1.0f;
                }
                if constexpr (TRAIN) outs[rl * F_LDO + hj] = v;
                else if (row < a.rows && hj < e.ncols) e.y[row * e.ldy + hj] = v;
            }
        }
        PH(4);
        if constexpr (!TRAIN) continue;  // forward only: the next tile's barriers order every LDS reuse
        __syncthreads();  // B3: logits complete
        PH(5);

        if constexpr (!EPI_LOSS) {
            // ---- loss heads: one thread per row, logits -> dlogits in place
            if (tid < F_TM) {
                const long row = row0 + tid;
                float* z = outs + tid * F_LDO;
                if (row < a.rows) wide_loss_row<MODE, KCAP>(a, row, z, st);
                else for (int k = 0; k < K; ++k) z[k] = 0.0f;
            }
            PH(6);
            __syncthreads();  // B4: dlogits complete
            PH(7);
        }
        if constexpr (!EXT0) store_x(buf ^ 1);  // visible after B5; the buffer's last readers (dW0 of the previous tile) passed B1
        if constexpr (EPI_LOSS) load_rin(tile + gridDim.x);

        // ---- per row block: dWout += dOut^T H1 (B = this wave's own H1 columns, read back down the rows; dbout in wave 0),
        //      dH1 = (dOut Wout) .* relu'(H1) -> this wave's columns of the H1 tile, dW1 += dH1^T H0 (A = the registers dH1 was left in)
        {
            float woutb[8];  // Wout[8 g + s][n] down a column of the head weight image
#pragma unroll
            for (int s8 = 0; s8 < 8; ++s8) woutb[s8] = Wos[(8 * kq + s8) * F_LDH + n];
#pragma unroll 2
            for (int rb = 0; rb < 4; ++rb) {
                float* hp = H1s + (16 * rb + 4 * kq) * F_LDH + n;
                const float* op = outs + (16 * rb + 4 * kq) * F_LDO + lc;
                float h1v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) h1v[i] = hp[i * F_LDH];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    if (16 * cb < K) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float av = op[i * F_LDO + 16 * cb];
                            accWo[cb] = mfma16(av, h1v[i], accWo[cb]);
                            dbo[cb] += av;
                        }
                    }
                }
                f32x4 c2 = f32x4{0.f, 0.f, 0.f, 0.f};
                const float* ap = outs + (16 * rb + lc) * F_LDO + 8 * kq;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const float4 av = *reinterpret_cast<const float4*>(ap + 4 * q);
                    c2 = mfma16(av.x, woutb[4 * q], c2);
                    c2 = mfma16(av.y, woutb[4 * q + 1], c2);
                    c2 = mfma16(av.z, woutb[4 * q + 2], c2);
                    c2 = mfma16(av.w, woutb[4 * q + 3], c2);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d = h1v[i] > 0.0f ? c2[i] : 0.0f;
                    db1 += d;
                    hp[i * F_LDH] = d;
                }
            }
        }
        PH(8);
        // dW1 in its own loop (dH1 read back from the wave's own columns): 16 steps (row block, row) of 8 MFMAs on eight independent
        // accumulators, software-pipelined by hand -- the nine operand reads of step s + 1 are issued before the MFMAs of step s and the
        // fence keeps that order (left to itself the compiler either hoists all 144 reads and spills the accumulators, or -- in a rolled
        // loop -- issues 7 reads, consumes them, and exposes the LDS latency every 7 MFMAs: 17.1 k cycles per tile for 8.2 k of MFMA)
        {
            float bq[2][9];
            // an opaque zero keeps the 32 per-step LDS addresses from being hoisted out of the tile loop as 32 live registers
            // (row strides exceed the 8-bit offsets of ds_read2_b32): they are one v_add each, recomputed where they are used
            int zoff;
            asm volatile("v_mov_b32 %0, 0" : "=v"(zoff));
            const float* H0z = H0s + zoff + lc;
            const float* H1z = H1s + zoff + n;
            auto ld = [&](int st_, float (&q)[9]) {
                const int r = 16 * (st_ >> 2) + (st_ & 3) + 4 * kq;
                q[8] = H1z[r * F_LDH];
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) q[kb] = H0z[r * F_LDH + 16 * kb];  // columns >= H of H0 are zero
            };
            ld(0, bq[0]);
#pragma unroll
            for (int st_ = 0; st_ < 16; ++st_) {
                if (st_ + 1 < 16) ld(st_ + 1, bq[(st_ + 1) & 1]);
                F_FENCE();  // ... and the reads may not sink below the MFMAs either
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) accW1[kb] = mfma16(bq[st_ & 1][8], bq[st_ & 1][kb], accW1[kb]);
                F_FENCE();
            }
        }
        PH(9);
        __syncthreads();  // B5: dH1 complete
        PH(10);

        // ---- per row block: dH0 = (dH1 W1) .* relu'(H0) (-> dz0), dW0 += dH0^T X
#pragma unroll 2
        for (int rb = 0; rb < 4; ++rb) {
            f32x4 c3 = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* ap = H1s + (16 * rb + lc) * F_LDH + koff;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 av = *reinterpret_cast<const float4*>(ap + 4 * q);
                c3 = mfma16(av.x, w1b[4 * q], c3);
                c3 = mfma16(av.y, w1b[4 * q + 1], c3);
                c3 = mfma16(av.z, w1b[4 * q + 2], c3);
                c3 = mfma16(av.w, w1b[4 * q + 3], c3);
            }
            float dh[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                dh[i] = ((mask0 >> (4 * rb + i)) & 1u) ? c3[i] : 0.0f;
                db0 += dh[i];
                const long row = row0 + 16 * rb + 4 * kq + i;
                if (e.dz0 && nok && row < a.rows) e.dz0[row * e.lddz0 + n] = dh[i];
            }
            if constexpr (!EXT0) {
                const float* bp = Xc + (16 * rb + 4 * kq) * F_LDX + lc;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb)
                        accW0[kb] = mfma16(dh[i], bp[i * F_LDX + 16 * kb], accW0[kb]);  // columns >= din of X are zero
            }
        }
        if constexpr (EPI_LOSS) store_rin();  // this tile's head read its inputs before B3
        PH(12);
    }
    PH_FLUSH;

    if constexpr (TRAIN) {
        // ---- this workgroup's partial row: accumulator (i', lane (c, g)) = d W[16 w + 4 g + i'][16 kb + c]
        float* part = a.partial + (size_t)blockIdx.x * a.PS;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int no = 16 * w + 4 * kq + i;
            if (no < H) {
                if constexpr (!EXT0) {
#pragma unroll
                    for (int kb = 0; kb < 4; ++kb) { const int k = 16 * kb + lc; if (k < din) part[off.W0 + no * din + k] = accW0[kb][i]; }
                }
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) { const int k = 16 * kb + lc; if (k < H) part[off.Wl(0) + no * H + k] = accW1[kb][i]; }
            }
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) { const int j = 16 * cb + 4 * kq + i; if (j < K && nok) part[off.Wout + j * H + n] = accWo[cb][i]; }
        }
        db0 = f128_kq_sum(db0); db1 = f128_kq_sum(db1);
        dbo[0] = f128_kq_sum(dbo[0]); dbo[1] = f128_kq_sum(dbo[1]);
        if (kq == 0 && nok) { part[off.b0 + n] = db0; part[off.bl(0) + n] = db1; }
        if (w == 0 && kq == 0) {
            if (lc < K) part[off.bout + lc] = dbo[0];
            if (16 + lc < K) part[off.bout + 16 + lc] = dbo[1];
        }
        if constexpr (EPI_LOSS) {  // statistics: partial sums in the lanes of every head wave -> per-wave sums -> fixed-order fold
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 6; ++i) st[i] = cm_wave_sum(st[i]);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i) outs[w * 8 + i] = st[i];
            }
            __syncthreads();
            if (tid < CM_NUM_STATS) {
                float v = 0.0f;
                if (tid < 6) for (int ww = 0; ww < F_NT / 64; ++ww) v += outs[ww * 8 + tid];
                part[off.P + tid] = v;
            }
        } else if (w == 0) {  // statistics: rows were handled by the threads of wave 0
#pragma unroll
            for (int i = 0; i < 6; ++i) st[i] = cm_wave_sum(st[i]);
            if (lane < CM_NUM_STATS) {
                float v = 0.0f;
#pragma unroll
                for (int i = 0; i < 6; ++i) if (lane == i) v = st[i];
                part[off.P + lane] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Forward only (M_FWD), round 4.  The training kernel's forward (above, MODE = M_FWD: kept as wide_schedule=fused_r3 for A/B runs) spends
// 23 k cycles per 64-row tile for 13 k of MFMA: one workgroup per CU (160 KB of LDS), three phases with a barrier each, and a head phase
// that keeps four of the eight waves busy with one dependent chain of 32 products.  This kernel computes the TRANSPOSED products
// (A = the wave's weight rows, B = the activation rows): the result of a 16x16x4 product then leaves lane (row c, group g) holding the
// four units 16 w + 4 g + r of tile row c, and
//   * H0 goes to LDS as one b128 per lane, the z0 addend arrives as one 16-byte load per lane and row block;
//   * H1 never goes to LDS: the four registers of a layer-1 result ARE the B operand of the head's product over the wave's own 16 units
//     (A = Wout[j][16 w + 4 g' + r], four products per row block on every wave), and the eight waves' partial logits are summed in a
//     fixed order (wave 0..7, then the bias) by one thread per (row, output) after the second barrier;
//   * W0 lives in registers too (16 per lane), X needs ONE buffer (the next tile's X is parked after the first barrier), so a workgroup
//     needs 68 KB of LDS at K <= 8 and TWO workgroups share a CU (four waves per SIMD, <= 128 registers): one's barriers and LDS
//     round trips hide under the other's products.  Two barriers per tile.
__host__ __device__ inline int f128f_ks(int K, int ncols) { const int m = K > ncols ? K : ncols; return (m + 3) & ~3; }
__host__ __device__ inline int f128f_lds_floats(bool ext0, int ks) { return (ext0 ? 0 : F_TM * F_LDX) + F_TM * F_LDH + 8 * F_TM * ks + KMAX; }

template <bool EXT0>
__global__ __launch_bounds__(F_NT, 4) void k_mlp128_fwd(const MlpArgs a, const F128X e, const int KS) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* H0s = smem + (EXT0 ? 0 : F_TM * F_LDX);
    float* Ps = H0s + F_TM * F_LDH;       // [8 waves][64 rows][KS] partial logits
    float* bouts = Ps + 8 * F_TM * KS;    // [KMAX]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, lc = lane & 15, kq = lane >> 4;
    F128_KOFF
    const int H = a.H, din = a.din, K = a.dout;
    const Offsets off = make_offsets(din, H, 1, K);
    const float* __restrict__ P = a.params;
    const int n = 16 * w + lc;    // the unit whose weight rows this lane supplies (A operand)
    const bool nok = n < H;
    const int u0 = 16 * w + 4 * kq;  // the first of the four units this lane holds of a product's result
    const long ntiles = (a.rows + F_TM - 1) / F_TM;

    float w1f[32], w0r[16], b0r[4], b1r[4], woA[2][4];
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const int k = koff + s;
        const float v = P[off.Wl(0) + min(n, H - 1) * H + min(k, H - 1)];
        w1f[s] = (nok && k < H) ? v : 0.0f;
    }
    if constexpr (!EXT0) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int k = 16 * kq + s;
            const float v = P[off.W0 + min(n, H - 1) * din + min(k, din - 1)];
            w0r[s] = (nok && k < din) ? v : 0.0f;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int u = u0 + r;
        b0r[r] = u < H ? P[off.b0 + u] : 0.0f;
        b1r[r] = u < H ? P[off.bl(0) + u] : 0.0f;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int j = 16 * cb + lc;
            woA[cb][r] = (j < K && u < H) ? P[off.Wout + min(j, K - 1) * H + min(u, H - 1)] : 0.0f;
        }
    }
    if (tid < KMAX) bouts[tid] = tid < K ? P[off.bout + tid] : 0.0f;
    const bool vecz = e.z0 && e.ldz0 >= F_HP && (e.ldz0 & 3) == 0 && (reinterpret_cast<uintptr_t>(e.z0) & 15) == 0;

    // ---- X tile loader: thread -> (row tid / 8, columns 8 (tid % 8) .. + 7)
    const int xr = tid >> 3, xc = (tid & 7) * 8;
    float4 xa = make_float4(0.f, 0.f, 0.f, 0.f), xb = xa;
    auto load_x = [&](long tile) {
        xa = make_float4(0.f, 0.f, 0.f, 0.f); xb = xa;
        const long row = tile * F_TM + xr;
        if (tile < ntiles && row < a.rows) {
            const float* p = a.x + row * a.x_stride + xc;
            if (e.vecx) {
                if (xc < din) xa = *reinterpret_cast<const float4*>(p);
                if (xc + 4 < din) xb = *reinterpret_cast<const float4*>(p + 4);
            } else {
                if (xc < din) xa.x = p[0];
                if (xc + 1 < din) xa.y = p[1];
                if (xc + 2 < din) xa.z = p[2];
                if (xc + 3 < din) xa.w = p[3];
                if (xc + 4 < din) xb.x = p[4];
                if (xc + 5 < din) xb.y = p[5];
                if (xc + 6 < din) xb.z = p[6];
                if (xc + 7 < din) xb.w = p[7];
            }
            if (xc + 1 >= din) xa.y = 0.f;
            if (xc + 2 >= din) xa.z = 0.f;
            if (xc + 3 >= din) xa.w = 0.f;
            if (xc + 5 >= din) xb.y = 0.f;
            if (xc + 6 >= din) xb.z = 0.f;
            if (xc + 7 >= din) xb.w = 0.f;
        }
    };
    auto store_x = [&]() {
        float* d = Xs + xr * F_LDX + xc;
        *reinterpret_cast<float4*>(d) = xa;
        *reinterpret_cast<float4*>(d + 4) = xb;
    };
    // z0 addend of (row 16 rb + c, units u0 .. u0 + 3): requested one tile ahead (under layer 1), consumed as the accumulators' initial value
    float4 zq[4];
    auto load_z = [&](long tile) {
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            const long row = tile * F_TM + 16 * rb + lc;
            zq[rb] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e.z0 && tile < ntiles && row < a.rows) {
                const float* zp = e.z0 + row * e.ldz0 + u0;
                if (vecz) zq[rb] = *reinterpret_cast<const float4*>(zp);
                else {
                    if (u0 < H) zq[rb].x = zp[0];
                    if (u0 + 1 < H) zq[rb].y = zp[1];
                    if (u0 + 2 < H) zq[rb].z = zp[2];
                    if (u0 + 3 < H) zq[rb].w = zp[3];
                }
            }
        }
    };
    load_z(blockIdx.x);
    if constexpr (!EXT0) { load_x(blockIdx.x); store_x(); }
    __syncthreads();

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * F_TM;
        if constexpr (!EXT0) load_x(tile + gridDim.x);  // in flight under layer 0, parked behind its barrier

        // ---- layer 0 (transposed): lane (row c, g) <- units u0 .. u0 + 3 of row 16 rb + c; the accumulators start at b0 + z0 (requested a tile ahead)
        f32x4 acc[4];
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb] = f32x4{b0r[0] + zq[rb].x, b0r[1] + zq[rb].y, b0r[2] + zq[rb].z, b0r[3] + zq[rb].w};
        if constexpr (!EXT0) {
#pragma unroll
            for (int rb = 0; rb < 4; ++rb) {
                const float* bp = Xs + (16 * rb + lc) * F_LDX + 16 * kq;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bv = *reinterpret_cast<const float4*>(bp + 4 * q);
                    acc[rb] = mfma16(w0r[4 * q], bv.x, acc[rb]);
                    acc[rb] = mfma16(w0r[4 * q + 1], bv.y, acc[rb]);
                    acc[rb] = mfma16(w0r[4 * q + 2], bv.z, acc[rb]);
                    acc[rb] = mfma16(w0r[4 * q + 3], bv.w, acc[rb]);
                }
            }
        }
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) {
            float4 h;
            h.x = (u0 < H) ? fmaxf(acc[rb][0], 0.0f) : 0.0f;
            h.y = (u0 + 1 < H) ? fmaxf(acc[rb][1], 0.0f) : 0.0f;
            h.z = (u0 + 2 < H) ? fmaxf(acc[rb][2], 0.0f) : 0.0f;
            h.w = (u0 + 3 < H) ? fmaxf(acc[rb][3], 0.0f) : 0.0f;
            *reinterpret_cast<float4*>(H0s + (16 * rb + lc) * F_LDH + u0) = h;
        }
        __syncthreads();  // B1: H0 complete; X and the partial-logit slabs free

        if constexpr (!EXT0) store_x();
        load_z(tile + gridDim.x);
        // ---- layer 1 (transposed) + this wave's share of the head, per row block
#pragma unroll 2
        for (int rb = 0; rb < 4; ++rb) {
            f32x4 c1 = f32x4{b1r[0], b1r[1], b1r[2], b1r[3]};
            const float* bp = H0s + (16 * rb + lc) * F_LDH + koff;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 bv = *reinterpret_cast<const float4*>(bp + 4 * q);
                c1 = mfma16(w1f[4 * q], bv.x, c1);
                c1 = mfma16(w1f[4 * q + 1], bv.y, c1);
                c1 = mfma16(w1f[4 * q + 2], bv.z, c1);
                c1 = mfma16(w1f[4 * q + 3], bv.w, c1);
            }
            float h1[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) h1[r] = fmaxf(c1[r], 0.0f);  // units >= H: zero weights and bias
            float* pp = Ps + ((w * F_TM + 16 * rb + lc) * KS + 4 * kq);
            f32x4 hq = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) hq = mfma16(woA[0][r], h1[r], hq);  // lane (row c, g) <- outputs 4 g .. 4 g + 3
            if (4 * kq < KS) *reinterpret_cast<float4*>(pp) = make_float4(hq[0], hq[1], hq[2], hq[3]);
            if (K > 16) {
                f32x4 hr = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 4; ++r) hr = mfma16(woA[1][r], h1[r], hr);
                if (16 + 4 * kq < KS) *reinterpret_cast<float4*>(pp + 16) = make_float4(hr[0], hr[1], hr[2], hr[3]);
            }
        }
        __syncthreads();  // B2: partial logits complete

        // ---- logits: one thread per (row, output), the eight partial sums in wave order, then the bias
        for (int o = tid; o < F_TM * KS; o += F_NT) {
            const int rl = o / KS, j = o - rl * KS;
            const long row = row0 + rl;
            float v = 0.0f;
            if (j < K) {
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) v += Ps[(ww * F_TM + rl) * KS + j];
                v += bouts[j];
                if (a.avail && row < a.rows && !a.avail[row * a.avail_stride + j]) v = -1e9f;  // masked_fill(~avail, -1e9)
            }
            if (row < a.rows && j < e.ncols) e.y[row * e.ldy + j] = v;
        }
    }
}

inline void fused128_fwd_launch(bool ext0, const MlpArgs& a, const F128X& e, hipStream_t s) {
    const int ks = f128f_ks(a.dout, e.ncols);
    const size_t lds = (size_t)f128f_lds_floats(ext0, ks) * sizeof(float);
    const long nt = (a.rows + F_TM - 1) / F_TM;
    const int per_cu = 2 * lds <= 160 * 1024 ? 2 : 1;
    const int grid = (int)(nt < per_cu * F_GRID ? nt : per_cu * F_GRID);
    if (ext0) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp128_fwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_mlp128_fwd<true>), dim3(grid), dim3(F_NT), lds, s, a, e, ks);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp128_fwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((k_mlp128_fwd<false>), dim3(grid), dim3(F_NT), lds, s, a, e, ks);
    }
}

// the shapes this kernel serves (cm_set_option("wide_schedule", "layered") keeps everything on the layered schedule: A/B runs, tests)
inline bool fused128_shape(int H, int L) { return H > HP && H <= F_HP && L == 1 && cm_option(CM_OPTION_WIDE_SCHEDULE) != 2; }
inline int fused128_grid(long rows) { const long nt = (rows + F_TM - 1) / F_TM; return (int)(nt < F_GRID ? nt : F_GRID); }
inline size_t fused128_part_floats(long rows, int din, int H, int dout) {
    const size_t PS = (size_t)((cm_mlp_param_count(din, H, 1, dout) + CM_NUM_STATS + 63) / 64 * 64);
    return (size_t)fused128_grid(rows) * PS;
}

template <int MODE, bool EXT0, int KCAP>
inline void fused128_launch_k(const MlpArgs& a, const F128X& e, hipStream_t s) {
    const size_t lds = (size_t)f128_lds_floats(EXT0) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp128<MODE, EXT0, KCAP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_mlp128<MODE, EXT0, KCAP>), dim3(fused128_grid(a.rows)), dim3(F_NT), lds, s, a, e);
}
template <int MODE, bool EXT0>
inline void fused128_launch(const MlpArgs& a, const F128X& e, hipStream_t s) {
    if constexpr (MODE == M_FWD) { if (cm_option(CM_OPTION_WIDE_SCHEDULE) != 3) { fused128_fwd_launch(EXT0, a, e, s); return; } }
    // the actor heads exist in a K <= 8 form (the softmax / gradient loops of wide_loss_row over 8 instead of 32 outputs)
    if constexpr (MODE == M_ACTOR || MODE == M_COMA_ACTOR) { if (a.dout <= 8) { fused128_launch_k<MODE, EXT0, 8>(a, e, s); return; } }
    fused128_launch_k<MODE, EXT0, KMAX>(a, e, s);
}

}  // namespace
