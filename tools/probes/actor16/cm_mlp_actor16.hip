// cm_mlp_actor16.hip -- wave-private variant of the PPO actor fwd+bwd pass (cm_ppo_actor_fwd_bwd), opt-in: CM_ACTOR_KERNEL=wave16.
//
// k_mlp (cm_mlp_kernel.h) splits a 64-row tile 2 x 2 over four waves and synchronises them ~12 times per tile.  Here every wave
// owns a 16-row tile END TO END: v_mfma_f32_16x16x4_f32 for every product, X straight from HBM into A-operand registers, the
// wave's activations in a private LDS slice, its own full dW0 / dW1 / dWout accumulators (144 registers) -- no workgroup
// barrier in the tile loop, so the eight waves of the CU's single workgroup drift apart and fill each other's VALU / LDS phases;
// the weights are shared in LDS once per CU.  Same arithmetic (exact fp32 products, fp32 accumulation; the order of the k / row
// sums differs from k_mlp, as k_mlp's differs from the reference's).  Shapes: din <= 64, H <= 64, exactly one hidden->hidden
// layer, <= 16 actions, 16-byte aligned rows; everything else stays on k_mlp.  docs/KERNEL_NOTES.md section 10 item 0 has the measurements
// behind it (tools/probes/wave_private_probe.hip is the skeleton this grew from).
//
// MFMA 16x16x4 operand / result layout (lane l: r = l & 15, g = l >> 4): A[m = r][k = g], B[k = g][n = r], D register q holds
// D[m = 4 g + q][n = r].  Any k permutation is a valid contraction order if both operands use it: for the 64-wide products lane
// group g takes k = 16 g + s at step s (16 contiguous floats per lane), for the row contractions k = row = 4 g + s
// (conflict-free LDS banks for both: 4 * 68 = 16 mod 64).
#include "cm_mlp_kernel.h"
#ifndef A16_ABL
#define A16_ABL 0
#endif
#include "cm_mlp_actor16.h"

namespace {

constexpr int A16_NW = 8, A16_NT = A16_NW * 64;  // one workgroup of 8 waves per CU
constexpr int A16_LD = 68, A16_DL = 20;          // row strides: 64-wide tiles (conflict-free b128 down rows), dlogits
constexpr int A16_BUF = 16 * A16_LD;
constexpr int A16_PRIV = 3 * A16_BUF + 16 * A16_DL;  // X | H0 | H1 | dlogits per wave
constexpr int A16_W1 = HP * A16_LD + 64;             // W1 image: row n at n * A16_LD + 16 * (n >> 4) (see w1i)
constexpr int A16_LDS_FLOATS = HP * A16_LD + A16_W1 + 16 * A16_LD + 16 + A16_NW * A16_PRIV;  // W0 | W1 | Wout | bout | private slices

// W1 is read down its rows by the dZ0 product (lane group g: rows 16 g + s): a 16-float shift per 16-row block puts the four groups
// on different banks (row stride 68 alone: 16 * 68 = 0 mod 64)
__device__ __forceinline__ constexpr int w1i(int n) { return n * A16_LD + 16 * (n >> 4); }

#define A16_WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)

__device__ __forceinline__ void a16_ld16(float (&d)[16], const float* p) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
        d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
    }
}

// all-reduce over the 16 lanes of a DPP row (= one lane group g): rotations by 8, 4, 2, 1
template <int N> __device__ __forceinline__ float a16_row_ror(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x120 + N, 0xF, 0xF, false));
}
__device__ __forceinline__ float a16_row_sum(float v) {
    v += a16_row_ror<8>(v); v += a16_row_ror<4>(v); v += a16_row_ror<2>(v); v += a16_row_ror<1>(v);
    return v;
}
// all-reduce over the four lane groups (lanes r, r + 16, r + 32, r + 48)
#ifdef A16_BPERMUTE
__device__ __forceinline__ float a16_grp_sum(float v) { v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64); return v; }
__device__ __forceinline__ float a16_grp_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64)); return v; }
#else
// gfx950 row swaps (VALU, no LDS round trip): v_permlane16_swap exchanges the odd 16-lane rows of its first operand with the even rows
// of its second, v_permlane32_swap the upper half of the first with the lower half of the second; fed the same value twice, the two
// results hold the two partners of every lane
template <class F> __device__ __forceinline__ float a16_grp_reduce(float v, F f) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    auto p = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    v = f(__builtin_bit_cast(float, (unsigned)p[0]), __builtin_bit_cast(float, (unsigned)p[1]));
    const unsigned w = __builtin_bit_cast(unsigned, v);
    auto q = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return f(__builtin_bit_cast(float, (unsigned)q[0]), __builtin_bit_cast(float, (unsigned)q[1]));
}
__device__ __forceinline__ float a16_grp_sum(float v) { return a16_grp_reduce(v, [](float x, float y) { return x + y; }); }
__device__ __forceinline__ float a16_grp_max(float v) { return a16_grp_reduce(v, [](float x, float y) { return fmaxf(x, y); }); }
#endif

// X rows of a tile as the A operand: lane (r, g) holds X[row0 + r][16 g .. 16 g + 15] (zero beyond din / rows)
__device__ __forceinline__ void a16_load_x(float (&xa)[16], const float* __restrict__ x, long x_stride, long rows, int din, long row0, int r, int g) {
    const long row = row0 + r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = 16 * g + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows && c < din) v = *reinterpret_cast<const float4*>(x + row * x_stride + c);  // din % 4 == 0
        xa[4 * q] = v.x; xa[4 * q + 1] = v.y; xa[4 * q + 2] = v.z; xa[4 * q + 3] = v.w;
    }
}



__global__ __launch_bounds__(A16_NT, 1) void k_actor16(const A16Args a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W0s = smem;
    float* W1s = W0s + HP * A16_LD;
    float* Wos = W1s + A16_W1;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), r = lane & 15, g = lane >> 4;
    float* bos = Wos + 16 * A16_LD;
    float* Xb = bos + 16 + wave * A16_PRIV;
    float* H0 = Xb + A16_BUF;
    float* H1 = H0 + A16_BUF;
    float* dl = H1 + A16_BUF;
    const Offsets off = make_offsets(a.din, a.H, 1, a.dout);
    const int H = a.H, din = a.din, K = a.dout;
    for (int i = tid; i < HP * HP; i += A16_NT) {
        const int n = i / HP, c = i % HP;
        W0s[n * A16_LD + c] = (n < H && c < din) ? a.params[off.W0 + n * din + c] : 0.0f;
        W1s[w1i(n) + c] = (n < H && c < H) ? a.params[off.Wl(0) + n * H + c] : 0.0f;
    }
    for (int i = tid; i < 16 * HP; i += A16_NT) {
        const int k = i / HP, c = i % HP;
        Wos[k * A16_LD + c] = (k < K && c < H) ? a.params[off.Wout + k * H + c] : 0.0f;
    }
    if (tid < HP) {
        W0s[tid * A16_LD + 64] = tid < H ? a.params[off.b0 + tid] : 0.0f;
        W1s[w1i(tid) + 64] = tid < H ? a.params[off.bl(0) + tid] : 0.0f;
        if (tid < 16) bos[tid] = tid < K ? a.params[off.bout + tid] : 0.0f;
    }
    __syncthreads();  // the only workgroup barrier before the final reduction
    // biases live in the 4 padding columns of the weight images: b0[c] at W0s row c col 64, b1[c] at W1s row c col 64

    f32x4 aW0[4][4], aW1[4][4], aWo[4];
    float db0[4], db1[4], dbo[4];
    float st_acc = 0.f, st_cnt = 0.f;  // st_acc: pg | entropy | kl | clip fraction in lane groups 0..3
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aWo[i] = f32x4{0, 0, 0, 0}; db0[i] = db1[i] = dbo[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { aW0[i][j] = f32x4{0, 0, 0, 0}; aW1[i][j] = f32x4{0, 0, 0, 0}; }
    }
    const float invA = 1.0f / (float)a.A;
    const int rows = (int)a.rows;  // the host guarantees rows < 2^31
    const int ntiles = (rows + 15) / 16;
    const int stride = (int)gridDim.x * A16_NW;
    int tile = (int)blockIdx.x * A16_NW + wave;
    for (; tile < ntiles; tile += stride) {
        const int row0 = tile * 16;
        float xa[16];
        a16_load_x(xa, a.x, a.x_stride, a.rows, din, (long)row0, r, g);
        // ---- this lane's row (row0 + r): loss inputs, requested now and consumed by the head two layers later (ahead of the next
        //      tile's X prefetch in issue order); rows past the end read the last row and are masked by valid_r
#if A16_ABL == 4
        const int grow = r;
#else
        const int grow = min(row0 + r, rows - 1);
#endif
        const int seq_r = grow / a.T;
        const int e_r = seq_r / a.A;
        const bool first_r = seq_r - e_r * a.A == 0;
        const bool valid_r = row0 + r < rows && grow - seq_r * a.T < a.ep_len[e_r];
        const int act_r = a.action[grow];
        const float lpo_r = a.logp_old[grow], adv_r = a.adv[grow];
        unsigned avm = 0u;  // avail bytes of outputs 4 g .. 4 g + 3 (index clamped: padded outputs are dropped by the k < K tests)
#pragma unroll
        for (int q = 0; q < 4; ++q) avm |= (unsigned)a.avail[(long)grow * a.avail_stride + min(4 * g + q, K - 1)] << (8 * q);
        // ---- X: A operand in registers, row-major copy in the private LDS slice (B operand of dW0)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<float4*>(Xb + r * A16_LD + 16 * g + 4 * q) = make_float4(xa[4 * q], xa[4 * q + 1], xa[4 * q + 2], xa[4 * q + 3]);
        // ---- forward layer 0
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float b[16];
            a16_ld16(b, W0s + (16 * j + r) * A16_LD + 16 * g);
            f32x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = mfma16(xa[s], b[s], acc);
            const float bj = W0s[(16 * j + r) * A16_LD + 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) H0[(4 * g + i) * A16_LD + 16 * j + r] = fmaxf(acc[i] + bj, 0.0f);
        }
        A16_WAVE_SYNC();
        // ---- forward layer 1
        {
            float x1[16];
            a16_ld16(x1, H0 + r * A16_LD + 16 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float b[16];
                a16_ld16(b, W1s + w1i(16 * j + r) + 16 * g);
                f32x4 acc = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = mfma16(x1[s], b[s], acc);
                const float bj = W1s[w1i(16 * j + r) + 64];
#pragma unroll
                for (int i = 0; i < 4; ++i) H1[(4 * g + i) * A16_LD + 16 * j + r] = fmaxf(acc[i] + bj, 0.0f);
            }
        }
        A16_WAVE_SYNC();
        // ---- head, TRANSPOSED: D = Wout H1^T, so lane (r, g) holds logits k = 4 g + q (q = 0..3) of ITS row r: per-row inputs are
        //      one row per lane, the softmax reductions cross the four lane groups (2 steps), and dlogits stay in registers as the
        //      A operand of dZ1.  Categorical statistics + clipped surrogate as in k_mlp's epilogue (cleanmarl/mappo_multienvs.py:527-570).
        float d[4];
        {
            float x2[16], b[16];
            a16_ld16(x2, H1 + r * A16_LD + 16 * g);
            a16_ld16(b, Wos + r * A16_LD + 16 * g);
            f32x4 z = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 16; ++s) z = mfma16(b[s], x2[s], z);
#if A16_ABL == 1
#pragma unroll
            for (int q = 0; q < 4; ++q) { d[q] = valid_r ? z[q] * 1e-3f * adv_r + lpo_r + (float)(act_r + (int)avm + (int)first_r) : 0.0f; dbo[q] += d[q]; }
            *reinterpret_cast<float4*>(dl + r * A16_DL + 4 * g) = make_float4(d[0], d[1], d[2], d[3]);
#else
            const float4 bo4 = *reinterpret_cast<const float4*>(bos + 4 * g);
            const float bo[4] = {bo4.x, bo4.y, bo4.z, bo4.w};
            float zz[4], e[4], lp[4];
            float m = -INFINITY;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool mine = 4 * g + q < K;
                zz[q] = mine ? (((avm >> (8 * q)) & 0xffu) ? z[q] + bo[q] : -1e9f) : -INFINITY;  // masked_fill(~avail, -1e9); padded outputs drop out
                m = fmaxf(m, zz[q]);
            }
            m = a16_grp_max(m);
            float ssum = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) { e[q] = expf(zz[q] - m); ssum += e[q]; }  // exp(-inf) = 0 for the padded outputs
            ssum = a16_grp_sum(ssum);
            const float lse = m + logf(ssum);
            float entp = 0.0f, lpap = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool mine = 4 * g + q < K;
                lp[q] = mine ? zz[q] - lse : 0.0f;
                e[q] = e[q] / ssum;  // p
                entp -= e[q] * lp[q];
                lpap += (4 * g + q == act_r) ? lp[q] : 0.0f;
            }
            const float ent = a16_grp_sum(entp);
            const float lpa = a16_grp_sum(lpap);
            const float log_ratio = lpa - lpo_r;
            const float ratio = expf(log_ratio);
            const float pg1 = adv_r * ratio;
            const float pg2 = adv_r * fminf(fmaxf(ratio, a.clip_lo), a.clip_hi);
            const bool inr = (ratio >= a.clip_lo) && (ratio <= a.clip_hi);
            float gg;  // d min(pg1, pg2) / d ratio with torch's tie rule
            if (pg1 < pg2) gg = adv_r;
            else if (pg1 > pg2) gg = inr ? adv_r : 0.0f;
            else gg = 0.5f * adv_r + (inr ? 0.5f * adv_r : 0.0f);
            {   // all four lanes of a row hold the same row values: lane group g accumulates statistic g (one register), group 0 also the count
                const float sv = g == 0 ? fminf(pg1, pg2) : g == 1 ? ent : g == 2 ? (ratio - 1.0f) - log_ratio : (fabsf(ratio - 1.0f) > a.clip_eps ? 1.0f : 0.0f);
                st_acc += valid_r ? invA * sv : 0.0f;
                st_cnt += (valid_r && first_r && g == 0) ? 1.0f : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float dq = invA * (-(gg * ratio) * ((4 * g + q == act_r ? 1.0f : 0.0f) - e[q]) + a.ent_coef * e[q] * (lp[q] + ent));
                if (4 * g + q >= K || !valid_r || zz[q] <= -5e8f) dq = 0.0f;  // padded rows / outputs; masked_fill blocks the gradient
                dbo[q] += dq;
                d[q] = dq;
            }
            *reinterpret_cast<float4*>(dl + r * A16_DL + 4 * g) = make_float4(d[0], d[1], d[2], d[3]);
#endif
        }
        A16_WAVE_SYNC();
        // ---- dWout += dl^T H1 (contraction over the 16 rows) ; dZ1 = (dl Wout) .* relu'(H1)
        {
            float al[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) al[s] = dl[(4 * g + s) * A16_DL + r];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) aWo[j] = mfma16(al[s], H1[(4 * g + s) * A16_LD + 16 * j + r], aWo[j]);
            f32x4 dz[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dz[j] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 4; ++s) dz[j] = mfma16(d[s], Wos[(4 * g + s) * A16_LD + 16 * j + r], dz[j]);
            }
            A16_WAVE_SYNC();  // every lane has read H1 as the dWout operand before it becomes dZ1
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float* p = H1 + (4 * g + i) * A16_LD + 16 * j + r;
                    const float v = *p > 0.0f ? dz[j][i] : 0.0f;
                    db1[j] += v;
                    *p = v;
                }
        }
        A16_WAVE_SYNC();
        // ---- dW1 += dZ1^T H0 ; dZ0 = (dZ1 W1) .* relu'(H0)
        {
            float az[4][4], bh[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) { az[t][s] = H1[(4 * g + s) * A16_LD + 16 * t + r]; bh[t][s] = H0[(4 * g + s) * A16_LD + 16 * t + r]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int s = 0; s < 4; ++s) aW1[i][j] = mfma16(az[i][s], bh[j][s], aW1[i][j]);
            float zr[16];
            a16_ld16(zr, H1 + r * A16_LD + 16 * g);
            f32x4 dz[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dz[j] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 16; ++s) dz[j] = mfma16(zr[s], W1s[w1i(16 * g + s) + 16 * j + r], dz[j]);
            }
            A16_WAVE_SYNC();
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float* p = H0 + (4 * g + i) * A16_LD + 16 * j + r;
                    const float v = *p > 0.0f ? dz[j][i] : 0.0f;
                    db0[j] += v;
                    *p = v;
                }
        }
        A16_WAVE_SYNC();
        {   // next tile's X pulled into L2 (one line per lane, value dropped): the tile-top loads then hit L2.  Holding the
            // tile itself in registers across the backward phases costs 16 of the 256 and makes the allocator spill accumulators.
            const int pr = (tile + stride) * 16 + (lane >> 2), pc = (lane & 3) * 32;
            if (pr < rows && pc < din) (void)*reinterpret_cast<const volatile float*>(a.x + (long)pr * a.x_stride + pc);
        }
        // ---- dW0 += dZ0^T X
        {
            float az[4][4], bx[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) { az[t][s] = H0[(4 * g + s) * A16_LD + 16 * t + r]; bx[t][s] = Xb[(4 * g + s) * A16_LD + 16 * t + r]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int s = 0; s < 4; ++s) aW0[i][j] = mfma16(az[i][s], bx[j][s], aW0[i][j]);
        }
        A16_WAVE_SYNC();
    }
    // ---- workgroup partial = sum over the 8 waves, in wave order (deterministic), assembled in LDS in the parameter layout
    __syncthreads();
    float* R = smem;  // [PS] floats: weights and private slices are dead
    for (int i = tid; i < a.PS; i += A16_NT) R[i] = 0.0f;
    // the four lane groups hold partial column sums of the same bias entries / statistics: fold them first
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        db0[j] += __shfl_xor(db0[j], 16, 64); db0[j] += __shfl_xor(db0[j], 32, 64);
        db1[j] += __shfl_xor(db1[j], 16, 64); db1[j] += __shfl_xor(db1[j], 32, 64);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) dbo[q] = a16_row_sum(dbo[q]);  // lane (r, g) held output 4 g + q of row r: fold the 16 rows
    st_acc = a16_row_sum(st_acc);  // per lane group: its statistic summed over the 16 rows
    const float sv[5] = {__shfl(st_acc, 0, 64), __shfl(st_acc, 16, 64), __shfl(st_acc, 32, 64), __shfl(st_acc, 48, 64), cm_wave_sum(st_cnt)};
    for (int w = 0; w < A16_NW; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = 16 * i + 4 * g + q, c = 16 * j + r;
                        if (n < H && c < din) R[off.W0 + n * din + c] += aW0[i][j][q];
                        if (n < H && c < H) R[off.Wl(0) + n * H + c] += aW1[i][j][q];
                    }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int k = 4 * g + q, c = 16 * i + r;
                    if (k < K && c < H) R[off.Wout + k * H + c] += aWo[i][q];
                }
                if (g == 0 && 16 * i + r < H) { R[off.b0 + 16 * i + r] += db0[i]; R[off.bl(0) + 16 * i + r] += db1[i]; }
            }
            if (r == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (4 * g + q < K) R[off.bout + 4 * g + q] += dbo[q];
            }
            if (lane == 0) {
                R[off.P + CM_STAT_PG] += sv[0]; R[off.P + CM_STAT_ENT] += sv[1]; R[off.P + CM_STAT_KL] += sv[2];
                R[off.P + CM_STAT_CLIP] += sv[3]; R[off.P + CM_STAT_COUNT] += sv[4];
            }
        }
    }
    __syncthreads();
    float* out = a.partial + (size_t)blockIdx.x * a.PS;
    for (int i = tid; i < a.PS; i += A16_NT) out[i] = R[i];
}

}  // namespace

bool cm_actor16_enabled() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("CM_ACTOR_KERNEL"); v = (e && strcmp(e, "wave16") == 0) ? 1 : 0; }
    return v == 1;
}
bool cm_actor16_supports(int din, int H, int L, int dout, bool rows_16B_aligned, int PS) {
    return L == 1 && H <= HP && din <= 64 && dout <= 16 && rows_16B_aligned && PS <= A16_LDS_FLOATS;
}
int cm_actor16_launch(const A16Args& a, hipStream_t s) {
    const size_t lds = (size_t)A16_LDS_FLOATS * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_actor16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const long ntiles = (a.rows + 15) / 16;
    const int grid = (int)min(256L, (ntiles + A16_NW - 1) / A16_NW);
    hipLaunchKernelGGL(k_actor16, dim3(grid), dim3(A16_NT), lds, s, a);
    return grid;
}
