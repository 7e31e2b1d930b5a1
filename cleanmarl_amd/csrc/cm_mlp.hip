// cm_mlp.hip -- fused actor / critic MLP kernels for gfx950 (fp32 MFMA, weights LDS-stationary).
//
// One kernel template covers the four program regions of cleanmarl/mappo_multienvs.py that run an MLP:
//   M_FWD    Actor.logits / Critic.forward         (:178-183, :197-200)   -> cm_mlp_forward
//   M_ACT    Actor.act (sample + log_prob)         (:172-176, :409-414)   -> cm_policy_act
//   M_ACTOR  PPO clipped-surrogate fwd + bwd       (:527-551, :561-582)   -> cm_ppo_actor_fwd_bwd
//   M_CRITIC value MSE fwd + bwd                   (:554-558, :582)       -> cm_critic_fwd_bwd
//
// Design: a workgroup = 4 wavefronts (2 along rows x 2 along hidden columns) owns a tile of
// TM = 64 rows and walks the whole network for that tile with every activation resident in LDS; rows
// never round-trip to HBM between layers or between forward and backward.  All GEMMs run on
// v_mfma_f32_32x32x2_f32 (exact fp32, == an fmaf chain), so results stay within fp32 round-off of the
// reference's CPU PyTorch run.  Three MFMA forms are used, each wave owning one 32x32 output tile:
//   rowpar_nt : Y[64 x 64]  = A[64 x K] * W[64 x K]^T        (forward layers)
//   rowpar_tn : dX[64 x 64] = dZ[64 x 64] * W[64 x 64]       (backward data path)
//   colred    : dW[64 x 64] += dZ[64 rows x 64]^T * X[64 rows x 64]  (weight gradients; the accumulators
//               stay in registers across ALL row tiles of the persistent workgroup and are written once)
// The tiny head (K <= 32 outputs) and all softmax / PPO / MSE math run on the VALU out of LDS.
// Hidden widths H <= 64 are zero-padded to 64 in LDS (dead units have zero activations and zero
// gradients), the input width is processed in chunks of 64 columns.
#include "cm_common.h"

namespace {

constexpr int HP = 64;    // padded hidden width
constexpr int TM = 64;    // rows per tile
constexpr int KC = 64;    // input chunk width
constexpr int LDT = 68;   // LDS row stride in floats (4*17: conflict-free ds_read_b128 down a column of rows)
constexpr int LMAX = 2;   // max hidden->hidden layers
constexpr int KMAX = 32;  // max head width
constexpr int DWO = KMAX * HP / 256;
constexpr int NTHREADS = 256;

enum Mode { M_FWD = 0, M_ACT = 1, M_ACTOR = 2, M_CRITIC = 3 };

struct MlpArgs {
    const float* x; long x_stride; long rows;
    int din, H, L, dout;
    const float* params;
    // M_FWD
    const uint8_t* avail; long avail_stride; float* y;
    // M_ACT
    unsigned long long seed; long row_offset; int t; int* action_out; float* logp_out; long out_stride;
    // training
    const int* action; const float* logp_old; const float* adv; const float* ret; const int* ep_len;
    int A, T, per_agent;
    float clip_lo, clip_hi, clip_eps, ent_coef;
    float* partial; int PS;  // per-workgroup partial gradients + stats, row stride PS floats
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
    __host__ __device__ int Hs(int l) const { return Hs0 + l * TM * LDT; }
    __host__ __device__ int bl(int l) const { return bl0 + l * HP; }
};
__host__ __device__ inline Lds make_lds(int L, int dout) {
    Lds s; int p = 0;
    s.Xs = p; p += TM * LDT;
    s.W0s = p; p += HP * LDT;
    s.Hs0 = p; p += (L + 1) * TM * LDT;
    s.Ws = p; if (L > 0) p += HP * LDT;
    s.wout = p; p += dout * HP;
    s.b0 = p; p += HP;
    s.bl0 = p; p += L * HP;
    s.bout = p; p += KMAX;
    s.ls = p; p += TM * (dout + 1);
    p = (p + 3) & ~3;
    s.red = p; p += 4 * HP;
    s.total = p;
    return s;
}

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// acc[32x32] += A[32 rows][8*kb] * B[32 rows(n)][8*kb]^T ; A,B row-major in LDS with stride LDT.
// k is consumed in the permuted order {8j+i, 8j+4+i}: lane half h reads floats [8j+4h, 8j+4h+4) as one b128.
__device__ __forceinline__ void rowpar_nt(f32x16& acc, const float* As, const float* Bs, int kb) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float* ap = As + r * LDT + 4 * h;
    const float* bp = Bs + r * LDT + 4 * h;
#pragma unroll 2
    for (int j = 0; j < kb; ++j) {
        const float4 a = *reinterpret_cast<const float4*>(ap + 8 * j);
        const float4 b = *reinterpret_cast<const float4*>(bp + 8 * j);
        acc = mfma32(a.x, b.x, acc);
        acc = mfma32(a.y, b.y, acc);
        acc = mfma32(a.z, b.z, acc);
        acc = mfma32(a.w, b.w, acc);
    }
}

// acc[32x32] += dZ[32 rows][64 (n)] * W[64 (n)][32 cols]  (W row-major [n][k] in LDS, read transposed)
__device__ __forceinline__ void rowpar_tn(f32x16& acc, const float* As, const float* Ws_c0) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float* ap = As + r * LDT + 4 * h;
    const float* bp = Ws_c0 + (4 * h) * LDT + r;
#pragma unroll 2
    for (int j = 0; j < HP / 8; ++j) {
        const float4 a = *reinterpret_cast<const float4*>(ap + 8 * j);
        const float b0 = bp[(8 * j + 0) * LDT], b1 = bp[(8 * j + 1) * LDT];
        const float b2 = bp[(8 * j + 2) * LDT], b3 = bp[(8 * j + 3) * LDT];
        acc = mfma32(a.x, b0, acc);
        acc = mfma32(a.y, b1, acc);
        acc = mfma32(a.z, b2, acc);
        acc = mfma32(a.w, b3, acc);
    }
}

// acc[32 (n) x 32 (k)] += sum_rows dZ[row][n0 + i] * X[row][k0 + j]   over the TM rows of the tile
__device__ __forceinline__ void colred(f32x16& acc, const float* Zs_n0, const float* Xs_k0) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float* ap = Zs_n0 + h * LDT + r;
    const float* bp = Xs_k0 + h * LDT + r;
#pragma unroll 4
    for (int kk = 0; kk < TM / 2; ++kk) {
        acc = mfma32(ap[2 * kk * LDT], bp[2 * kk * LDT], acc);
    }
}

__device__ __forceinline__ void stage_rows(float* dst, const float* src, long row0, long nrows, long stride,
                                           int col0, int ncols) {
    // dst[r][k] = src[(row0+r)*stride + col0 + k]  for r < TM, k < KC; zero outside [nrows) x [ncols)
#pragma unroll 4
    for (int i = threadIdx.x; i < TM * KC; i += NTHREADS) {
        const int r = i >> 6, k = i & 63;
        const long row = row0 + r;
        float v = 0.0f;
        if (row < nrows && k < ncols) v = src[row * stride + col0 + k];
        dst[r * LDT + k] = v;
    }
}

template <int NCH, int MODE>
__global__ __launch_bounds__(NTHREADS) void k_mlp(const MlpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr bool TRAIN = (MODE == M_ACTOR || MODE == M_CRITIC);
    const Offsets off = make_offsets(a.din, a.H, a.L, a.dout);
    const Lds lds = make_lds(a.L, a.dout);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, lc = lane & 31;
    const int H = a.H, L = a.L, dout = a.dout, din = a.din;
    const int nch = (din + KC - 1) / KC;
    const bool w0_resident = (nch == 1);
    const bool ws_resident = (L == 1);
    float* Xs = smem + lds.Xs;
    float* W0s = smem + lds.W0s;
    float* Ws = smem + lds.Ws;
    float* wouts = smem + lds.wout;
    float* ls = smem + lds.ls;
    float* red = smem + lds.red;
    const int lstride = dout + 1;

    // ---- one-time staging of small tensors (+ resident weights)
    for (int i = tid; i < dout * HP; i += NTHREADS) {
        const int k = i / HP, c = i % HP;
        wouts[i] = (c < H) ? a.params[off.Wout + k * H + c] : 0.0f;
    }
    for (int i = tid; i < HP; i += NTHREADS) {
        smem[lds.b0 + i] = (i < H) ? a.params[off.b0 + i] : 0.0f;
        for (int l = 0; l < LMAX; ++l)
            if (l < L) smem[lds.bl(l) + i] = (i < H) ? a.params[off.bl(l) + i] : 0.0f;
    }
    for (int i = tid; i < KMAX; i += NTHREADS) smem[lds.bout + i] = (i < dout) ? a.params[off.bout + i] : 0.0f;
    if (w0_resident) stage_rows(W0s, a.params + off.W0, 0, H, din, 0, din);
    if (ws_resident) stage_rows(Ws, a.params + off.Wl(0), 0, H, H, 0, H);

    // ---- persistent accumulators (training)
    f32x16 accW0[NCH > 0 ? NCH : 1];
    f32x16 accWl[LMAX];
    float dwo[DWO];
    float dbh[LMAX + 1];
    float dbo = 0.0f;
    float st_pg = 0.f, st_ent = 0.f, st_kl = 0.f, st_clip = 0.f, st_vl = 0.f, st_cnt = 0.f;
    if (TRAIN) {
#pragma unroll
        for (int c = 0; c < (NCH > 0 ? NCH : 1); ++c)
#pragma unroll
            for (int g = 0; g < 16; ++g) accW0[c][g] = 0.0f;
#pragma unroll
        for (int l = 0; l < LMAX; ++l)
#pragma unroll
            for (int g = 0; g < 16; ++g) accWl[l][g] = 0.0f;
#pragma unroll
        for (int j = 0; j < DWO; ++j) dwo[j] = 0.0f;
#pragma unroll
        for (int l = 0; l <= LMAX; ++l) dbh[l] = 0.0f;
    }

    const long ntiles = (a.rows + TM - 1) / TM;
    const int hrow = tid >> 2, hq = tid & 3;  // head mapping: 4 lanes per row, 16 hidden columns per lane

    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * TM;
        // ================= forward, layer 0 (input chunks) =================
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
        for (int c = 0; c < nch; ++c) {
            __syncthreads();  // previous readers of Xs / W0s are done
            const int w = min(KC, din - c * KC);
            stage_rows(Xs, a.x, row0, a.rows, a.x_stride, c * KC, w);
            if (!w0_resident) stage_rows(W0s, a.params + off.W0, 0, H, din, c * KC, w);
            __syncthreads();
            rowpar_nt(acc, Xs + 32 * wm * LDT, W0s + 32 * wn * LDT, (w + 7) >> 3);
        }
        {
            float* H0 = smem + lds.Hs(0);
            const float bias = smem[lds.b0 + 32 * wn + lc];
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                H0[row * LDT + 32 * wn + lc] = fmaxf(acc[g] + bias, 0.0f);
            }
        }
        __syncthreads();
        // ================= forward, hidden layers =================
#pragma unroll
        for (int l = 1; l <= LMAX; ++l) {
            if (l <= L) {
                if (!ws_resident) {
                    stage_rows(Ws, a.params + off.Wl(l - 1), 0, H, H, 0, H);
                    __syncthreads();
                }
#pragma unroll
                for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
                rowpar_nt(acc, smem + lds.Hs(l - 1) + 32 * wm * LDT, Ws + 32 * wn * LDT, HP / 8);
                float* Hl = smem + lds.Hs(l);
                const float bias = smem[lds.bl(l - 1) + 32 * wn + lc];
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                    Hl[row * LDT + 32 * wn + lc] = fmaxf(acc[g] + bias, 0.0f);
                }
                __syncthreads();
            }
        }
        // ================= head forward (VALU) =================
        float* HL = smem + lds.Hs(L);
        float hreg[16];
        {
            const float4* hp4 = reinterpret_cast<const float4*>(HL + hrow * LDT + 16 * hq);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 v = hp4[i];
                hreg[4 * i] = v.x; hreg[4 * i + 1] = v.y; hreg[4 * i + 2] = v.z; hreg[4 * i + 3] = v.w;
            }
        }
        const long grow = row0 + hrow;  // this lane-group's global row
        const bool rvalid = grow < a.rows;
        for (int k = 0; k < dout; ++k) {
            const float4* wp4 = reinterpret_cast<const float4*>(wouts + k * HP + 16 * hq);
            float p = 0.0f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 w4 = wp4[i];
                p = fmaf(hreg[4 * i], w4.x, p); p = fmaf(hreg[4 * i + 1], w4.y, p);
                p = fmaf(hreg[4 * i + 2], w4.z, p); p = fmaf(hreg[4 * i + 3], w4.w, p);
            }
            p += __shfl_xor(p, 1, 64);
            p += __shfl_xor(p, 2, 64);
            if (hq == (k & 3)) {
                float v = p + smem[lds.bout + k];
                if (MODE == M_FWD) {
                    if (rvalid) {
                        if (a.avail && !a.avail[grow * a.avail_stride + k]) v = -1e9f;
                        a.y[grow * dout + k] = v;
                    }
                } else if (MODE == M_ACT) {
                    if (rvalid && a.avail && !a.avail[grow * a.avail_stride + k]) v = -1e9f;
                    ls[hrow * lstride + k] = v;
                } else if (MODE == M_ACTOR) {
                    if (rvalid && !a.avail[grow * (long)dout + k]) v = -1e9f;
                    ls[hrow * lstride + k] = v;
                } else {
                    ls[hrow * lstride + k] = v;
                }
            }
        }
        if (MODE == M_FWD) continue;  // next tile (the loop-top barrier protects LDS reuse)
        __syncthreads();

        // ================= per-row head math: lane hq == 0 of every row =================
        if (MODE == M_ACT) {
            if (hq == 0 && rvalid) {
                float* z = ls + hrow * lstride;
                float m = -INFINITY;
                for (int k = 0; k < dout; ++k) m = fmaxf(m, z[k]);
                float s = 0.0f;
                for (int k = 0; k < dout; ++k) s += expf(z[k] - m);
                const float lse = m + logf(s);
                const unsigned long long gr = (unsigned long long)(a.row_offset + grow);
                const cm_u4 rnd = cm_philox4x32((uint32_t)gr, (uint32_t)(gr >> 32), (uint32_t)a.t, CM_STREAM_ACT,
                                                (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
                const float u = cm_u01(rnd.x);
                float cum = 0.0f;
                int chosen = -1, last = 0;
                for (int k = 0; k < dout; ++k) {
                    if (z[k] > -5e8f) {
                        cum += expf(z[k] - lse);
                        last = k;
                        if (chosen < 0 && u < cum) chosen = k;
                    }
                }
                if (chosen < 0) chosen = last;
                a.action_out[grow * a.out_stride] = chosen;
                a.logp_out[grow * a.out_stride] = z[chosen] - lse;
            }
            continue;
        }

        if (TRAIN) {
            // row decode: row = seq*T + t ; seq = e*Aseq + ag
            const int Aseq = (MODE == M_ACTOR || a.per_agent) ? a.A : 1;
            if (hq == 0) {
                float* z = ls + hrow * lstride;
                bool valid = false;
                int ag = 0, e = 0, t = 0;
                if (rvalid) {
                    const long seq = grow / a.T;
                    t = (int)(grow - seq * a.T);
                    e = (int)(seq / Aseq);
                    ag = (int)(seq - (long)e * Aseq);
                    valid = t < a.ep_len[e];
                }
                const float invA = 1.0f / (float)a.A;
                if (MODE == M_ACTOR) {
                    if (valid) {
                        float m = -INFINITY;
                        for (int k = 0; k < dout; ++k) m = fmaxf(m, z[k]);
                        float s = 0.0f;
                        for (int k = 0; k < dout; ++k) s += expf(z[k] - m);
                        const float lse = m + logf(s);
                        float ent = 0.0f;
                        for (int k = 0; k < dout; ++k) {
                            const float lp = z[k] - lse;
                            ent -= expf(lp) * lp;
                        }
                        const int act = a.action[grow];
                        const float lpa = z[act] - lse;
                        const float log_ratio = lpa - a.logp_old[grow];
                        const float ratio = expf(log_ratio);
                        const float advv = a.adv[grow];
                        const float pg1 = advv * ratio;
                        const float pg2 = advv * fminf(fmaxf(ratio, a.clip_lo), a.clip_hi);
                        const bool inr = (ratio >= a.clip_lo) && (ratio <= a.clip_hi);
                        // d min(pg1,pg2)/d ratio with torch's tie rule (grad/2 to each operand)
                        float g;
                        if (pg1 < pg2) g = advv;
                        else if (pg1 > pg2) g = inr ? advv : 0.0f;
                        else g = 0.5f * advv + (inr ? 0.5f * advv : 0.0f);
                        st_pg += invA * fminf(pg1, pg2);
                        st_ent += invA * ent;
                        st_kl += invA * ((ratio - 1.0f) - log_ratio);
                        st_clip += (fabsf(ratio - 1.0f) > a.clip_eps) ? invA : 0.0f;
                        if (ag == 0) st_cnt += 1.0f;
                        const float gr = g * ratio;
                        for (int k = 0; k < dout; ++k) {
                            const float zk = z[k];
                            const float lp = zk - lse;
                            const float p = expf(lp);
                            float d = invA * (-gr * ((k == act ? 1.0f : 0.0f) - p) + a.ent_coef * p * (lp + ent));
                            if (zk <= -5e8f) d = 0.0f;  // masked_fill blocks the gradient
                            z[k] = d;
                        }
                    } else {
                        for (int k = 0; k < dout; ++k) z[k] = 0.0f;
                    }
                } else {  // M_CRITIC
                    float d = 0.0f;
                    if (valid) {
                        const float v = z[0];
                        if (a.per_agent) {
                            const float df = v - a.ret[grow];
                            st_vl += invA * df * df;
                            d = 2.0f * invA * df;
                            if (ag == 0) st_cnt += 1.0f;
                        } else {
                            float sd = 0.0f, sq = 0.0f;
                            for (int q = 0; q < a.A; ++q) {
                                const float df = v - a.ret[((long)e * a.A + q) * a.T + t];
                                sd += df; sq += df * df;
                            }
                            st_vl += invA * sq;
                            d = 2.0f * invA * sd;
                            st_cnt += 1.0f;
                        }
                    }
                    z[0] = d;
                }
            }
            __syncthreads();
            // ---- dWout, dbout (contraction over the tile's rows; reads HL before it is overwritten)
#pragma unroll
            for (int j = 0; j < DWO; ++j) {
                const int o = tid + NTHREADS * j;
                if (o < dout * HP) {
                    const int k = o >> 6, c = o & 63;
                    float s = 0.0f;
#pragma unroll 8
                    for (int r = 0; r < TM; ++r) s = fmaf(ls[r * lstride + k], HL[r * LDT + c], s);
                    dwo[j] += s;
                }
            }
            if (tid < dout) {
                float s = 0.0f;
                for (int r = 0; r < TM; ++r) s += ls[r * lstride + tid];
                dbo += s;
            }
            __syncthreads();
            // ---- dZ_L = (dlogits * Wout) .* relu'(H_L), in place (each lane owns its 16 columns)
            {
                float dz[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) dz[i] = 0.0f;
                for (int k = 0; k < dout; ++k) {
                    const float d = ls[hrow * lstride + k];
                    const float4* wp4 = reinterpret_cast<const float4*>(wouts + k * HP + 16 * hq);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float4 w4 = wp4[i];
                        dz[4 * i] = fmaf(d, w4.x, dz[4 * i]); dz[4 * i + 1] = fmaf(d, w4.y, dz[4 * i + 1]);
                        dz[4 * i + 2] = fmaf(d, w4.z, dz[4 * i + 2]); dz[4 * i + 3] = fmaf(d, w4.w, dz[4 * i + 3]);
                    }
                }
                float4* hp4 = reinterpret_cast<float4*>(HL + hrow * LDT + 16 * hq);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float4 v;
                    v.x = hreg[4 * i] > 0.0f ? dz[4 * i] : 0.0f;
                    v.y = hreg[4 * i + 1] > 0.0f ? dz[4 * i + 1] : 0.0f;
                    v.z = hreg[4 * i + 2] > 0.0f ? dz[4 * i + 2] : 0.0f;
                    v.w = hreg[4 * i + 3] > 0.0f ? dz[4 * i + 3] : 0.0f;
                    hp4[i] = v;
                }
            }
            __syncthreads();
            // ================= backward through hidden layers =================
#pragma unroll
            for (int l = LMAX; l >= 1; --l) {
                if (l <= L) {
                    float* Zl = smem + lds.Hs(l);       // holds dZ_l
                    float* Hm = smem + lds.Hs(l - 1);   // holds H_{l-1}
                    if (!ws_resident) {
                        stage_rows(Ws, a.params + off.Wl(l - 1), 0, H, H, 0, H);
                        __syncthreads();
                    }
                    {   // bias gradient: column sums
                        const int c = tid & 63, part = tid >> 6;
                        float s = 0.0f;
#pragma unroll
                        for (int r = 0; r < TM / 4; ++r) s += Zl[(part * (TM / 4) + r) * LDT + c];
                        dbh[l] += s;
                    }
                    colred(accWl[l - 1], Zl + 32 * wm, Hm + 32 * wn);
#pragma unroll
                    for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
                    rowpar_tn(acc, Zl + 32 * wm * LDT, Ws + 32 * wn);
                    __syncthreads();
#pragma unroll
                    for (int g = 0; g < 16; ++g) {
                        const int row = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                        float* p = Hm + row * LDT + 32 * wn + lc;
                        *p = (*p > 0.0f) ? acc[g] : 0.0f;
                    }
                    __syncthreads();
                }
            }
            // ================= layer 0 backward: bias + dW0 chunks =================
            {
                float* Z0 = smem + lds.Hs(0);
                {
                    const int c = tid & 63, part = tid >> 6;
                    float s = 0.0f;
#pragma unroll
                    for (int r = 0; r < TM / 4; ++r) s += Z0[(part * (TM / 4) + r) * LDT + c];
                    dbh[0] += s;
                }
#pragma unroll
                for (int c = 0; c < (NCH > 0 ? NCH : 1); ++c) {
                    if (NCH > 1) {
                        __syncthreads();
                        const int w = min(KC, din - c * KC);
                        stage_rows(Xs, a.x, row0, a.rows, a.x_stride, c * KC, w);
                        __syncthreads();
                    }
                    colred(accW0[c], Z0 + 32 * wm, Xs + 32 * wn);
                }
            }
        }
    }

    // ================= write this workgroup's partial gradient + stats =================
    if (TRAIN) {
        float* out = a.partial + (size_t)blockIdx.x * a.PS;
        // dW0: wave (wm, wn) holds rows n = 32wm + i, cols k = 64c + 32wn + j
#pragma unroll
        for (int c = 0; c < (NCH > 0 ? NCH : 1); ++c) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int n = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                const int k = KC * c + 32 * wn + lc;
                if (n < H && k < din) out[off.W0 + n * din + k] = accW0[c][g];
            }
        }
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < L) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int n = 32 * wm + (g & 3) + 8 * (g >> 2) + 4 * h;
                    const int k = 32 * wn + lc;
                    if (n < H && k < H) out[off.Wl(l) + n * H + k] = accWl[l][g];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < DWO; ++j) {
            const int o = tid + NTHREADS * j;
            if (o < dout * HP) {
                const int k = o >> 6, c = o & 63;
                if (c < H) out[off.Wout + k * H + c] = dwo[j];
            }
        }
        if (tid < dout) out[off.bout + tid] = dbo;
        // bias grads: 4 row-parts per column -> LDS -> sum
#pragma unroll
        for (int l = 0; l <= LMAX; ++l) {
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

// sum per-workgroup partials: out[i] = sum_w partial[w][i]
__global__ __launch_bounds__(256) void k_reduce_partials(const float* __restrict__ partial, int nparts, int PS, int n,
                                                         float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int w = 0;
    for (; w + 3 < nparts; w += 4) {
        s0 += partial[(size_t)w * PS + i];
        s1 += partial[(size_t)(w + 1) * PS + i];
        s2 += partial[(size_t)(w + 2) * PS + i];
        s3 += partial[(size_t)(w + 3) * PS + i];
    }
    for (; w < nparts; ++w) s0 += partial[(size_t)w * PS + i];
    out[i] = (s0 + s1) + (s2 + s3);
}

constexpr int MAX_GRID = 256;  // one persistent workgroup per CU (LDS-limited)

int check_shapes(const char* who, int din, int H, int L, int dout) {
    CM_REQUIRE(din > 0 && H > 0 && L >= 0 && dout > 0, "%s: bad dims din=%d H=%d L=%d dout=%d", who, din, H, L, dout);
    CM_REQUIRE(H <= HP, "%s: hidden_dim=%d > %d is not supported by this build", who, H, HP);
    CM_REQUIRE(L <= LMAX, "%s: num_layers=%d > %d is not supported by this build", who, L, LMAX);
    CM_REQUIRE(dout <= KMAX, "%s: output width %d > %d is not supported by this build", who, dout, KMAX);
    return 0;
}

int grid_for(long rows) {
    long nt = (rows + TM - 1) / TM;
    return (int)(nt < MAX_GRID ? nt : MAX_GRID);
}

template <int MODE>
int launch_train(const MlpArgs& a, int grid, size_t lds_bytes, hipStream_t s) {
    const int nch = (a.din + KC - 1) / KC;
#define CM_CASE(N) case N: { \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp<N, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        hipLaunchKernelGGL((k_mlp<N, MODE>), dim3(grid), dim3(NTHREADS), lds_bytes, s, a); break; }
    switch (nch) {
        CM_CASE(1) CM_CASE(2) CM_CASE(3) CM_CASE(4) CM_CASE(5) CM_CASE(6) CM_CASE(7) CM_CASE(8)
        default: CM_FAIL(-1, "input width %d > %d is not supported by the fused training kernels", a.din, 8 * KC);
    }
#undef CM_CASE
    return 0;
}

}  // namespace

extern "C" int cm_mlp_forward(const float* x, int64_t rows, int din, int hidden, int n_hidden_layers, int dout,
                              const float* params, const uint8_t* avail, float* y, cm_stream_t stream) {
    if (int rc = check_shapes("cm_mlp_forward", din, hidden, n_hidden_layers, dout)) return rc;
    if (rows <= 0) return 0;
    MlpArgs a = {};
    a.x = x; a.x_stride = din; a.rows = rows; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = dout;
    a.params = params; a.avail = avail; a.avail_stride = dout; a.y = y;
    const size_t lds_bytes = (size_t)make_lds(a.L, a.dout).total * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp<0, M_FWD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((k_mlp<0, M_FWD>), dim3(grid_for(rows)), dim3(NTHREADS), lds_bytes, (hipStream_t)stream, a);
    CM_CHECK_LAUNCH("cm_mlp_forward");
    return 0;
}

extern "C" int cm_policy_act(const float* x, int64_t x_row_stride, const uint8_t* avail, int64_t avail_row_stride,
                             int64_t rows, int din, int hidden, int n_hidden_layers, int n_actions,
                             const float* params, uint64_t seed, int64_t row_offset, int t,
                             int32_t* action, float* logp, int64_t out_stride, cm_stream_t stream) {
    if (int rc = check_shapes("cm_policy_act", din, hidden, n_hidden_layers, n_actions)) return rc;
    if (rows <= 0) return 0;
    MlpArgs a = {};
    a.x = x; a.x_stride = x_row_stride; a.rows = rows; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = n_actions;
    a.params = params; a.avail = avail; a.avail_stride = avail_row_stride;
    a.seed = seed; a.row_offset = row_offset; a.t = t; a.action_out = action; a.logp_out = logp; a.out_stride = out_stride;
    const size_t lds_bytes = (size_t)make_lds(a.L, a.dout).total * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp<0, M_ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL((k_mlp<0, M_ACT>), dim3(grid_for(rows)), dim3(NTHREADS), lds_bytes, (hipStream_t)stream, a);
    CM_CHECK_LAUNCH("cm_policy_act");
    return 0;
}

extern "C" size_t cm_mlp_train_workspace_bytes(int din, int hidden, int n_hidden_layers, int dout) {
    const int64_t P = cm_mlp_param_count(din, hidden, n_hidden_layers, dout);
    const size_t PS = (size_t)((P + CM_NUM_STATS + 63) / 64 * 64);
    return (size_t)MAX_GRID * PS * sizeof(float);
}

static int finish_train(const MlpArgs& a, int grid, int64_t P, float* grad_and_stats, hipStream_t s, const char* who) {
    const int n = (int)(P + CM_NUM_STATS);
    hipLaunchKernelGGL(k_reduce_partials, dim3((n + 255) / 256), dim3(256), 0, s, a.partial, grid, a.PS, n, grad_and_stats);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) CM_FAIL(-2, "%s: reduce launch failed: %s", who, hipGetErrorString(e));
    return 0;
}

extern "C" int cm_ppo_actor_fwd_bwd(const float* obs, const uint8_t* avail, const int32_t* action,
                                    const float* logp_old, const float* adv, const int32_t* ep_len,
                                    int E, int A, int T, int din, int hidden, int n_hidden_layers, int n_actions,
                                    const float* params, double ppo_clip, double entropy_coef,
                                    float* grad_and_stats, void* ws, size_t ws_bytes, cm_stream_t stream) {
    if (int rc = check_shapes("cm_ppo_actor_fwd_bwd", din, hidden, n_hidden_layers, n_actions)) return rc;
    CM_REQUIRE(E > 0 && A > 0 && T > 0, "cm_ppo_actor_fwd_bwd: bad dims E=%d A=%d T=%d", E, A, T);
    const size_t need = cm_mlp_train_workspace_bytes(din, hidden, n_hidden_layers, n_actions);
    CM_REQUIRE(ws && ws_bytes >= need, "cm_ppo_actor_fwd_bwd: workspace too small (%zu < %zu)", ws_bytes, need);
    const int64_t P = cm_mlp_param_count(din, hidden, n_hidden_layers, n_actions);
    MlpArgs a = {};
    a.x = obs; a.x_stride = din; a.rows = (long)E * A * T; a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = n_actions;
    a.params = params; a.avail = avail; a.avail_stride = n_actions;
    a.action = action; a.logp_old = logp_old; a.adv = adv; a.ep_len = ep_len; a.A = A; a.T = T; a.per_agent = 1;
    a.clip_lo = (float)(1.0 - ppo_clip); a.clip_hi = (float)(1.0 + ppo_clip); a.clip_eps = (float)ppo_clip;
    a.ent_coef = (float)entropy_coef;
    a.partial = (float*)ws; a.PS = (int)((P + CM_NUM_STATS + 63) / 64 * 64);
    const int grid = grid_for(a.rows);
    const size_t lds_bytes = (size_t)make_lds(a.L, a.dout).total * sizeof(float);
    if (int rc = launch_train<M_ACTOR>(a, grid, lds_bytes, (hipStream_t)stream)) return rc;
    CM_CHECK_LAUNCH("cm_ppo_actor_fwd_bwd");
    return finish_train(a, grid, P, grad_and_stats, (hipStream_t)stream, "cm_ppo_actor_fwd_bwd");
}

extern "C" int cm_critic_fwd_bwd(const float* x, const float* ret, const int32_t* ep_len,
                                 int E, int A, int T, int per_agent, int din, int hidden, int n_hidden_layers,
                                 const float* params, float* grad_and_stats, void* ws, size_t ws_bytes,
                                 cm_stream_t stream) {
    if (int rc = check_shapes("cm_critic_fwd_bwd", din, hidden, n_hidden_layers, 1)) return rc;
    CM_REQUIRE(E > 0 && A > 0 && T > 0, "cm_critic_fwd_bwd: bad dims E=%d A=%d T=%d", E, A, T);
    const size_t need = cm_mlp_train_workspace_bytes(din, hidden, n_hidden_layers, 1);
    CM_REQUIRE(ws && ws_bytes >= need, "cm_critic_fwd_bwd: workspace too small (%zu < %zu)", ws_bytes, need);
    const int64_t P = cm_mlp_param_count(din, hidden, n_hidden_layers, 1);
    MlpArgs a = {};
    a.x = x; a.x_stride = din; a.rows = per_agent ? (long)E * A * T : (long)E * T;
    a.din = din; a.H = hidden; a.L = n_hidden_layers; a.dout = 1;
    a.params = params; a.ret = ret; a.ep_len = ep_len; a.A = A; a.T = T; a.per_agent = per_agent ? 1 : 0;
    a.partial = (float*)ws; a.PS = (int)((P + CM_NUM_STATS + 63) / 64 * 64);
    const int grid = grid_for(a.rows);
    const size_t lds_bytes = (size_t)make_lds(a.L, a.dout).total * sizeof(float);
    if (int rc = launch_train<M_CRITIC>(a, grid, lds_bytes, (hipStream_t)stream)) return rc;
    CM_CHECK_LAUNCH("cm_critic_fwd_bwd");
    return finish_train(a, grid, P, grad_and_stats, (hipStream_t)stream, "cm_critic_fwd_bwd");
}
