// cm_gru_v2.h -- second generation of the 32-row TBPTT sweeps (included by cm_gru.hip inside its anonymous namespace).
//
// Same arithmetic as the 64-row sweeps k_gru_chunk_fwd / k_gru_chunk_bwd (reference: cleanmarl/mappo_lstm_multienvs.py:162-184, 562-620); what
// changes is where things live, because at config 5 the sweeps are a latency chain (5120 sequences = 160 workgroups on 256 CUs, one
// per CU, time strictly sequential), not a throughput problem:
//   * WEIGHTS IN REGISTERS.  Every gate block is the B operand of a 32x32x2 MFMA whose lane (n = lane & 31, h = lane >> 5) always
//     supplies the same 32 weights; a wave keeps the blocks it multiplies with in VGPRs for the whole chunk (forward: 4 blocks =
//     114 registers, backward: 3 transposed blocks = 96).  No weight tile in LDS, no per-step staging, half the LDS reads per MFMA,
//     and LDS is free for what follows.
//   * ROW-MAJOR WORKSPACE STORES.  r, z, n, W_hn h go through LDS tiles and leave as 16-byte stores (12 per thread and step instead
//     of 64 scattered 4-byte stores from the accumulator layout, which were 2.4 us of the 11.6 us forward step).
//   * THE HEAD LEAVES THE RECURRENCE.  logits -> PPO loss -> dlogits, the fc2 weight gradient and the head's contribution to dh
//     ((dlogits W2) .* (h' > 0)) depend on h'_t only, not on the chain: the forward kernel computes them AFTER its step loop for two
//     steps at a time on the MFMA (64 (row, step) items per pass) and hands the backward sweep a ready dh_head[s] tile in the
//     workspace.  The backward step loses its head phase (2 barriers, a 128-thread VALU GEMV) and its weight staging (6 barriers).
//   * 4 barriers per step in both sweeps (were 7 / 12).
// Limits: K <= 32 actions (head padded to 16 or 32), din <= 64, H <= 64 (the dispatcher keeps the first-generation kernels for anything else).
// Workspace: 7 slots per (step, row): x1 | r | z | n | W_hn h + b_hn | h' | dh_head.
#pragma once

#ifdef CM_PHASE_PROF
#define PH2_FLUSH(base) do { if (a.prof && threadIdx.x == 0) { _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) a.prof[(size_t)((base) + blockIdx.x) * 16 + i_] = ph_[i_]; } } while (0)
#else
#define PH2_FLUSH(base)
#endif
constexpr int WS2 = 7 * HP;

// B-operand register image, "nt" form (Y = A W^T): w[4j + i] = W[(n0 + r) * ld + 8j + 4h + i], zero outside [nrows) x [ncols)
template <bool VEC>
__device__ __forceinline__ void load_nt_regs(float (&w)[32], const float* W, int n0, int nrows, int ld, int ncols) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int n = n0 + r;
    const float* p = W + (long)n * ld + 4 * h;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (VEC) {  // ld % 4 == 0, 16-byte aligned base, ncols % 4 == 0
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < nrows && 8 * j + 4 * h < ncols) v = *reinterpret_cast<const float4*>(p + 8 * j);
            w[4 * j] = v.x; w[4 * j + 1] = v.y; w[4 * j + 2] = v.z; w[4 * j + 3] = v.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) w[4 * j + i] = (n < nrows && 8 * j + 4 * h + i < ncols) ? p[8 * j + i] : 0.0f;
        }
    }
}
// "tn" form (dX = dZ W): w[4j + q] = W[(8j + 4h + q) * ld + c0 + r], zero outside [nrows) x [ncols)
__device__ __forceinline__ void load_tn_regs(float (&w)[32], const float* W, int c0, int nrows, int ld, int ncols) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const int c = c0 + r;
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = 8 * j + 4 * h + q;
            w[4 * j + q] = (n < nrows && c < ncols) ? W[(long)n * ld + c] : 0.0f;
        }
}
// acc[32 rows x 32 cols] += A[32 rows][8 kb] * (register image); k order identical to rowpar_nt / rowpar_tn
__device__ __forceinline__ void rowpar_rb(f32x16& acc, const float* As, const float (&w)[32], int kb) {
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

constexpr int g2f_lds_floats(int KP) { return 8 * T32 * LDT + TM * LDT + KP * WLD + 2 * TM * KP + KMAX + 2 * NTHREADS; }
inline size_t gru2_fwd_lds_bytes(int KP) { return (size_t)g2f_lds_floats(KP) * sizeof(float); }

// KP: padded head width, 16 (K <= 16) or 32
template <bool WV, int KP>
__global__ __launch_bounds__(NTHREADS, 1) void k_gru2_fwd(const GruArgs a) {
    constexpr int KJ = KP / 4, NCT = KP / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GruOff off = gru_offsets(a.din, a.H, a.K);
    float* p = smem;
    float* X0 = p; p += T32 * LDT;   // obs tile of the step
    float* X1 = p; p += T32 * LDT;   // x1 = relu(fc1(obs))
    float* hp = p; p += T32 * LDT;   // h_{t-1}
    float* hn = p; p += T32 * LDT;   // h_t
    float* SR = p; p += T32 * LDT;
    float* SZ = p; p += T32 * LDT;
    float* SN = p; p += T32 * LDT;
    float* SG = p; p += T32 * LDT;   // W_hn h + b_hn
    float* HB = p; p += TM * LDT;    // head pass: relu(h') of two steps (64 items), then dh_head
    float* wouts = p; p += KP * WLD;
    float* ls = p; p += TM * KP;     // logits of the 64 items
    float* ls2 = p; p += TM * KP;    // dlogits of the 64 items
    float* b2 = p; p += KMAX;
    float* red = p;                  // 2 * NTHREADS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, g = wave >> 1, h = lane >> 5, lc = lane & 31;  // head pass: wave (g, wn) = items 32 g .., columns 32 wn ..
    const int col = 32 * wn + lc;
    const int H = a.H, K = a.K, din = a.din, T = a.T, CL = a.t1 - a.t0;
    const long R = (long)a.E * a.A;
    // ---- one-time: this wave's weight columns -> registers, head weights + biases -> LDS
    G2W w;
    g2_load_weights<WV>(w, a.params, off, din, H);
    for (int i = tid; i < KP * WLD; i += NTHREADS) {
        const int k = i / WLD, c = i % WLD;
        wouts[i] = (c < H && k < K) ? a.params[off.W2 + k * H + c] : 0.0f;
    }
    for (int i = tid; i < KMAX; i += NTHREADS) b2[i] = (i < K) ? a.params[off.b2 + i] : 0.0f;

    PH_DECL
    float st_pg = 0.f, st_ent = 0.f, st_kl = 0.f, st_clip = 0.f, st_cnt = 0.f;
    f32x4 accWo[NCT];  // dW2[k = 16 ct + 4 (lane >> 4) + q][hidden column 16 wave + (lane & 15)]
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) accWo[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbo = 0.f;
    const long ntiles = (R + T32 - 1) / T32;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * T32;
        __syncthreads();
        for (int i = tid; i < T32 * HP; i += NTHREADS) {
            const int r = i >> 6, c = i & 63;
            hp[r * LDT + c] = (row0 + r < R && c < H && a.h_in) ? a.h_in[(row0 + r) * H + c] : 0.0f;
        }
        X32 xr;
        x32_load(xr, a.obs + (long)a.t0 * din, row0, R, (long)T * din, din);
        x32_store(X0, xr);
        if (CL > 1) x32_load(xr, a.obs + (long)(a.t0 + 1) * din, row0, R, (long)T * din, din);
        // ================================================================ the recurrence: 3 barriers per step (gru2_step)
        for (int s = 0; s < CL; ++s) {
            const int t = a.t0 + s;
            lds_barrier();  // X0 = obs(t), hp = h_{t-1}; the previous step's stores have read X1 / S* / hn
            PH(0);
            gru2_step<true>(w, X0, X1, hp, hn, SR, SZ, SN, SG, din, H);
            PH(1);
            // obs(t + 1) (requested a step ago) -> X0: every read of obs(t) is two barriers back.  obs(t + 2) is requested BEFORE this
            // step's workspace stores: queued behind those 12 KB per wave the loads stalled the wave for ~1400 cycles at issue
            if (s + 1 < CL) x32_store(X0, xr);
            if (s + 2 < CL) x32_load(xr, a.obs + (long)(t + 2) * din, row0, R, (long)T * din, din);
            {   // six row-major tiles leave as 16-byte stores
                float* wsS = a.ws_act + ((long)s * R + row0) * WS2;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int idx = tid + NTHREADS * q, r = idx >> 4, c4 = (idx & 15) * 4;
                    if (row0 + r < R) {
                        float* wp = wsS + (long)r * WS2 + c4;
                        const int o = r * LDT + c4;
                        *reinterpret_cast<float4*>(wp) = *reinterpret_cast<const float4*>(X1 + o);
                        *reinterpret_cast<float4*>(wp + HP) = *reinterpret_cast<const float4*>(SR + o);
                        *reinterpret_cast<float4*>(wp + 2 * HP) = *reinterpret_cast<const float4*>(SZ + o);
                        *reinterpret_cast<float4*>(wp + 3 * HP) = *reinterpret_cast<const float4*>(SN + o);
                        *reinterpret_cast<float4*>(wp + 4 * HP) = *reinterpret_cast<const float4*>(SG + o);
                        *reinterpret_cast<float4*>(wp + 5 * HP) = *reinterpret_cast<const float4*>(hn + o);
                    }
                }
            }
            float* tmp = hp; hp = hn; hn = tmp;
            PH(5);
        }
        __syncthreads();
        if (a.h_out)
            for (int i = tid; i < T32 * HP; i += NTHREADS) {
                const int r = i >> 6, c = i & 63;
                if (row0 + r < R && c < H) a.h_out[(row0 + r) * H + c] = hp[r * LDT + c];
            }
        // ================================================================ the head, two steps (64 (row, step) items) per pass
        const int hrow = tid >> 2, hq = tid & 3;   // PPO math: 4 lanes per item
        const int irow = hrow & 31;                // tile row of the item
        const long grow = row0 + irow;
        const bool rvalid = grow < R;
        const int e_row = rvalid ? (int)(grow / a.A) : 0;
        const int ag = (int)(grow - (long)e_row * a.A);
        const int eplen = rvalid ? a.ep_len[e_row] : 0;
        const float invA = 1.0f / (float)a.A;
        // relu(h') tiles of a pass are requested one pass ahead (they come back from L2: ~3400 cycles if waited for in place)
        float4 hpre[4];
        auto head_load = [&](int sA) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = tid + NTHREADS * q, it = idx >> 4, c4 = (idx & 15) * 4;
                const int ss = sA + (it >> 5), r = it & 31;
                hpre[q] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ss < CL && row0 + r < R) hpre[q] = *reinterpret_cast<const float4*>(a.ws_act + ((long)ss * R + row0 + r) * WS2 + 5 * HP + c4);
            }
        };
        head_load(0);  // the rows were written by this workgroup: the __syncthreads() above made them visible
        // per-item inputs of a pass (4 lanes per (row, step) item), also requested one pass ahead and BEFORE the previous pass's dh_head
        // stores: queued behind those the loads stalled the wave ~1800 cycles per pass at issue
        struct Item { unsigned char avb[KJ]; int act; float lpo, advv; };
        auto item_load = [&](int sA, Item& it) {
            const int s_i = sA + (hrow >> 5), t_i = a.t0 + s_i;
            const bool ok = rvalid && s_i < CL;
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                it.avb[j] = 1;
                if (ok && 4 * j + hq < K) it.avb[j] = a.avail[(grow * T + t_i) * K + 4 * j + hq];
            }
            const long o = grow * T + t_i;
            it.act = ok ? a.action[o] : 0;
            it.lpo = ok ? a.logp_old[o] : 0.f; it.advv = ok ? a.adv[o] : 0.f;
        };
        Item cur;
        item_load(0, cur);
        for (int s0 = 0; s0 < CL; s0 += 2) {
            const int s_it = s0 + (hrow >> 5), t_it = a.t0 + s_it;
            const bool ivalid = rvalid && s_it < CL;
            unsigned char avb[KJ];
#pragma unroll
            for (int j = 0; j < KJ; ++j) avb[j] = cur.avb[j];
            const int act = cur.act;
            const float lpo = cur.lpo, advv = cur.advv;
            lds_barrier();  // HB / ls / ls2 of the previous pass are dead
            PH(6);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = tid + NTHREADS * q, it = idx >> 4, c4 = (idx & 15) * 4;
                float4 v = hpre[q];
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                *reinterpret_cast<float4*>(HB + it * LDT + c4) = v;
            }
            if (s0 + 2 < CL) head_load(s0 + 2);
            lds_barrier();
            PH(7);
            {   // logits on the 16x16x4 MFMA: wave w = items 16w..16w+15, one call per 16 head outputs
                const int n = lane & 15, g4 = lane >> 4;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const f32x4 lg = head_logits_mfma(HB + 16 * wave * LDT, wouts + 16 * ct * WLD);
                    const float bias = b2[16 * ct + n];
#pragma unroll
                    for (int q = 0; q < 4; ++q) ls[(16 * wave + 4 * g4 + q) * KP + 16 * ct + n] = lg[q] + bias;
                }
            }
            lds_barrier();
            PH(8);
            {   // PPO clipped-surrogate head (arithmetic of k_gru_chunk_fwd): statistics + dlogits -> ls2
                float zreg[KJ];
#pragma unroll
                for (int j = 0; j < KJ; ++j) zreg[j] = (4 * j + hq < K && avb[j]) ? ls[hrow * KP + 4 * j + hq] : -1e9f;
                const bool valid = ivalid && t_it < eplen;
                float m = -INFINITY;
#pragma unroll
                for (int j = 0; j < KJ; ++j) if (4 * j + hq < K) m = fmaxf(m, zreg[j]);
                m = quad_max(m);
                float ssum = 0.0f, pj[KJ];
#pragma unroll
                for (int j = 0; j < KJ; ++j) { pj[j] = 0.f; if (4 * j + hq < K) { pj[j] = expf(zreg[j] - m); ssum += pj[j]; } }
                ssum = quad_sum(ssum);
                const float lse = m + logf(ssum), rs = 1.0f / ssum;
                float ent = 0.f, lpa = 0.f;
#pragma unroll
                for (int j = 0; j < KJ; ++j) if (4 * j + hq < K) {
                    const float lp = zreg[j] - lse;
                    pj[j] *= rs; ent -= pj[j] * lp;
                    if (4 * j + hq == act) lpa = lp;
                }
                ent = quad_sum(ent); lpa = quad_sum(lpa);
                const float log_ratio = lpa - lpo, ratio = expf(log_ratio);
                const float pg1 = advv * ratio, pg2 = advv * fminf(fmaxf(ratio, a.clip_lo), a.clip_hi);
                const bool inr = (ratio >= a.clip_lo) && (ratio <= a.clip_hi);
                float gsel;
                if (pg1 < pg2) gsel = advv; else if (pg1 > pg2) gsel = inr ? advv : 0.f; else gsel = 0.5f * advv + (inr ? 0.5f * advv : 0.f);
                if (valid && hq == 0) {
                    st_pg += invA * fminf(pg1, pg2); st_ent += invA * ent; st_kl += invA * ((ratio - 1.f) - log_ratio);
                    st_clip += (fabsf(ratio - 1.f) > a.clip_eps) ? invA : 0.f;
                    if (ag == 0) st_cnt += 1.f;
                }
                const float gr = gsel * ratio;
#pragma unroll
                for (int j = 0; j < KJ; ++j) {
                    const int k = 4 * j + hq;
                    float d = 0.f;
                    if (k < K && valid && zreg[j] > -5e8f) {
                        const float lp = zreg[j] - lse;
                        d = invA * (-gr * ((k == act ? 1.f : 0.f) - pj[j]) + a.ent_coef * pj[j] * (lp + ent));
                    }
                    ls2[hrow * KP + k] = d;
                }
            }
            lds_barrier();
            PH(9);
            // fc2 gradient: dW2 += dlogits^T relu(h') over the 64 items (wave = 16 hidden columns), db2 += column sums
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) colred_head16<KP>(accWo[ct], ls2, 16 * ct, HB + 16 * wave);
            {
                constexpr int PARTS = NTHREADS / KP, RPP = TM / PARTS;
                const int k = tid % KP, part = tid / KP;
                float sb = 0.f;
#pragma unroll
                for (int r = 0; r < RPP; ++r) sb += ls2[(part * RPP + r) * KP + k];
                dbo += sb;
            }
            // dh_head = (dlogits W2) .* (h' > 0): wave (g, wn) = items 32 g .., columns 32 wn ..
            f32x16 dh;
#pragma unroll
            for (int i = 0; i < 16; ++i) dh[i] = 0.0f;
            head_bwd_mfma<KP>(dh, ls2 + 32 * g * KP, wouts + 32 * wn);
            lds_barrier();  // every read of HB (fc2 gradient) is done: it becomes the dh_head tile
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float* q = HB + (32 * g + (i & 3) + 8 * (i >> 2) + 4 * h) * LDT + col;
                *q = (*q > 0.0f) ? dh[i] : 0.0f;
            }
            lds_barrier();
            if (s0 + 2 < CL) item_load(s0 + 2, cur);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = tid + NTHREADS * q, it = idx >> 4, c4 = (idx & 15) * 4;
                const int ss = s0 + (it >> 5), r = it & 31;
                if (ss < CL && row0 + r < R)
                    *reinterpret_cast<float4*>(a.ws_act + ((long)ss * R + row0 + r) * WS2 + 6 * HP + c4) = *reinterpret_cast<const float4*>(HB + it * LDT + c4);
            }
            PH(10);
        }
    }
    PH2_FLUSH(0);
    // ================================ this workgroup's partial row: fc2 gradient + statistics (the rest comes from k_gru2_bwd)
    float* out = a.partial + (size_t)blockIdx.x * a.PS;
    {
        const int n = lane & 15, g4 = lane >> 4;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * ct + 4 * g4 + r, c = 16 * wave + n;
                if (k < K && c < H) out[off.W2 + k * H + c] = accWo[ct][r];
            }
    }
    __syncthreads();
    red[tid] = dbo;  // [NTHREADS / KP parts][KP]
    __syncthreads();
    if (tid < K) {
        float sb = 0.f;
#pragma unroll
        for (int q = 0; q < NTHREADS / KP; ++q) sb += red[q * KP + tid];
        out[off.b2 + tid] = sb;
    }
    float sv6[6] = {st_pg, st_ent, st_kl, st_clip, 0.f, st_cnt};
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 6; ++q) {
        const float v = cm_wave_sum(sv6[q]);
        if (lane == 0) red[q * 4 + wave] = v;
    }
    __syncthreads();
    if (tid < CM_NUM_STATS) {
        float v = 0.f;
        if (tid < 6) v = red[tid * 4] + red[tid * 4 + 1] + red[tid * 4 + 2] + red[tid * 4 + 3];
        out[off.P + tid] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_gru2_fwd8: the forward sweep with EIGHT waves per 32-row tile.  Waves 0-3 run the recurrence of k_gru2_fwd and nothing else;
// waves 4-7 ("helpers") take everything that only CONSUMES a finished step, one and two steps behind the chain:
//   * the six row-major workspace tiles of step s - 1 leave during step s (they were 2.2 k of the 17 k cycles of a step);
//   * the head (logits -> PPO loss -> dlogits -> fc2 gradient -> dh_head), which k_gru2_fwd runs AFTER its step loop (8.6 k cycles per
//     pair of steps with the recurrence waves idle), runs DURING the loop: relu(h') of two steps is collected straight from the LDS
//     tiles into HB[pass & 1] (no reload from the workspace), the four phases of a pass are spread over the intervals between the
//     recurrence's own three barriers per step, and dh_head leaves from the MFMA accumulators (128-byte runs per half wave).
// Every wave executes the same barrier sequence (3 per step, two drain steps at the end of the chunk); x1 is double-buffered so that
// its tile survives until the helpers have stored it.  Interval map of step s (sp = s - 1, pass p = steps 2p, 2p + 1):
//     I0 (top .. "x1 complete"):   r, z, n, W_hn h tiles of sp -> workspace;  relu(h'_sp) -> HB[(sp >> 1) & 1];  s = 2p + 3: PPO math of pass p
//     I1 ("x1 complete" .. "h'"):  x1, h' tiles of sp -> workspace;  s = 2p + 2: logits of pass p;  s = 2p + 3: fc2 gradient + dh_head of pass p
// Same arithmetic in the same order as k_gru2_fwd (same thread mapping inside the helper group): both leave bit-identical results.
constexpr int NT8 = 2 * NTHREADS;
constexpr int g2f8_lds_floats(int KP) { return 9 * T32 * LDT + 2 * TM * LDT + KP * WLD + 2 * TM * KP + KMAX + 2 * NTHREADS + 8 * TM; }
inline size_t gru2_fwd8_lds_bytes(int KP) { return (size_t)g2f8_lds_floats(KP) * sizeof(float); }
#ifdef CM_PHASE_PROF
#define PH8_FLUSH() do { if (a.prof && (threadIdx.x == 0 || threadIdx.x == NTHREADS)) { const int b_ = threadIdx.x ? 8 : 0; _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) a.prof[(size_t)blockIdx.x * 16 + b_ + i_] = ph_[b_ + i_]; } } while (0)
#else
#define PH8_FLUSH()
#endif

template <bool WV, int KP>
__global__ __launch_bounds__(NT8) void k_gru2_fwd8(const GruArgs a) {
    constexpr int KJ = KP / 4, NCT = KP / 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GruOff off = gru_offsets(a.din, a.H, a.K);
    float* p = smem;
    float* X0 = p; p += T32 * LDT;    // obs tile of the step
    float* X1a = p; p += T32 * LDT;   // x1 = relu(fc1(obs)), even steps
    float* X1b = p; p += T32 * LDT;   //                     odd steps
    float* hp = p; p += T32 * LDT;    // h_{t-1}
    float* hn = p; p += T32 * LDT;    // h_t
    float* SR = p; p += T32 * LDT;
    float* SZ = p; p += T32 * LDT;
    float* SN = p; p += T32 * LDT;
    float* SG = p; p += T32 * LDT;    // W_hn h + b_hn
    float* HBa = p; p += TM * LDT;    // relu(h') of the two steps of an even pass (64 items)
    float* HBb = p; p += TM * LDT;    //                               odd pass
    float* wouts = p; p += KP * WLD;
    float* ls = p; p += TM * KP;      // logits of the 64 items
    float* ls2 = p; p += TM * KP;     // dlogits of the 64 items
    float* b2 = p; p += KMAX;
    float* red = p; p += 2 * NTHREADS;
    float* itf = p;                   // per-item inputs of a pass, staged by the recurrence waves: [64 items][act, logp_old, adv, -] ...
    unsigned* itav = reinterpret_cast<unsigned*>(p + 4 * TM);  // ... and [64 items][4 lanes] availability bits (bit j: action 4 j + lane)
    // the role is wave-uniform and the compiler must know it (a scalar branch): a divergent one keeps the helpers' registers live
    // through the recurrence
    const bool helper = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;
    const int tid = threadIdx.x & (NTHREADS - 1), lane = tid & 63, wave = tid >> 6;   // inside the role's group of four waves
    const int wn = wave & 1, g = wave >> 1, h = lane >> 5, lc = lane & 31;            // head: wave (g, wn) = items 32 g .., columns 32 wn ..
    const int col = 32 * wn + lc;
    const int H = a.H, K = a.K, din = a.din, T = a.T, CL = a.t1 - a.t0;
    const int NP = (CL + 1) >> 1, SV = 2 * NP + 2;   // passes of the head; steps including the drain
    const long R = (long)a.E * a.A;
    for (int i = threadIdx.x; i < KP * WLD; i += NT8) {
        const int k = i / WLD, c = i % WLD;
        wouts[i] = (c < H && k < K) ? a.params[off.W2 + k * H + c] : 0.0f;
    }
    for (int i = threadIdx.x; i < KMAX; i += NT8) b2[i] = (i < K) ? a.params[off.b2 + i] : 0.0f;

    PH_DECL
    float st_pg = 0.f, st_ent = 0.f, st_kl = 0.f, st_clip = 0.f, st_cnt = 0.f;
    f32x4 accWo[NCT];  // helpers: dW2[k = 16 ct + 4 (lane >> 4) + q][hidden column 16 wave + (lane & 15)]
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) accWo[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbo = 0.f;
    const long ntiles = (R + T32 - 1) / T32;
    if (!helper) {
        // ================================================================ waves 0-3: the recurrence, 3 barriers per step (gru2_step)
        G2W w;
        g2_load_weights<WV>(w, a.params, off, din, H);
        for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const long row0 = tile * T32;
            __syncthreads();
            for (int i = tid; i < T32 * HP; i += NTHREADS) {
                const int r = i >> 6, c = i & 63;
                hp[r * LDT + c] = (row0 + r < R && c < H && a.h_in) ? a.h_in[(row0 + r) * H + c] : 0.0f;
            }
            X32 xr;
            x32_load(xr, a.obs + (long)a.t0 * din, row0, R, (long)T * din, din);
            x32_store(X0, xr);
            if (CL > 1) x32_load(xr, a.obs + (long)(a.t0 + 1) * din, row0, R, (long)T * din, din);
            // per-item inputs of the head (4 lanes per (row, step) item): requested at the end of step 2p, in LDS at the end of step 2p + 1,
            // read by the helpers in step 2p + 3.  These waves issue no stores, so waiting for a load here waits for loads only -- on the
            // helper waves every such wait also drained their queue of workspace stores (in-order vmcnt)
            const int hrow = tid >> 2, hq = tid & 3;
            const long grow = row0 + (hrow & 31);
            const bool rvalid = grow < R;
            unsigned avbits = 0;
            int it_act = 0;
            float it_lpo = 0.f, it_adv = 0.f;
            for (int s = 0; s < SV; ++s) {
                lds_barrier();  // X0 = obs(t), hp = h_{t-1}
                PH(0);
                if (s < CL) {
                    gru2_step<true>(w, X0, (s & 1) ? X1b : X1a, hp, hn, SR, SZ, SN, SG, din, H);
                    PH(1);
                    if (s + 1 < CL) x32_store(X0, xr);
                    if (s + 2 < CL) x32_load(xr, a.obs + (long)(a.t0 + s + 2) * din, row0, R, (long)T * din, din);
                    float* tmp = hp; hp = hn; hn = tmp;
                    PH(2);
                } else {
                    if (s == CL && a.h_out)
                        for (int i = tid; i < T32 * HP; i += NTHREADS) {
                            const int r = i >> 6, c = i & 63;
                            if (row0 + r < R && c < H) a.h_out[(row0 + r) * H + c] = hp[r * LDT + c];
                        }
                    lds_barrier();
                    lds_barrier();
                    PH(3);
                }
                if (s & 1) {
                    if ((s >> 1) < NP) {
                        itav[hrow * 4 + hq] = avbits;
                        if (hq == 0) *reinterpret_cast<float4*>(itf + 4 * hrow) = make_float4(__int_as_float(it_act), it_lpo, it_adv, 0.f);
                    }
                } else if ((s >> 1) < NP) {
                    const int s_i = s + (hrow >> 5), t_i = a.t0 + s_i;
                    const bool ok = rvalid && s_i < CL;
                    avbits = 0;
#pragma unroll
                    for (int j = 0; j < KJ; ++j) {
                        unsigned v = 1;
                        if (ok && 4 * j + hq < K) v = a.avail[(grow * T + t_i) * K + 4 * j + hq] ? 1u : 0u;
                        avbits |= v << j;
                    }
                    const long o = grow * T + t_i;
                    it_act = (ok && hq == 0) ? a.action[o] : 0;
                    it_lpo = (ok && hq == 0) ? a.logp_old[o] : 0.f;
                    it_adv = (ok && hq == 0) ? a.adv[o] : 0.f;
                }
            }
        }
    } else {
        // ================================================================ waves 4-7: workspace stores + the head, behind the chain
        for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
            const long row0 = tile * T32;
            __syncthreads();
            const int hrow = tid >> 2, hq = tid & 3;   // PPO math: 4 lanes per (row, step) item
            const int irow = hrow & 31;                // tile row of the item
            const long grow = row0 + irow;
            const bool rvalid = grow < R;
            const int e_row = rvalid ? (int)(grow / a.A) : 0;
            const int ag = (int)(grow - (long)e_row * a.A);
            const int eplen = rvalid ? a.ep_len[e_row] : 0;
            const float invA = 1.0f / (float)a.A;
            for (int s = 0; s < SV; ++s) {
                const int sp = s - 1;
                float* HBw = ((sp >> 1) & 1) ? HBb : HBa;           // the pass step sp belongs to
                float* HBp = (((s - 2) >> 1) & 1) ? HBb : HBa;      // the pass whose phases run in this step: (s - 2) >> 1 for s = 2p + 2, 2p + 3
                const int s0 = ((s - 2) >> 1) * 2;                  // its first step
                lds_barrier();
                PH(8);
                // ---------------------------------------------------------------- I0
                if (sp >= 0 && sp < 2 * NP) {
                    const bool live = sp < CL;
                    float* wsS = a.ws_act + ((long)sp * R + row0) * WS2;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int idx = tid + NTHREADS * q, r = idx >> 4, c4 = (idx & 15) * 4;
                        const int o = r * LDT + c4;
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (live && row0 + r < R) {
                            float* wp = wsS + (long)r * WS2 + c4;
#ifndef CM_X_NOST4
                            *reinterpret_cast<float4*>(wp + HP) = *reinterpret_cast<const float4*>(SR + o);
                            *reinterpret_cast<float4*>(wp + 2 * HP) = *reinterpret_cast<const float4*>(SZ + o);
                            *reinterpret_cast<float4*>(wp + 3 * HP) = *reinterpret_cast<const float4*>(SN + o);
                            *reinterpret_cast<float4*>(wp + 4 * HP) = *reinterpret_cast<const float4*>(SG + o);
#endif
                            v = *reinterpret_cast<const float4*>(hp + o);
                            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                        }
                        *reinterpret_cast<float4*>(HBw + (32 * (sp & 1) + r) * LDT + c4) = v;
                    }
                }
                PH(9);
                if ((s & 1) && s >= 3) {
                    // PPO clipped-surrogate head (arithmetic of k_gru_chunk_fwd): statistics + dlogits -> ls2
                    const int s_it = s0 + (hrow >> 5), t_it = a.t0 + s_it;
                    const bool ivalid = rvalid && s_it < CL;
                    const unsigned avbits = itav[hrow * 4 + hq];
                    const float4 itv = *reinterpret_cast<const float4*>(itf + 4 * hrow);
                    const int act = __float_as_int(itv.x);
                    const float lpo = itv.y, advv = itv.z;
                    float zreg[KJ];
#pragma unroll
                    for (int j = 0; j < KJ; ++j) zreg[j] = (4 * j + hq < K && ((avbits >> j) & 1u)) ? ls[hrow * KP + 4 * j + hq] : -1e9f;
                    const bool valid = ivalid && t_it < eplen;
                    float m = -INFINITY;
#pragma unroll
                    for (int j = 0; j < KJ; ++j) if (4 * j + hq < K) m = fmaxf(m, zreg[j]);
                    m = quad_max(m);
                    float ssum = 0.0f, pj[KJ];
#pragma unroll
                    for (int j = 0; j < KJ; ++j) { pj[j] = 0.f; if (4 * j + hq < K) { pj[j] = expf(zreg[j] - m); ssum += pj[j]; } }
                    ssum = quad_sum(ssum);
                    const float lse = m + logf(ssum), rs = 1.0f / ssum;
                    float ent = 0.f, lpa = 0.f;
#pragma unroll
                    for (int j = 0; j < KJ; ++j) if (4 * j + hq < K) {
                        const float lp = zreg[j] - lse;
                        pj[j] *= rs; ent -= pj[j] * lp;
                        if (4 * j + hq == act) lpa = lp;
                    }
                    ent = quad_sum(ent); lpa = quad_sum(lpa);
                    const float log_ratio = lpa - lpo, ratio = expf(log_ratio);
                    const float pg1 = advv * ratio, pg2 = advv * fminf(fmaxf(ratio, a.clip_lo), a.clip_hi);
                    const bool inr = (ratio >= a.clip_lo) && (ratio <= a.clip_hi);
                    float gsel;
                    if (pg1 < pg2) gsel = advv; else if (pg1 > pg2) gsel = inr ? advv : 0.f; else gsel = 0.5f * advv + (inr ? 0.5f * advv : 0.f);
                    if (valid && hq == 0) {
                        st_pg += invA * fminf(pg1, pg2); st_ent += invA * ent; st_kl += invA * ((ratio - 1.f) - log_ratio);
                        st_clip += (fabsf(ratio - 1.f) > a.clip_eps) ? invA : 0.f;
                        if (ag == 0) st_cnt += 1.f;
                    }
                    const float gr = gsel * ratio;
#pragma unroll
                    for (int j = 0; j < KJ; ++j) {
                        const int k = 4 * j + hq;
                        float d = 0.f;
                        if (k < K && valid && zreg[j] > -5e8f) {
                            const float lp = zreg[j] - lse;
                            d = invA * (-gr * ((k == act ? 1.f : 0.f) - pj[j]) + a.ent_coef * pj[j] * (lp + ent));
                        }
                        ls2[hrow * KP + k] = d;
                    }
                }
                PH(13);
                lds_barrier();
                PH(10);
                // ---------------------------------------------------------------- I1
                if (sp >= 0 && sp < CL) {
                    const float* X1 = (sp & 1) ? X1b : X1a;
                    float* wsS = a.ws_act + ((long)sp * R + row0) * WS2;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int idx = tid + NTHREADS * q, r = idx >> 4, c4 = (idx & 15) * 4;
                        if (row0 + r < R) {
                            float* wp = wsS + (long)r * WS2 + c4;
                            const int o = r * LDT + c4;
#ifndef CM_X_NOST2
                            *reinterpret_cast<float4*>(wp) = *reinterpret_cast<const float4*>(X1 + o);
                            *reinterpret_cast<float4*>(wp + 5 * HP) = *reinterpret_cast<const float4*>(hp + o);
#endif
                        }
                    }
                }
                PH(11);
                if (!(s & 1) && s >= 2) {
                    // logits on the 16x16x4 MFMA: wave w = items 16w..16w+15, one call per 16 head outputs
                    const int n = lane & 15, g4 = lane >> 4;
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) {
                        const f32x4 lg = head_logits_mfma(HBp + 16 * wave * LDT, wouts + 16 * ct * WLD);
                        const float bias = b2[16 * ct + n];
#pragma unroll
                        for (int q = 0; q < 4; ++q) ls[(16 * wave + 4 * g4 + q) * KP + 16 * ct + n] = lg[q] + bias;
                    }
                } else if ((s & 1) && s >= 3) {
                    // fc2 gradient: dW2 += dlogits^T relu(h') over the 64 items (wave = 16 hidden columns), db2 += column sums
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) colred_head16<KP>(accWo[ct], ls2, 16 * ct, HBp + 16 * wave);
                    {
                        constexpr int PARTS = NTHREADS / KP, RPP = TM / PARTS;
                        const int k = tid % KP, part = tid / KP;
                        float sb = 0.f;
#pragma unroll
                        for (int r = 0; r < RPP; ++r) sb += ls2[(part * RPP + r) * KP + k];
                        dbo += sb;
                    }
                    // dh_head = (dlogits W2) .* (h' > 0): wave (g, wn) = items 32 g .. (step s0 + g), columns 32 wn ..; straight to the workspace
                    f32x16 dh;
#pragma unroll
                    for (int i = 0; i < 16; ++i) dh[i] = 0.0f;
                    head_bwd_mfma<KP>(dh, ls2 + 32 * g * KP, wouts + 32 * wn);
                    const int ss = s0 + g;
                    if (ss < CL) {
                        float* wsD = a.ws_act + ((long)ss * R + row0) * WS2 + 6 * HP + col;
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int r = (i & 3) + 8 * (i >> 2) + 4 * h;
#ifndef CM_X_NODH
                            if (row0 + r < R) wsD[(long)r * WS2] = (HBp[(32 * g + r) * LDT + col] > 0.0f) ? dh[i] : 0.0f;
#else
                            if (dh[i] == 123.456f) wsD[(long)r * WS2] = dh[i];
#endif
                        }
                    }
                }
                PH(14);
                lds_barrier();
                PH(12);
                if (s < CL) { float* tmp = hp; hp = hn; hn = tmp; }
            }
        }
    }
    PH8_FLUSH();
    // ================================ this workgroup's partial row: fc2 gradient + statistics (the rest comes from k_gru2_bwd)
    float* out = a.partial + (size_t)blockIdx.x * a.PS;
    if (helper) {
        const int n = lane & 15, g4 = lane >> 4;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * ct + 4 * g4 + r, c = 16 * wave + n;
                if (k < K && c < H) out[off.W2 + k * H + c] = accWo[ct][r];
            }
    }
    __syncthreads();
    if (helper) red[tid] = dbo;  // [NTHREADS / KP parts][KP]
    __syncthreads();
    if (helper && tid < K) {
        float sb = 0.f;
#pragma unroll
        for (int q = 0; q < NTHREADS / KP; ++q) sb += red[q * KP + tid];
        out[off.b2 + tid] = sb;
    }
    float sv6[6] = {st_pg, st_ent, st_kl, st_clip, 0.f, st_cnt};
    __syncthreads();
    if (helper) {
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const float v = cm_wave_sum(sv6[q]);
            if (lane == 0) red[q * 4 + wave] = v;
        }
    }
    __syncthreads();
    if (helper && tid < CM_NUM_STATS) {
        float v = 0.f;
        if (tid < 6) v = red[tid * 4] + red[tid * 4 + 1] + red[tid * 4 + 2] + red[tid * 4 + 3];
        out[off.P + tid] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// k_gru2_fwdx: the forward sweep with the head on OTHER compute units.  Two waves of one SIMD share its matrix pipe and its VALU issue
// slots, so inside a workgroup the head only trades places with the recurrence (k_gru2_fwd8 gains 3-6 %); but at the batch sizes these
// sweeps serve the tiles do not fill the chip (config 5: 160 tiles on 256 CUs).  Here the launch has two kinds of workgroups:
//   * blockIdx <  nt  "chain" workgroups, one tile each: waves 0-3 run the recurrence, waves 4-7 store the six workspace tiles of step
//     s - 1 during step s and PUBLISH the step: h' leaves first, as agent-scope (sc1) stores, the wave waits for them (vmcnt, off the
//     chain) and sets its word of flags[tile][4] to {launch tag, steps published}.  No fence: an agent-scope release would write back the
//     whole L2 (cm_optim.hip has the measurements); the data the consumer needs is written through by the stores themselves.
//   * blockIdx >= nt  "head" workgroups on the CUs the tiles leave idle: each half (4 waves) follows one tile per round, waits for the
//     flags of the two steps of a pass, reads their h' rows with agent-scope loads and runs the pass of k_gru2_fwd (logits -> PPO loss ->
//     dlogits -> fc2 gradient -> dh_head) -- dh_head goes to the workspace for the backward sweep (next launch), the fc2 gradient and
//     the statistics to the partial row of the head workgroup (the chain workgroups zero theirs).
// Chain workgroups never wait for head workgroups and are dispatched first (lower blockIdx), so the launch cannot deadlock whatever
// shares the device; a head workgroup gives up after a bounded number of polls and poisons its statistics with NaN.
// Numerics: the arithmetic of k_gru2_fwd per item; the fc2 gradient / statistics are summed per head workgroup instead of per tile
// (a different association of the same terms; everything else bit-identical).
struct GruXArgs { unsigned long long* flags; unsigned tag; int nt, nh; unsigned long long* dhq;  // dhq: [CL][R][64] {dh_t, tag} words (backward)
                  float* gi; };  // [CL][nt][GI_UNIT] W_ih accumulators of k_gru2_pre (split forward sweep; NULL: the chain computes them itself)
constexpr int GX_RED = 4 * 64 * 8 + 256;   // head workgroup's reduction scratch: half 1's fc2 accumulators (4 waves x 64 lanes x 8), b2 parts, statistics
constexpr int g2fx_lds_floats(int KP) {
    // chain workgroup: 9 tiles (10 in the split form); head workgroup: per half HB + ls + ls2, shared wouts + b2 + red
    // (split forward sweep, in the workgroup: 8 tiles + 24 KB of W_ih accumulators; as a pre-pass launch: 10 tiles)
    return (8 * T32 * LDT + GI_UNIT > 2 * (TM * LDT + 2 * TM * KP) + KP * WLD + KMAX + GX_RED) ? 8 * T32 * LDT + GI_UNIT : 2 * (TM * LDT + 2 * TM * KP) + KP * WLD + KMAX + GX_RED;
}
inline size_t gru2_fwdx_lds_bytes(int KP) {
    const size_t b = (size_t)g2fx_lds_floats(KP) * sizeof(float);
    return b > 84 * 1024 ? b : 84 * 1024;   // > half of the CU's LDS: one workgroup per CU whatever the register count
}
__device__ __forceinline__ void st_agent64(float* p, float x, float y) {
    unsigned long long v = ((unsigned long long)__float_as_uint(y) << 32) | __float_as_uint(x);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld_agent64(const float* p) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((unsigned)v), __uint_as_float((unsigned)(v >> 32)));
}
constexpr int GX_POLLS = 1 << 18;   // ~0.3 s of polling before a head workgroup gives up (a chunk sweep takes ~0.1 ms)

template <int KP>
__device__ __forceinline__ void gru2_head_wg(const GruArgs& a, const GruXArgs& x, float* smem) {
    constexpr int KJ = KP / 4, NCT = KP / 16;
    const GruOff off = gru_offsets(a.din, a.H, a.K);
    const int hg = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));   // half of the workgroup: its own tile, its own LDS buffers
    const int tid = threadIdx.x & (NTHREADS - 1), lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, g = wave >> 1, h = lane >> 5, lc = lane & 31;
    const int col = 32 * wn + lc;
    float* p = smem + hg * (TM * LDT + 2 * TM * KP);
    float* HB = p; p += TM * LDT;
    float* ls = p; p += TM * KP;
    float* ls2 = p;
    p = smem + 2 * (TM * LDT + 2 * TM * KP);
    float* wouts = p; p += KP * WLD;
    float* b2 = p; p += KMAX;
    float* red = p;   // GX_RED
    const int H = a.H, K = a.K, T = a.T, CL = a.t1 - a.t0, NP = (CL + 1) >> 1;
    const long R = (long)a.E * a.A;
    for (int i = threadIdx.x; i < KP * WLD; i += NT8) {
        const int k = i / WLD, c = i % WLD;
        wouts[i] = (c < H && k < K) ? a.params[off.W2 + k * H + c] : 0.0f;
    }
    for (int i = threadIdx.x; i < KMAX; i += NT8) b2[i] = (i < K) ? a.params[off.b2 + i] : 0.0f;
    PH_DECL
    float st_pg = 0.f, st_ent = 0.f, st_kl = 0.f, st_clip = 0.f, st_cnt = 0.f;
    f32x4 accWo[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) accWo[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbo = 0.f;
    bool dead = false;   // a poll ran out: stop waiting, poison the statistics
    const int j = (int)blockIdx.x - x.nt;
    const int rounds = (x.nt + 2 * x.nh - 1) / (2 * x.nh);
    const int hrow = tid >> 2, hq = tid & 3;
    const int irow = hrow & 31;
    const float invA = 1.0f / (float)a.A;
    for (int rd = 0; rd < rounds; ++rd) {
        const int q = (rd * x.nh + j) * 2 + hg;          // this half's tile of the round
        const bool tvalid = q < x.nt;
        const long row0 = (long)q * T32;
        const long grow = row0 + irow;
        const bool rvalid = tvalid && grow < R;
        const int e_row = rvalid ? (int)(grow / a.A) : 0;
        const int ag = (int)(grow - (long)e_row * a.A);
        const int eplen = rvalid ? a.ep_len[e_row] : 0;
        const unsigned long long* fl = x.flags + (size_t)(tvalid ? q : 0) * 4;
        for (int pss = 0; pss < NP; ++pss) {
            const int s0 = 2 * pss;
            // per-item inputs (4 lanes per (row, step) item)
            const int s_it = s0 + (hrow >> 5), t_it = a.t0 + s_it;
            const bool ivalid = rvalid && s_it < CL;
            unsigned char avb[KJ];
#pragma unroll
            for (int jj = 0; jj < KJ; ++jj) {
                avb[jj] = 1;
                if (ivalid && 4 * jj + hq < K) avb[jj] = a.avail[(grow * T + t_it) * K + 4 * jj + hq];
            }
            const long o = grow * T + t_it;
            const int act = ivalid ? a.action[o] : 0;
            const float lpo = ivalid ? a.logp_old[o] : 0.f, advv = ivalid ? a.adv[o] : 0.f;
            // the two steps of the pass as the chain workgroup publishes them (each wave polls the four words itself): the rows of the
            // first step are on their way while the second is still being computed.  Agent-scope loads: the rows were written by
            // another workgroup in this launch
            float4 hv[4];
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const unsigned want = (unsigned)(s0 + half + 1 < CL ? s0 + half + 1 : CL);
                if (tvalid && !dead) {
                    for (int it = 0;; ++it) {
                        bool ok = true;
                        if (lane < 4) {
                            const unsigned long long v = __hip_atomic_load(fl + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = (unsigned)(v >> 32) == x.tag && (unsigned)v >= want;
                        }
                        if (__builtin_amdgcn_ballot_w64(!ok) == 0ull) break;
                        if (it >= GX_POLLS) { dead = true; break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                }
#pragma unroll
                for (int qq = 2 * half; qq < 2 * half + 2; ++qq) {
                    const int idx = tid + NTHREADS * qq, it = idx >> 4, c4 = (idx & 15) * 4;
                    const int ss = s0 + (it >> 5), r = it & 31;   // it >> 5 == half
                    hv[qq] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (tvalid && !dead && ss < CL && row0 + r < R) {
                        const float* src = a.ws_act + ((long)ss * R + row0 + r) * WS2 + 5 * HP + c4;
                        const float2 lo = ld_agent64(src), hi = ld_agent64(src + 2);
                        hv[qq] = make_float4(fmaxf(lo.x, 0.f), fmaxf(lo.y, 0.f), fmaxf(hi.x, 0.f), fmaxf(hi.y, 0.f));
                    }
                }
            }
            PH(0);
            lds_barrier();  // HB / ls / ls2 of the previous pass are dead
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int idx = tid + NTHREADS * qq, it = idx >> 4, c4 = (idx & 15) * 4;
                *reinterpret_cast<float4*>(HB + it * LDT + c4) = hv[qq];
            }
            PH(1);
            lds_barrier();
            {   // logits on the 16x16x4 MFMA: wave w = items 16w..16w+15, one call per 16 head outputs
                const int n = lane & 15, g4 = lane >> 4;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const f32x4 lg = head_logits_mfma(HB + 16 * wave * LDT, wouts + 16 * ct * WLD);
                    const float bias = b2[16 * ct + n];
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) ls[(16 * wave + 4 * g4 + qq) * KP + 16 * ct + n] = lg[qq] + bias;
                }
            }
            PH(2);
            lds_barrier();
            {   // PPO clipped-surrogate head (arithmetic of k_gru_chunk_fwd): statistics + dlogits -> ls2
                float zreg[KJ];
#pragma unroll
                for (int jj = 0; jj < KJ; ++jj) zreg[jj] = (4 * jj + hq < K && avb[jj]) ? ls[hrow * KP + 4 * jj + hq] : -1e9f;
                const bool valid = ivalid && t_it < eplen;
                float m = -INFINITY;
#pragma unroll
                for (int jj = 0; jj < KJ; ++jj) if (4 * jj + hq < K) m = fmaxf(m, zreg[jj]);
                m = quad_max(m);
                float ssum = 0.0f, pj[KJ];
#pragma unroll
                for (int jj = 0; jj < KJ; ++jj) { pj[jj] = 0.f; if (4 * jj + hq < K) { pj[jj] = expf(zreg[jj] - m); ssum += pj[jj]; } }
                ssum = quad_sum(ssum);
                const float lse = m + logf(ssum), rs = 1.0f / ssum;
                float ent = 0.f, lpa = 0.f;
#pragma unroll
                for (int jj = 0; jj < KJ; ++jj) if (4 * jj + hq < K) {
                    const float lp = zreg[jj] - lse;
                    pj[jj] *= rs; ent -= pj[jj] * lp;
                    if (4 * jj + hq == act) lpa = lp;
                }
                ent = quad_sum(ent); lpa = quad_sum(lpa);
                const float log_ratio = lpa - lpo, ratio = expf(log_ratio);
                const float pg1 = advv * ratio, pg2 = advv * fminf(fmaxf(ratio, a.clip_lo), a.clip_hi);
                const bool inr = (ratio >= a.clip_lo) && (ratio <= a.clip_hi);
                float gsel;
                if (pg1 < pg2) gsel = advv; else if (pg1 > pg2) gsel = inr ? advv : 0.f; else gsel = 0.5f * advv + (inr ? 0.5f * advv : 0.f);
                if (valid && hq == 0) {
                    st_pg += invA * fminf(pg1, pg2); st_ent += invA * ent; st_kl += invA * ((ratio - 1.f) - log_ratio);
                    st_clip += (fabsf(ratio - 1.f) > a.clip_eps) ? invA : 0.f;
                    if (ag == 0) st_cnt += 1.f;
                }
                const float gr = gsel * ratio;
#pragma unroll
                for (int jj = 0; jj < KJ; ++jj) {
                    const int k = 4 * jj + hq;
                    float d = 0.f;
                    if (k < K && valid && zreg[jj] > -5e8f) {
                        const float lp = zreg[jj] - lse;
                        d = invA * (-gr * ((k == act ? 1.f : 0.f) - pj[jj]) + a.ent_coef * pj[jj] * (lp + ent));
                    }
                    ls2[hrow * KP + k] = d;
                }
            }
            PH(3);
            lds_barrier();
            // fc2 gradient: dW2 += dlogits^T relu(h') over the 64 items (wave = 16 hidden columns), db2 += column sums
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) colred_head16<KP>(accWo[ct], ls2, 16 * ct, HB + 16 * wave);
            {
                constexpr int PARTS = NTHREADS / KP, RPP = TM / PARTS;
                const int k = tid % KP, part = tid / KP;
                float sb = 0.f;
#pragma unroll
                for (int r = 0; r < RPP; ++r) sb += ls2[(part * RPP + r) * KP + k];
                dbo += sb;
            }
            // dh_head = (dlogits W2) .* (h' > 0): wave (g, wn) = items 32 g .. (step s0 + g), columns 32 wn ..
            f32x16 dh;
#pragma unroll
            for (int i = 0; i < 16; ++i) dh[i] = 0.0f;
            head_bwd_mfma<KP>(dh, ls2 + 32 * g * KP, wouts + 32 * wn);
            lds_barrier();  // every read of HB (fc2 gradient) is done: it becomes the dh_head tile
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float* qd = HB + (32 * g + (i & 3) + 8 * (i >> 2) + 4 * h) * LDT + col;
                *qd = (*qd > 0.0f) ? dh[i] : 0.0f;
            }
            lds_barrier();
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int idx = tid + NTHREADS * qq, it = idx >> 4, c4 = (idx & 15) * 4;
                const int ss = s0 + (it >> 5), r = it & 31;
                if (tvalid && ss < CL && row0 + r < R)
                    *reinterpret_cast<float4*>(a.ws_act + ((long)ss * R + row0 + r) * WS2 + 6 * HP + c4) = *reinterpret_cast<const float4*>(HB + it * LDT + c4);
            }
            PH(4);
        }
    }
#ifdef CM_PHASE_PROF
    if (a.prof && threadIdx.x == 0) { for (int i_ = 0; i_ < 8; ++i_) a.prof[(size_t)blockIdx.x * 16 + i_] = ph_[i_]; }
#endif
    // ================================ partial row of this head workgroup: fc2 gradient + statistics of the tiles its two halves followed
    float* out = a.partial + (size_t)j * a.PS;
    __syncthreads();
    {
        const int n = lane & 15, g4 = lane >> 4;
        float* rw = red + (size_t)wave * 64 * 4 * NCT;   // half 1 hands its accumulators to half 0
        if (hg == 1) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) rw[(ct * 4 + r) * 64 + lane] = accWo[ct][r];
        }
        __syncthreads();
        if (hg == 0) {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * ct + 4 * g4 + r, c = 16 * wave + n;
                    if (k < K && c < H) out[off.W2 + k * H + c] = accWo[ct][r] + rw[(ct * 4 + r) * 64 + lane];
                }
        }
    }
    __syncthreads();
    red[threadIdx.x] = dbo;  // [2 halves][NTHREADS / KP parts][KP]
    __syncthreads();
    if (threadIdx.x < K) {
        float sb = 0.f;
#pragma unroll
        for (int qq = 0; qq < NT8 / KP; ++qq) sb += red[qq * KP + threadIdx.x];
        out[off.b2 + threadIdx.x] = sb;
    }
    float sv6[6] = {st_pg, st_ent, st_kl, st_clip, 0.f, st_cnt};
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < 6; ++qq) {
        const float v = cm_wave_sum(sv6[qq]);
        if (lane == 0) red[qq * 8 + (threadIdx.x >> 6)] = v;
    }
    if (threadIdx.x == 0) red[64] = 0.f;   // (statistics occupy red[0 .. 47])
    __syncthreads();
    if (dead && lane == 0) red[64] = 1.f;
    __syncthreads();
    if (threadIdx.x < CM_NUM_STATS) {
        float v = 0.f;
        if (threadIdx.x < 6) {
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) v += red[threadIdx.x * 8 + w8];
        }
        if (red[64] != 0.f) v = __builtin_nanf("");
        out[off.P + threadIdx.x] = v;
    }
}

// ---- the h-independent half of the forward sweep as a throughput launch over all (tile, step) units of the chunk (gru2_pre_unit,
// cm_gru_step2.h): x1 = relu(fc1(obs_t)) goes to its workspace slot (what the chain's helper waves used to store), the three W_ih products
// to x.gi.  Two workgroups per CU (17 KB of LDS, 64 weight registers): one's obs staging and stores under the other's products.
template <bool WV>
__global__ __launch_bounds__(NTHREADS, 2) void k_gru2_pre(const GruArgs a, const GruXArgs x) {
    __shared__ __attribute__((aligned(16))) float X0[T32 * LDT];
    __shared__ __attribute__((aligned(16))) float X1[T32 * LDT];
    const GruOff off = gru_offsets(a.din, a.H, a.K);
    const int tid = threadIdx.x, din = a.din, T = a.T, CL = a.t1 - a.t0;
    const long R = (long)a.E * a.A;
    const long units = (long)x.nt * CL;
    G2W w;
    g2_load_weights_x<WV>(w, a.params, off, din, a.H);
    X32 xr;
    long u = blockIdx.x;
    if (u < units) x32_load(xr, a.obs + (long)(a.t0 + (int)(u / x.nt)) * din, (u % x.nt) * T32, R, (long)T * din, din);
    for (; u < units; u += gridDim.x) {
        const int s = (int)(u / x.nt);
        const long row0 = (u % x.nt) * T32;
        x32_store(X0, xr);  // (the previous unit's readers of X0 passed its "x1 complete" barrier; X1's pass the barrier below)
        const long un = u + gridDim.x;
        if (un < units) x32_load(xr, a.obs + (long)(a.t0 + (int)(un / x.nt)) * din, (un % x.nt) * T32, R, (long)T * din, din);
        lds_barrier();
        gru2_pre_unit(w, X0, X1, x.gi + (size_t)u * GI_UNIT, din);
        float* wsS = a.ws_act + ((long)s * R + row0) * WS2;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = tid + NTHREADS * q, r = idx >> 4, c4 = (idx & 15) * 4;
            if (row0 + r < R) *reinterpret_cast<float4*>(wsS + (long)r * WS2 + c4) = *reinterpret_cast<const float4*>(X1 + r * LDT + c4);
        }
    }
}

// PRE: the split sweep -- fc1 and the W_ih products of every step come from k_gru2_pre (x.gi), the chain keeps W_hh h + gates: one barrier
// per step, the saved tiles double-buffered so that the helper waves store step s - 1 while the recurrence writes step s.
// PRE == 2 (round 6, the default): the same split INSIDE the workgroup -- waves 4-7, which only stored finished tiles so far, run the
// h-independent half of step s + 1 (fc1, then the three W_ih products) while waves 0-3 run the h-dependent half of step s, handed over through
// 24 KB of LDS in lane order.  Two barriers per step, arranged so that the two halves want different pipes at the same time:
//     interval A:  recurrence: W_hh h_{t-1} (96 MFMAs)        | helpers: tiles of step s - 1 -> workspace, publish; fc1 of step s + 1 (24 MFMAs)
//     interval B:  recurrence: gate math (VALU), h', tiles     | helpers: W_ih x1 of step s + 1 (96 MFMAs) -> LDS, x1 -> workspace, next obs tile
// No extra launch, no extra workgroup role; the MFMA sequences and operands are those of gru2_step: bit-identical results.
template <bool WV, int KP, int PRE = 0>
__global__ __launch_bounds__(NT8) void k_gru2_fwdx(const GruArgs a, const GruXArgs x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef CM_X_NOHEAD
    if ((int)blockIdx.x >= x.nt) return;
#endif
    if ((int)blockIdx.x >= x.nt) { gru2_head_wg<KP>(a, x, smem); return; }
    const GruOff off = gru_offsets(a.din, a.H, a.K);
    if constexpr (PRE == 2) {
        float* hp = smem; float* hn = smem + T32 * LDT;
        float* S = smem + 2 * T32 * LDT;    // SR, SZ, SN, SG
        float* X0 = smem + 6 * T32 * LDT;   // obs tile of the step the helpers work on
        float* X1 = smem + 7 * T32 * LDT;   // its x1
        float* GI = smem + 8 * T32 * LDT;   // [4 waves][6][64 lanes][4]: the W_ih accumulators of the next step, lane order
        const bool helper = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;
        const int tid = threadIdx.x & (NTHREADS - 1), lane = tid & 63, wave = tid >> 6, n = lane & 15, g = lane >> 4;
        const int H = a.H, din = a.din, T = a.T, CL = a.t1 - a.t0;
        const long R = (long)a.E * a.A;
        const long row0 = (long)blockIdx.x * T32;
        f32x4* gil = reinterpret_cast<f32x4*>(GI + wave * (6 * 64 * 4)) + lane;
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        PH_DECL
        if (!helper) {
            G2W w;
            g2_load_weights_h<WV>(w, a.params, off, H);
            for (int i = tid; i < T32 * HP; i += NTHREADS) {
                const int r = i >> 6, c = i & 63;
                hp[r * LDT + c] = (row0 + r < R && c < H && a.h_in) ? a.h_in[(row0 + r) * H + c] : 0.0f;
            }
            lds_barrier(); lds_barrier();   // the helpers' prologue: x1 and the W_ih products of step 0
            for (int s = 0; s <= CL; ++s) {
                lds_barrier();              // top: h_{t-1}, the tiles of step s - 1 and GI(s) are complete
                PH(0);
                f32x4 hr[2] = {zero, zero}, hz[2] = {zero, zero}, hnn[2] = {zero, zero};
                f32x4 gi[6];
                if (s < CL) {
#pragma unroll
                    for (int k = 0; k < 6; ++k) gi[k] = gil[64 * k];
                    g2_prod3(hr, hz, hnn, hp, w.hr, w.hz, w.hn);
                }
                PH(1);
                lds_barrier();              // mid: GI is read, the helpers are done with the tiles of step s - 1
                PH(2);
                if (s < CL) {
                    const int col = 16 * wave + n;
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
                            S[o] = r; S[T32 * LDT + o] = z; S[2 * T32 * LDT + o] = nn; S[3 * T32 * LDT + o] = ghn;
                        }
                    float* tmp = hp; hp = hn; hn = tmp;
                } else if (a.h_out) {
                    for (int i = tid; i < T32 * HP; i += NTHREADS) {
                        const int r = i >> 6, c = i & 63;
                        if (row0 + r < R && c < H) a.h_out[(row0 + r) * H + c] = hp[r * LDT + c];
                    }
                }
                PH(3);
            }
        } else {
            // ---- helpers: the h-independent half, one step ahead; their own obs staging (x32_* of the recurrence waves, same row mapping)
            float w1[16], wxr[16], wxz[16], wxn[16];
            const int c0 = 16 * wave, col = c0 + n;
            load_nt16_regs<false>(w1, a.params + off.W1, c0, H, din, din);
            load_nt16_regs<WV>(wxr, a.params + off.Wih, c0, H, H, H);
            load_nt16_regs<WV>(wxz, a.params + off.Wih + H * H, c0, H, H, H);
            load_nt16_regs<WV>(wxn, a.params + off.Wih + 2 * H * H, c0, H, H, H);
            const float b1 = (col < H) ? a.params[off.b1 + col] : 0.0f;
            const int r0 = wave * 8;
            auto obs_load = [&](X32& xx, int t) {
                const float* p = a.obs + (long)t * din + (row0 + r0) * ((long)T * din) + lane;
#pragma unroll
                for (int i = 0; i < 8; ++i) xx.v[i] = (row0 + r0 + i < R && lane < din) ? p[i * ((long)T * din)] : 0.0f;
            };
            auto obs_store = [&](const X32& xx) {
#pragma unroll
                for (int i = 0; i < 8; ++i) X0[(r0 + i) * LDT + lane] = xx.v[i];
            };
            auto half_a = [&]() {   // x1 = relu(fc1(obs)) of the step whose obs tile is in X0
                f32x4 a1[2] = {zero, zero};
                g2_prod1(a1, X0, w1, (din + 15) >> 4);
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) X1[(16 * rb + 4 * g + q) * LDT + col] = fmaxf(a1[rb][q] + b1, 0.0f);
            };
            auto half_b = [&](int st) {   // the three W_ih products of step st -> GI (lane order); x1 -> its workspace slot
                f32x4 xr[2] = {zero, zero}, xz[2] = {zero, zero}, xn[2] = {zero, zero};
                g2_prod3(xr, xz, xn, X1, wxr, wxz, wxn);
                gil[0] = xr[0]; gil[64] = xr[1]; gil[128] = xz[0]; gil[192] = xz[1]; gil[256] = xn[0]; gil[320] = xn[1];
                float* wsS = a.ws_act + ((long)st * R + row0) * WS2;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int idx = tid + NTHREADS * q, r = idx >> 4, c4 = (idx & 15) * 4;
                    if (row0 + r < R) *reinterpret_cast<float4*>(wsS + (long)r * WS2 + c4) = *reinterpret_cast<const float4*>(X1 + r * LDT + c4);
                }
            };
            unsigned long long* fl = x.flags + (size_t)blockIdx.x * 4 + wave;
            X32 xo;
            obs_load(xo, a.t0);
            obs_store(xo);
            if (CL > 1) obs_load(xo, a.t0 + 1);
            lds_barrier();
            half_a();
            lds_barrier();
            half_b(0);
            if (CL > 1) obs_store(xo);
            if (CL > 2) obs_load(xo, a.t0 + 2);
            for (int s = 0; s <= CL; ++s) {
                const int sp = s - 1;
                lds_barrier();              // top
                PH(8);
                float* wsS = a.ws_act + ((long)(sp < 0 ? 0 : sp) * R + row0) * WS2;
                if (sp >= 0) {              // h' of step sp leaves first (the head workgroups wait for it) ...
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int idx = tid + NTHREADS * q, r = idx >> 4, c4 = (idx & 15) * 4;
                        if (row0 + r < R) {
                            const float4 hv = *reinterpret_cast<const float4*>(hp + r * LDT + c4);
                            float* wp = wsS + (long)r * WS2 + 5 * HP + c4;
                            st_agent64(wp, hv.x, hv.y);
                            st_agent64(wp + 2, hv.z, hv.w);
                        }
                    }
                }
                if (s + 1 < CL) half_a();   // ... fc1 of step s + 1 runs under the stores' way to memory (its obs tile went to X0 in the previous interval B) ...
                if (sp >= 0) {              // ... then the step is published, and the four saved tiles follow
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(fl, ((unsigned long long)x.tag << 32) | (unsigned)(sp + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int idx = tid + NTHREADS * q, r = idx >> 4, c4 = (idx & 15) * 4;
                        if (row0 + r < R) {
                            float* wp = wsS + (long)r * WS2 + c4;
                            const int o = r * LDT + c4;
                            *reinterpret_cast<float4*>(wp + HP) = *reinterpret_cast<const float4*>(S + o);
                            *reinterpret_cast<float4*>(wp + 2 * HP) = *reinterpret_cast<const float4*>(S + T32 * LDT + o);
                            *reinterpret_cast<float4*>(wp + 3 * HP) = *reinterpret_cast<const float4*>(S + 2 * T32 * LDT + o);
                            *reinterpret_cast<float4*>(wp + 4 * HP) = *reinterpret_cast<const float4*>(S + 3 * T32 * LDT + o);
                        }
                    }
                }
                PH(9);
                lds_barrier();              // mid
                PH(10);
                if (s + 1 < CL) {
                    half_b(s + 1);
                    if (s + 2 < CL) obs_store(xo);
                    if (s + 3 < CL) obs_load(xo, a.t0 + s + 3);
                }
                if (s < CL) { float* tmp = hp; hp = hn; hn = tmp; }
                PH(11);
            }
        }
        PH8_FLUSH();
        if ((int)blockIdx.x >= x.nh) {
            float* out = a.partial + (size_t)blockIdx.x * a.PS;
            for (int i = threadIdx.x; i < a.K * H; i += NT8) out[off.W2 + i] = 0.0f;
            if (threadIdx.x < a.K) out[off.b2 + threadIdx.x] = 0.0f;
            if (threadIdx.x < CM_NUM_STATS) out[off.P + threadIdx.x] = 0.0f;
        }
        return;
    }
    if constexpr (PRE == 1) {
        float* hp = smem; float* hn = smem + T32 * LDT;
        float* S0 = smem + 2 * T32 * LDT;   // [2 sets][SR, SZ, SN, SG]
        const bool helper = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;
        const int tid = threadIdx.x & (NTHREADS - 1), lane = tid & 63, wave = tid >> 6;
        const int H = a.H, CL = a.t1 - a.t0;
        const long R = (long)a.E * a.A;
        const long row0 = (long)blockIdx.x * T32;
        if (!helper) {
            G2W w;
            g2_load_weights_h<WV>(w, a.params, off, H);
            for (int i = tid; i < T32 * HP; i += NTHREADS) {
                const int r = i >> 6, c = i & 63;
                hp[r * LDT + c] = (row0 + r < R && c < H && a.h_in) ? a.h_in[(row0 + r) * H + c] : 0.0f;
            }
            f32x4 gq[6];
            gru2_gi_load(gq, x.gi + (size_t)blockIdx.x * GI_UNIT);
            for (int s = 0; s <= CL; ++s) {
                lds_barrier();  // hp = h_{t-1} complete (and the saved tiles of step s - 1)
                if (s < CL) {
                    f32x4 gc[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) gc[k] = gq[k];
                    if (s + 1 < CL) gru2_gi_load(gq, x.gi + ((size_t)(s + 1) * x.nt + blockIdx.x) * GI_UNIT);  // a step ahead
                    float* S = S0 + (s & 1) * 4 * T32 * LDT;
                    gru2_step_h<true>(w, gc, hp, hn, S, S + T32 * LDT, S + 2 * T32 * LDT, S + 3 * T32 * LDT, H);
                    float* tmp = hp; hp = hn; hn = tmp;
                } else if (a.h_out) {
                    for (int i = tid; i < T32 * HP; i += NTHREADS) {
                        const int r = i >> 6, c = i & 63;
                        if (row0 + r < R && c < H) a.h_out[(row0 + r) * H + c] = hp[r * LDT + c];
                    }
                }
            }
        } else {
            unsigned long long* fl = x.flags + (size_t)blockIdx.x * 4 + wave;
            for (int s = 0; s <= CL; ++s) {
                const int sp = s - 1;
                lds_barrier();
                if (sp >= 0) {  // the tiles of step sp leave during step s; h' first (the head workgroups wait for it), then publish
                    float* wsS = a.ws_act + ((long)sp * R + row0) * WS2;
                    const float* S = S0 + (sp & 1) * 4 * T32 * LDT;
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int idx = tid + NTHREADS * q, r = idx >> 4, c4 = (idx & 15) * 4;
                        if (row0 + r < R) {
                            const float4 hv = *reinterpret_cast<const float4*>(hp + r * LDT + c4);
                            float* wp = wsS + (long)r * WS2 + 5 * HP + c4;
                            st_agent64(wp, hv.x, hv.y);
                            st_agent64(wp + 2, hv.z, hv.w);
                        }
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(fl, ((unsigned long long)x.tag << 32) | (unsigned)(sp + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const int idx = tid + NTHREADS * q, r = idx >> 4, c4 = (idx & 15) * 4;
                        if (row0 + r < R) {
                            float* wp = wsS + (long)r * WS2 + c4;
                            const int o = r * LDT + c4;
                            *reinterpret_cast<float4*>(wp + HP) = *reinterpret_cast<const float4*>(S + o);
                            *reinterpret_cast<float4*>(wp + 2 * HP) = *reinterpret_cast<const float4*>(S + T32 * LDT + o);
                            *reinterpret_cast<float4*>(wp + 3 * HP) = *reinterpret_cast<const float4*>(S + 2 * T32 * LDT + o);
                            *reinterpret_cast<float4*>(wp + 4 * HP) = *reinterpret_cast<const float4*>(S + 3 * T32 * LDT + o);
                        }
                    }
                }
                if (s < CL) { float* tmp = hp; hp = hn; hn = tmp; }
            }
        }
        if ((int)blockIdx.x >= x.nh) {
            float* out = a.partial + (size_t)blockIdx.x * a.PS;
            for (int i = threadIdx.x; i < a.K * H; i += NT8) out[off.W2 + i] = 0.0f;
            if (threadIdx.x < a.K) out[off.b2 + threadIdx.x] = 0.0f;
            if (threadIdx.x < CM_NUM_STATS) out[off.P + threadIdx.x] = 0.0f;
        }
        return;
    }
    float* p = smem;
    float* X0 = p; p += T32 * LDT;    // obs tile of the step
    float* X1a = p; p += T32 * LDT;   // x1 = relu(fc1(obs)), even steps
    float* X1b = p; p += T32 * LDT;   //                     odd steps
    float* hp = p; p += T32 * LDT;    // h_{t-1}
    float* hn = p; p += T32 * LDT;    // h_t
    float* SR = p; p += T32 * LDT;
    float* SZ = p; p += T32 * LDT;
    float* SN = p; p += T32 * LDT;
    float* SG = p;                    // W_hn h + b_hn
    const bool helper = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;   // wave-uniform, and the compiler must know it
    const int tid = threadIdx.x & (NTHREADS - 1), lane = tid & 63, wave = tid >> 6;
    const int H = a.H, din = a.din, T = a.T, CL = a.t1 - a.t0;
    const long R = (long)a.E * a.A;
    const long row0 = (long)blockIdx.x * T32;
    PH_DECL
    if (!helper) {
        // ================================================================ waves 0-3: the recurrence, 3 barriers per step (gru2_step)
        G2W w;
        g2_load_weights<WV>(w, a.params, off, din, H);
        for (int i = tid; i < T32 * HP; i += NTHREADS) {
            const int r = i >> 6, c = i & 63;
            hp[r * LDT + c] = (row0 + r < R && c < H && a.h_in) ? a.h_in[(row0 + r) * H + c] : 0.0f;
        }
        X32 xr;
        x32_load(xr, a.obs + (long)a.t0 * din, row0, R, (long)T * din, din);
        x32_store(X0, xr);
        if (CL > 1) x32_load(xr, a.obs + (long)(a.t0 + 1) * din, row0, R, (long)T * din, din);
        for (int s = 0; s <= CL; ++s) {
            lds_barrier();  // X0 = obs(t), hp = h_{t-1}
            PH(0);
            if (s < CL) {
                gru2_step<true>(w, X0, (s & 1) ? X1b : X1a, hp, hn, SR, SZ, SN, SG, din, H);
                PH(1);
#ifndef CM_X_NOOBS
#ifndef CM_X_NOOBSST
                if (s + 1 < CL) x32_store(X0, xr);
#endif
#ifndef CM_X_NOOBSLD
                if (s + 2 < CL) x32_load(xr, a.obs + (long)(a.t0 + s + 2) * din, row0, R, (long)T * din, din);
#endif
#endif
                float* tmp = hp; hp = hn; hn = tmp;
                PH(2);
            } else {
                if (a.h_out)
                    for (int i = tid; i < T32 * HP; i += NTHREADS) {
                        const int r = i >> 6, c = i & 63;
                        if (row0 + r < R && c < H) a.h_out[(row0 + r) * H + c] = hp[r * LDT + c];
                    }
                lds_barrier();
                lds_barrier();
                PH(3);
            }
        }
    } else {
        // ================================================================ waves 4-7: the tiles of step s - 1 leave during step s; publish
        unsigned long long* fl = x.flags + (size_t)blockIdx.x * 4 + wave;
        for (int s = 0; s <= CL; ++s) {
            const int sp = s - 1;
            float* wsS = a.ws_act + ((long)sp * R + row0) * WS2;
            lds_barrier();
            PH(8);
#ifdef CM_X_NOHEAD
            if (false) {
#else
            if (sp >= 0) {
#endif
                // h' first, alone: it is what the head workgroups wait for.  The wave waits for its own stores (the older ones are a
                // step old) and publishes the step; everything else follows
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int idx = tid + NTHREADS * q, r = idx >> 4, c4 = (idx & 15) * 4;
                    if (row0 + r < R) {
                        const float4 hv = *reinterpret_cast<const float4*>(hp + r * LDT + c4);
                        float* wp = wsS + (long)r * WS2 + 5 * HP + c4;
                        st_agent64(wp, hv.x, hv.y);
                        st_agent64(wp + 2, hv.z, hv.w);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(fl, ((unsigned long long)x.tag << 32) | (unsigned)(sp + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int idx = tid + NTHREADS * q, r = idx >> 4, c4 = (idx & 15) * 4;
                    if (row0 + r < R) {
                        float* wp = wsS + (long)r * WS2 + c4;
                        const int o = r * LDT + c4;
                        *reinterpret_cast<float4*>(wp + HP) = *reinterpret_cast<const float4*>(SR + o);
                        *reinterpret_cast<float4*>(wp + 2 * HP) = *reinterpret_cast<const float4*>(SZ + o);
                        *reinterpret_cast<float4*>(wp + 3 * HP) = *reinterpret_cast<const float4*>(SN + o);
                        *reinterpret_cast<float4*>(wp + 4 * HP) = *reinterpret_cast<const float4*>(SG + o);
                    }
                }
            }
            PH(9);
            lds_barrier();
            PH(10);
#ifdef CM_X_NOHEAD
            if (false) {
#else
            if (sp >= 0) {
#endif
                const float* X1 = (sp & 1) ? X1b : X1a;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int idx = tid + NTHREADS * q, r = idx >> 4, c4 = (idx & 15) * 4;
                    if (row0 + r < R) *reinterpret_cast<float4*>(wsS + (long)r * WS2 + c4) = *reinterpret_cast<const float4*>(X1 + r * LDT + c4);
                }
            }
            PH(11);
            lds_barrier();
            PH(12);
            if (s < CL) { float* tmp = hp; hp = hn; hn = tmp; }
        }
    }
    PH8_FLUSH();
    // the fc2 gradient / statistics columns of this row are summed by the head workgroups (rows 0 .. nh - 1): zero them from nh on
    if ((int)blockIdx.x >= x.nh) {
        float* out = a.partial + (size_t)blockIdx.x * a.PS;
        for (int i = threadIdx.x; i < a.K * H; i += NT8) out[off.W2 + i] = 0.0f;
        if (threadIdx.x < a.K) out[off.b2 + threadIdx.x] = 0.0f;
        if (threadIdx.x < CM_NUM_STATS) out[off.P + threadIdx.x] = 0.0f;
    }
}

// transposed tiles [64 columns][32 rows] with row stride LTT: the operands of the weight-gradient MFMAs (contraction over the tile's
// rows) come out of them as 16-byte reads, four MFMAs per pair of reads, instead of two 4-byte reads per MFMA
constexpr int LTT = 36;
// acc[32 (n) x 32 (k)] += sum over the 32 rows of Z[row][n0 + i] * X[row][k0 + j], both operands from TRANSPOSED tiles
__device__ __forceinline__ void colred32t(f32x16& acc, const float* ZT_n0, const float* XT_k0) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float4* ap = reinterpret_cast<const float4*>(ZT_n0 + r * LTT + 4 * h);
    const float4* bp = reinterpret_cast<const float4*>(XT_k0 + r * LTT + 4 * h);
#pragma unroll
    for (int j = 0; j < T32 / 8; ++j) {
        const float4 a = ap[2 * j], b = bp[2 * j];
        acc = mfma32(a.x, b.x, acc);
        acc = mfma32(a.y, b.y, acc);
        acc = mfma32(a.z, b.z, acc);
        acc = mfma32(a.w, b.w, acc);
    }
}
// acc[32 rows x 32 cols] += dG[32 rows][64 (n)] * (register image of W[64 n][32 cols]), dG read from its TRANSPOSED tile [n][row]:
// the A operand of MFMA (j, q) is dG[row r][n = 8j + 4h + q] = GT[(8j + 4h + q) * LTT + r] -- 4-byte reads, lanes r contiguous
__device__ __forceinline__ void rowpar_rb_t(f32x16& acc, const float* GT, const float (&w)[32]) {
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const float* ap = GT + (4 * h) * LTT + r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float a0 = ap[(8 * j) * LTT], a1 = ap[(8 * j + 1) * LTT], a2 = ap[(8 * j + 2) * LTT], a3 = ap[(8 * j + 3) * LTT];
        acc = mfma32(a0, w[4 * j], acc);
        acc = mfma32(a1, w[4 * j + 1], acc);
        acc = mfma32(a2, w[4 * j + 2], acc);
        acc = mfma32(a3, w[4 * j + 3], acc);
    }
}
constexpr int G2B_TILES = 8;  // per set: G0T G1T G2T G3T X1T HPT OBT D1T
inline size_t gru2_bwd_lds_bytes() { return (size_t)(5 * T32 * LDT + G2B_TILES * HP * LTT + 2 * NTHREADS) * sizeof(float); }

// Weight-gradient workgroups of the pipelined backward sweep (k_gru2_bwd<true>): on the CUs the tiles leave idle.  The chain workgroups
// publish dh_t (the carried gradient plus the head's share) of every step but the last one processed as self-certifying 64-bit words
// {value, launch tag} (agent-scope stores, no flag, no wait on the chain); a unit of work here is one (tile, step):
// reload the step's activations (written by the forward launch), redo the element-wise gate derivatives from dh_t, lay the operands
// out as transposed tiles and run five of the seven weight-gradient products (W_ih x 3, W_hr, W_hz) -- the chain keeps dW_1 and dW_hn,
// which fit under its own element-wise phase.  Units are dealt round-robin in publication order (u = step-from-the-end * tiles + tile,
// workgroup u mod ng), so every workgroup lags the chain by the same amount whatever the tile count.

__device__ __forceinline__ void gru2_grad_wg(const GruArgs& a, const GruXArgs& x, float* smem) {
    const GruOff off = gru_offsets(a.din, a.H, a.K);
    float* S = smem;   // 5 transposed tiles [64][LTT]
    float *G0T = S, *G1T = S + HP * LTT, *G2T = S + 2 * HP * LTT, *X1T = S + 3 * HP * LTT, *HPT = S + 4 * HP * LTT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, role = wave >> 1, h = lane >> 5, lc = lane & 31;
    const int H = a.H, CL = a.t1 - a.t0;
    const long R = (long)a.E * a.A;
    const int col = 32 * wn + lc;
    const int ec = tid & 63, er0 = 8 * (tid >> 6);
    f32x16 accWih[3], accWhh[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#pragma unroll
        for (int q = 0; q < 3; ++q) accWih[q][i] = 0.f;
        accWhh[0][i] = 0.f; accWhh[1][i] = 0.f;
    }
    PH_DECL
#ifdef CM_PHASE_PROF
    const unsigned long long ph_start = __builtin_amdgcn_s_memrealtime();   // 100 MHz, one clock for the whole device
#endif
    const int g = (int)blockIdx.x - x.nt, ng = (int)gridDim.x - x.nt;
    const long U = (long)x.nt * (CL - 1);
    bool dead = false;
    struct Pre { float x1[8], rr[8], zz[8], nn[8], ghn[8], hprev[8]; } P;
    auto load_stable = [&](long u) {   // everything of the unit that the forward launch wrote
        const int q = (int)(u % x.nt), s = CL - 1 - (int)(u / x.nt);
        const long row0 = (long)q * T32;
        const float* wsS = a.ws_act + ((long)s * R + row0) * WS2;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = er0 + e;
            const float* w = wsS + (long)r * WS2;
            P.x1[e] = P.rr[e] = P.zz[e] = P.nn[e] = P.ghn[e] = P.hprev[e] = 0.0f;
            if (row0 + r < R && ec < H) {
                P.x1[e] = w[ec]; P.rr[e] = w[HP + ec]; P.zz[e] = w[2 * HP + ec]; P.nn[e] = w[3 * HP + ec]; P.ghn[e] = w[4 * HP + ec];
                P.hprev[e] = a.ws_act[((long)(s - 1) * R + row0 + r) * WS2 + 5 * HP + ec];   // s >= 1 here
            }
        }
    };
    // dh_t of a unit: 64-bit words {value, launch tag} that the chain workgroup stores with agent scope.  A word certifies itself, so
    // the loads can be issued long before the unit is due (under the products of the unit before) and simply repeated if a tag is still
    // missing -- one round trip to memory (~2 us under this load) per unit, hidden, where a flag followed by the data costs two in a row
    unsigned long long dq[8];
    auto load_dh = [&](long u) {
        const int q = (int)(u % x.nt), s = CL - 1 - (int)(u / x.nt);
        const long row0 = (long)q * T32;
        const unsigned long long* wd = x.dhq + ((long)s * R + row0) * HP + ec;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            dq[e] = (unsigned long long)x.tag << 32;   // rows / columns outside the tile: valid, zero
            if (row0 + er0 + e < R && ec < H) dq[e] = __hip_atomic_load(wd + (long)(er0 + e) * HP, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto have_dh = [&]() -> bool {
        bool ok = true;
#pragma unroll
        for (int e = 0; e < 8; ++e) ok = ok && (unsigned)(dq[e] >> 32) == x.tag;
        return __builtin_amdgcn_ballot_w64(!ok) == 0ull;
    };
    if (g < U) { load_stable(g); load_dh(g); }
    for (long u = g; u < U; u += ng) {
        const long un = u + ng;
        const bool has_next = un < U;
        if (!dead) {
            for (int it = 0; !have_dh(); ++it) {
                if (it >= GX_POLLS) { dead = true; break; }
                __builtin_amdgcn_s_sleep(2);
                load_dh(u);
            }
        }
        PH(0);
        float v0[8], v1[8], v2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {   // the element-wise arithmetic of the chain (k_gru2_bwd), same order
            const float rr = P.rr[e], zz = P.zz[e], nn = P.nn[e], ghn = P.ghn[e], hprev = P.hprev[e];
            const float dh = dead ? 0.0f : __uint_as_float((unsigned)dq[e]);
            const float dn = dh * (1.0f - zz), dzg = dh * (hprev - nn);
            const float dn_pre = dn * (1.0f - nn * nn);
            v0[e] = dn_pre * ghn * rr * (1.0f - rr); v1[e] = dzg * zz * (1.0f - zz); v2[e] = dn_pre;
        }
        lds_barrier();  // the products of the previous unit have read the tiles
        {
            const int ot = ec * LTT + er0;
#define CM_ST8(dst, v) do { *reinterpret_cast<float4*>(dst + ot) = make_float4(v[0], v[1], v[2], v[3]); \
                            *reinterpret_cast<float4*>(dst + ot + 4) = make_float4(v[4], v[5], v[6], v[7]); } while (0)
            CM_ST8(G0T, v0); CM_ST8(G1T, v1); CM_ST8(G2T, v2); CM_ST8(X1T, P.x1); CM_ST8(HPT, P.hprev);
#undef CM_ST8
        }
        if (has_next) { load_dh(un); load_stable(un); }   // under the products
        PH(1);
        lds_barrier();
        PH(2);
        colred32t(accWih[0], G0T + 32 * role * LTT, X1T + 32 * wn * LTT);
        colred32t(accWih[1], G1T + 32 * role * LTT, X1T + 32 * wn * LTT);
        colred32t(accWih[2], G2T + 32 * role * LTT, X1T + 32 * wn * LTT);
        colred32t(accWhh[0], G0T + 32 * role * LTT, HPT + 32 * wn * LTT);
        colred32t(accWhh[1], G1T + 32 * role * LTT, HPT + 32 * wn * LTT);
        PH(3);
    }
#ifdef CM_PHASE_PROF
    ph_[7] = __builtin_amdgcn_s_memrealtime(); ph_[6] = ph_start;
    if (a.prof && threadIdx.x == 0) { for (int i_ = 0; i_ < 8; ++i_) a.prof[(size_t)(512 + blockIdx.x) * 16 + i_] = ph_[i_]; }
#endif
    // ================================ this workgroup's partial row (behind the rows of the chain workgroups): its five blocks, zero elsewhere
    float* out = a.partial + (size_t)blockIdx.x * a.PS;
    const float poison = dead ? __builtin_nanf("") : 0.0f;
    for (int i = tid; i < a.PS; i += NTHREADS) {
        const bool mine = (i >= off.Wih && i < off.Whh + 2 * H * H);   // W_ih (3 blocks), W_hr, W_hz
        if (!mine) out[i] = (i >= off.P && i < off.P + CM_NUM_STATS) ? poison : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n = 32 * role + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (n < H && col < H) {
#pragma unroll
            for (int q = 0; q < 3; ++q) out[off.Wih + (q * H + n) * H + col] = accWih[q][i];
            out[off.Whh + n * H + col] = accWhh[0][i];
            out[off.Whh + (H + n) * H + col] = accWhh[1][i];
        }
    }
}

// Backward sweep.  The recurrence of a step is: gate derivatives (element-wise, needs dh) -> data path (dx1 / dh_prev = dG W, 96 MFMAs
// per wave) -> dh.  The weight gradients of the step (112 MFMAs per wave) are NOT on that chain: they are issued one step LATER, in the
// barrier interval in which the NEXT step's gate derivatives are computed -- into registers, so that the tiles the MFMAs read stay
// intact -- and the matrix pipe works through them while the VALU produces the derivatives; the tiles of the new step are written
// after the next barrier.  Three barriers per step (twelve in the first generation).  Weight-gradient operands come from transposed
// [column][row] tiles (16-byte reads along the contracted rows), the data path reads row-major tiles and register weights.
// XG (pipelined): five of the seven weight-gradient products of every step but the last one processed run on other CUs (gru2_grad_wg
// above: workgroups blockIdx >= x.nt); the chain workgroups publish dh_t for them and keep dW_1 and dW_hn.  x is unused without XG.
template <bool XG>
__global__ __launch_bounds__(NTHREADS, 1) void k_gru2_bwd(const GruArgs a, const GruXArgs x) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (XG && (int)blockIdx.x >= x.nt) { gru2_grad_wg(a, x, smem); return; }
    const GruOff off = gru_offsets(a.din, a.H, a.K);
    float* DH = smem;                                   // row-major [32][LDT]: dh carried backwards
    float* G0 = DH + T32 * LDT;                         // row-major pre-activation gradients of the step (data path)
    float* G1 = G0 + T32 * LDT;
    float* G2 = G1 + T32 * LDT;
    float* G3 = G2 + T32 * LDT;
    float* S = G3 + T32 * LDT;                          // 8 transposed tiles [64][LTT]: G0T G1T G2T G3T X1T HPT OBT D1T
    float *G0T = S, *G1T = S + HP * LTT, *G2T = S + 2 * HP * LTT, *G3T = S + 3 * HP * LTT;
    float *X1T = S + 4 * HP * LTT, *HPT = S + 5 * HP * LTT, *OBT = S + 6 * HP * LTT, *D1T = S + 7 * HP * LTT;
    float* red = S + G2B_TILES * HP * LTT;              // 2 * NTHREADS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave & 1, role = wave >> 1, h = lane >> 5, lc = lane & 31;
    const int H = a.H, din = a.din, T = a.T, CL = a.t1 - a.t0;
    const long R = (long)a.E * a.A;
    const int col = 32 * wn + lc;
    const int ec = tid & 63, er0 = 8 * (tid >> 6);  // element-wise phase: this thread owns column ec of rows er0 .. er0 + 7
    // role 0: dx1 = sum_q dG_i[q] W_ih[q]; role 1: dh_prev = sum_q dG_h[q] W_hh[q] -- this wave's 32 columns of the three blocks
    float wG[3][32];
#pragma unroll
    for (int q = 0; q < 3; ++q) load_tn_regs(wG[q], a.params + (role == 0 ? off.Wih : off.Whh) + q * H * H, 32 * wn, H, H, H);

    f32x16 accW1, accWih[3], accWhh[3];
    float db1 = 0.f, dbg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        accW1[i] = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) { accWih[q][i] = 0.f; accWhh[q][i] = 0.f; }
    }
    PH_DECL
#ifdef CM_PHASE_PROF
    const unsigned long long ph_start = __builtin_amdgcn_s_memrealtime();   // 100 MHz, one clock for the whole device
#endif
    const long ntiles = (R + T32 - 1) / T32;
    struct Pre { float x1[8], rr[8], zz[8], nn[8], ghn[8], hprev[8], dhh[8], ob[8]; } P;
    // weight gradients of the step whose tiles are in LDS
    auto wgrad = [&]() {
        colred32t(accWih[0], G0T + 32 * role * LTT, X1T + 32 * wn * LTT);
        colred32t(accWih[1], G1T + 32 * role * LTT, X1T + 32 * wn * LTT);
        colred32t(accWih[2], G2T + 32 * role * LTT, X1T + 32 * wn * LTT);
        colred32t(accWhh[0], G0T + 32 * role * LTT, HPT + 32 * wn * LTT);
        colred32t(accWhh[1], G1T + 32 * role * LTT, HPT + 32 * wn * LTT);
        colred32t(accWhh[2], G3T + 32 * role * LTT, HPT + 32 * wn * LTT);
        colred32t(accW1, D1T + 32 * role * LTT, OBT + 32 * wn * LTT);
        const float4 u = *reinterpret_cast<const float4*>(D1T + ec * LTT + er0), w4 = *reinterpret_cast<const float4*>(D1T + ec * LTT + er0 + 4);
        db1 += ((u.x + u.y) + (u.z + u.w)) + ((w4.x + w4.y) + (w4.z + w4.w));
    };
    auto wgrad2 = [&]() {   // XG: what the chain keeps of a published step
        colred32t(accWhh[2], G3T + 32 * role * LTT, HPT + 32 * wn * LTT);
        colred32t(accW1, D1T + 32 * role * LTT, OBT + 32 * wn * LTT);
        const float4 u = *reinterpret_cast<const float4*>(D1T + ec * LTT + er0), w4 = *reinterpret_cast<const float4*>(D1T + ec * LTT + er0 + 4);
        db1 += ((u.x + u.y) + (u.z + u.w)) + ((w4.x + w4.y) + (w4.z + w4.w));
    };
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const long row0 = tile * T32;
        // everything a step reads from HBM is requested one step AHEAD, under the MFMA phases of the step before
        auto load_pre = [&](int s) {
            const float* wsS = a.ws_act + ((long)s * R + row0) * WS2;
            const float* ob = a.obs + (long)(a.t0 + s) * din;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int r = er0 + e;
                const bool rok = row0 + r < R;
                const float* w = wsS + (long)r * WS2;
                P.x1[e] = P.rr[e] = P.zz[e] = P.nn[e] = P.ghn[e] = P.hprev[e] = P.dhh[e] = P.ob[e] = 0.0f;
                if (rok && ec < H) {
                    P.x1[e] = w[ec]; P.rr[e] = w[HP + ec]; P.zz[e] = w[2 * HP + ec]; P.nn[e] = w[3 * HP + ec]; P.ghn[e] = w[4 * HP + ec];
                    P.dhh[e] = w[6 * HP + ec];
                    if (s > 0) P.hprev[e] = a.ws_act[((long)(s - 1) * R + row0 + r) * WS2 + 5 * HP + ec];
                    else if (a.h_in) P.hprev[e] = a.h_in[(row0 + r) * H + ec];
                }
                if (rok && ec < din) P.ob[e] = ob[(row0 + r) * (long)T * din + ec];
            }
        };
        __syncthreads();
        for (int i = tid; i < T32 * LDT; i += NTHREADS) DH[i] = 0.0f;
        load_pre(CL - 1);
        for (int s = CL - 1; s >= 0; --s) {
            lds_barrier();  // dh_prev of the step before is in DH, its dx1 in D1T: the tiles in LDS are those of step s + 1, complete
            PH(0);
            // ---- gate derivatives of step s INTO REGISTERS (dh = carried dh + the head's share, k_gru2_fwd) ...
            float v0[8], v1[8], v2[8], v3[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int o = (er0 + e) * LDT + ec;
                const float rr = P.rr[e], zz = P.zz[e], nn = P.nn[e], ghn = P.ghn[e], hprev = P.hprev[e];
                const float dh = DH[o] + P.dhh[e];
                const float dn = dh * (1.0f - zz), dzg = dh * (hprev - nn);
                const float dn_pre = dn * (1.0f - nn * nn);
                v0[e] = dn_pre * ghn * rr * (1.0f - rr); v1[e] = dzg * zz * (1.0f - zz); v2[e] = dn_pre; v3[e] = dn_pre * rr;
                DH[o] = dh * zz;
                if (XG && s > 0 && row0 + er0 + e < R && ec < H)   // {dh_t, launch tag} for the weight-gradient workgroups: fire and forget
                    __hip_atomic_store(x.dhq + ((long)s * R + row0 + er0 + e) * HP + ec, ((unsigned long long)x.tag << 32) | __float_as_uint(dh),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // ---- ... while the matrix pipe works through the weight gradients of step s + 1 (independent of the lines above)
            if (s + 1 < CL) { if (XG) wgrad2(); else wgrad(); }
            lds_barrier();  // every read of the tiles of step s + 1 is done
            PH(1);
            {   // tiles of step s: row-major for the data path, transposed for the weight gradients
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int o = (er0 + e) * LDT + ec;
                    G0[o] = v0[e]; G1[o] = v1[e]; G2[o] = v2[e]; G3[o] = v3[e];
                }
                const int ot = ec * LTT + er0;
#define CM_ST8(dst, v) do { *reinterpret_cast<float4*>(dst + ot) = make_float4(v[0], v[1], v[2], v[3]); \
                            *reinterpret_cast<float4*>(dst + ot + 4) = make_float4(v[4], v[5], v[6], v[7]); } while (0)
                if (!XG || s == 0) { CM_ST8(G0T, v0); CM_ST8(G1T, v1); CM_ST8(G2T, v2); }  // published steps: their products run elsewhere
                CM_ST8(G3T, v3); CM_ST8(X1T, P.x1); CM_ST8(HPT, P.hprev); CM_ST8(OBT, P.ob);
#undef CM_ST8
                dbg[0] += ((v0[0] + v0[1]) + (v0[2] + v0[3])) + ((v0[4] + v0[5]) + (v0[6] + v0[7]));
                dbg[1] += ((v1[0] + v1[1]) + (v1[2] + v1[3])) + ((v1[4] + v1[5]) + (v1[6] + v1[7]));
                dbg[2] += ((v2[0] + v2[1]) + (v2[2] + v2[3])) + ((v2[4] + v2[5]) + (v2[6] + v2[7]));
                dbg[3] += ((v3[0] + v3[1]) + (v3[2] + v3[3])) + ((v3[4] + v3[5]) + (v3[6] + v3[7]));
            }
            lds_barrier();
            PH(2);
            if (s > 0) load_pre(s - 1);
            // ---- data path of step s, weights from registers
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            rowpar_rb(acc, G0, wG[0], HP / 8);
            rowpar_rb(acc, G1, wG[1], HP / 8);
            rowpar_rb(acc, role == 0 ? G2 : G3, wG[2], HP / 8);
            PH(3);
            if (role == 0) {  // dx1 through relu' -> D1T (nobody reads it before the next barrier)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 x = *reinterpret_cast<const float4*>(X1T + col * LTT + 8 * q + 4 * h);
                    *reinterpret_cast<float4*>(D1T + col * LTT + 8 * q + 4 * h) =
                        make_float4(x.x > 0.f ? acc[4 * q] : 0.f, x.y > 0.f ? acc[4 * q + 1] : 0.f,
                                    x.z > 0.f ? acc[4 * q + 2] : 0.f, x.w > 0.f ? acc[4 * q + 3] : 0.f);
                }
            } else {          // dh_{t-1}: the element-wise phase is two barriers back
#pragma unroll
                for (int i = 0; i < 16; ++i) DH[((i & 3) + 8 * (i >> 2) + 4 * h) * LDT + col] += acc[i];
            }
            PH(4);
        }
        lds_barrier();
        wgrad();  // step 0
    }
#ifdef CM_PHASE_PROF
    ph_[7] = __builtin_amdgcn_s_memrealtime(); ph_[6] = ph_start;
#endif
    PH2_FLUSH(512);
    // ================================ partial gradient of this workgroup (fc2 + statistics were written by k_gru2_fwd)
    float* out = a.partial + (size_t)blockIdx.x * a.PS;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int n = 32 * role + (i & 3) + 8 * (i >> 2) + 4 * h;
        if (n < H && col < din) out[off.W1 + n * din + col] = accW1[i];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (n < H && col < H) {
                out[off.Wih + (q * H + n) * H + col] = accWih[q][i];
                out[off.Whh + (q * H + n) * H + col] = accWhh[q][i];
            }
        }
    }
    {   // column-sum biases: 4 row parts per column (thread = column ec, rows er0 ..)
        float vals[5] = {db1, dbg[0], dbg[1], dbg[2], dbg[3]};
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            __syncthreads();
            red[(tid >> 6) * HP + (tid & 63)] = vals[q];
            __syncthreads();
            if (tid < H) {
                const float sv = red[tid] + red[HP + tid] + red[2 * HP + tid] + red[3 * HP + tid];
                if (q == 0) out[off.b1 + tid] = sv;
                else if (q == 1) { out[off.bih + tid] = sv; out[off.bhh + tid] = sv; }
                else if (q == 2) { out[off.bih + H + tid] = sv; out[off.bhh + H + tid] = sv; }
                else if (q == 3) out[off.bih + 2 * H + tid] = sv;
                else out[off.bhh + 2 * H + tid] = sv;
            }
        }
    }
}
