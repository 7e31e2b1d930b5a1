// wave_private_probe.hip -- feasibility probe for docs/KERNEL_NOTES.md §10 item 0 (NOT part of the product; numerics unchecked).
//
// Question: how fast does the MFMA / LDS / HBM skeleton of the actor fwd+bwd pass run when every wave owns a 16-row tile end
// to end (no workgroup barrier in the tile loop), compared with the 2 x 2 wave split of k_mlp (1.87 ms at config 3, matrix
// pipe busy 65 %)?  Same work per row as k_mlp<1, M_ACTOR> at Din = 64 (padded), H = 64, one hidden->hidden layer, 16 padded
// head outputs: 368 v_mfma_f32_16x16x4_f32 per 16 rows, three private LDS row buffers per wave, 144 weight-gradient
// accumulator registers per lane.  One workgroup of 8 waves per CU (weights shared once in LDS: 154 KB).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/wpp tools/probes/wave_private_probe.hip && /tmp/wpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int H = 64, LD = 68, DL = 20, NW = 8, NT = NW * 64;
constexpr int WBUF = 16 * LD;                       // one private row buffer (floats)
constexpr int PRIV = 3 * WBUF + 16 * DL;            // X | H0 | H1 | dlogits
constexpr int LDS_FLOATS = 2 * H * LD + 16 * LD + NW * PRIV;

#ifdef NO_FENCE
#define WAVE_SYNC() do { __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define WAVE_SYNC_REAL
#endif
#ifdef WAVE_SYNC_REAL
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); } while (0)
#endif

__device__ __forceinline__ void ld16(float (&d)[16], const float* p) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(p + 4 * q);
        d[4 * q] = v.x; d[4 * q + 1] = v.y; d[4 * q + 2] = v.z; d[4 * q + 3] = v.w;
    }
}

__global__ __launch_bounds__(NT, 1) void k_probe(const float* __restrict__ x, long rows, const float* __restrict__ params,
                                                 const int* __restrict__ action, const float* __restrict__ adv, float* __restrict__ out,
                                                 unsigned long long* __restrict__ ticks) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* W0s = smem;
    float* W1s = W0s + H * LD;
    float* Wos = W1s + H * LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    float* Xb = Wos + 16 * LD + wave * PRIV;
    float* H0 = Xb + WBUF;
    float* H1 = H0 + WBUF;
    float* dl = H1 + WBUF;
    for (int i = tid; i < H * H; i += NT) {
        W0s[(i / H) * LD + (i % H)] = params[i];
        W1s[(i / H) * LD + (i % H)] = params[H * H + i];
    }
    for (int i = tid; i < 16 * H; i += NT) Wos[(i / H) * LD + (i % H)] = (i / H) < 5 ? params[2 * H * H + i] : 0.0f;
    __syncthreads();  // the only workgroup barrier
    float b0[4], b1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { b0[j] = params[3 * H * H + 16 * j + r]; b1[j] = params[3 * H * H + H + 16 * j + r]; }

    f32x4 aW0[4][4], aW1[4][4], aWo[4];
    float db0[4], db1[4], st = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        aWo[i] = f32x4{0, 0, 0, 0}; db0[i] = db1[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { aW0[i][j] = f32x4{0, 0, 0, 0}; aW1[i][j] = f32x4{0, 0, 0, 0}; }
    }
    const long ntiles = rows / 16;
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
    const long stride = (long)gridDim.x * NW;
    long tile = (long)blockIdx.x * NW + wave;
    float xa[16];
#ifdef NO_XLOAD
    for (int i = 0; i < 16; ++i) xa[i] = 1e-3f * (lane + i);
#else
    if (tile < ntiles) ld16(xa, x + (tile * 16 + r) * H + 16 * g);
#endif
    for (; tile < ntiles; tile += stride) {
        const long row0 = tile * 16;
        // ---- X: A operand in registers (k = 16 g + s), B-layout copy in the private LDS slice for dW0
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(Xb + r * LD + 16 * g + 4 * q) = make_float4(xa[4 * q], xa[4 * q + 1], xa[4 * q + 2], xa[4 * q + 3]);
#ifdef NO_XLOAD
        const int act = 0; const float av = 0.5f;
#else
        const int act = action[row0 + r];
        const float av = adv[row0 + r];
#endif
        // ---- forward layer 0
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float b[16];
            ld16(b, W0s + (16 * j + r) * LD + 16 * g);
            f32x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 16; ++s) acc = mfma16(xa[s], b[s], acc);
#pragma unroll
            for (int i = 0; i < 4; ++i) H0[(4 * g + i) * LD + 16 * j + r] = fmaxf(acc[i] + b0[j], 0.0f);
        }
        {   // next tile's X in flight under the rest of this tile
#ifndef NO_XLOAD
            const long nt = tile + stride;
            if (nt < ntiles) ld16(xa, x + (nt * 16 + r) * H + 16 * g);
#endif
        }
        WAVE_SYNC();
        // ---- forward layer 1
        {
            float a[16];
            ld16(a, H0 + r * LD + 16 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float b[16];
                ld16(b, W1s + (16 * j + r) * LD + 16 * g);
                f32x4 acc = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 16; ++s) acc = mfma16(a[s], b[s], acc);
#pragma unroll
                for (int i = 0; i < 4; ++i) H1[(4 * g + i) * LD + 16 * j + r] = fmaxf(acc[i] + b1[j], 0.0f);
            }
        }
        WAVE_SYNC();
        // ---- head logits [16 rows][16 outputs] + softmax / surrogate (row m = 4g + i lives in the 16 lanes r of group g)
        {
            float a[16], b[16];
            ld16(a, H1 + r * LD + 16 * g);
            ld16(b, Wos + r * LD + 16 * g);
            f32x4 z = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 16; ++s) z = mfma16(a[s], b[s], z);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#ifdef NO_SOFTMAX
                dl[(4 * g + i) * DL + r] = z[i] * av;
                continue;
#endif
                float zz = r < 5 ? z[i] : -1e9f;
                float m = zz;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
                float e = __expf(zz - m), sum = e;
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
                const float p = e / sum;
                const int arow = __shfl(act, 4 * g + i, 64);
                const float advr = __shfl(av, 4 * g + i, 64);
                const float lp = zz - m - __logf(sum);
                st += (r == arow) ? advr * __expf(lp) : 0.0f;
                dl[(4 * g + i) * DL + r] = r < 5 ? advr * (p - (r == arow ? 1.0f : 0.0f)) : 0.0f;
            }
        }
        WAVE_SYNC();
        // ---- dWout += dl^T H1 (contraction over the 16 rows) ; dZ1 = (dl Wout) .* relu'(H1)
        {
            float a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = dl[(4 * s + g) * DL + r];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) aWo[j] = mfma16(a[s], H1[(4 * s + g) * LD + 16 * j + r], aWo[j]);
            const float4 d4 = *reinterpret_cast<const float4*>(dl + r * DL + 4 * g);
            const float da[4] = {d4.x, d4.y, d4.z, d4.w};
            f32x4 dz[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dz[j] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 4; ++s) dz[j] = mfma16(da[s], Wos[(4 * g + s) * LD + 16 * j + r], dz[j]);
            }
            WAVE_SYNC();  // all lanes have read H1 as the dWout operand before it becomes dZ1
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float* p = H1 + (4 * g + i) * LD + 16 * j + r;
                    const float v = *p > 0.0f ? dz[j][i] : 0.0f;
                    db1[j] += v;
                    *p = v;
                }
        }
        WAVE_SYNC();
        // ---- dW1 += dZ1^T H0 ; dZ0 = (dZ1 W1) .* relu'(H0)
        {
            float a[4][4], b[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) { a[t][s] = H1[(4 * s + g) * LD + 16 * t + r]; b[t][s] = H0[(4 * s + g) * LD + 16 * t + r]; }
#ifndef NO_DW1
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int s = 0; s < 4; ++s) aW1[i][j] = mfma16(a[i][s], b[j][s], aW1[i][j]);
#else
            aW1[0][0][0] += a[0][0] + b[0][0] + a[3][3] + b[3][3];
#endif
            float az[16];
            ld16(az, H1 + r * LD + 16 * g);
            f32x4 dz[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dz[j] = f32x4{0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < 16; ++s) dz[j] = mfma16(az[s], W1s[(16 * g + s) * LD + 16 * j + r], dz[j]);
            }
            WAVE_SYNC();
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float* p = H0 + (4 * g + i) * LD + 16 * j + r;
                    const float v = *p > 0.0f ? dz[j][i] : 0.0f;
                    db0[j] += v;
                    *p = v;
                }
        }
        WAVE_SYNC();
        // ---- dW0 += dZ0^T X
#ifndef NO_DW0
        {
            float a[4][4], b[4][4];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) { a[t][s] = H0[(4 * s + g) * LD + 16 * t + r]; b[t][s] = Xb[(4 * s + g) * LD + 16 * t + r]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int s = 0; s < 4; ++s) aW0[i][j] = mfma16(a[i][s], b[j][s], aW0[i][j]);
        }
#endif
        WAVE_SYNC();
    }
    if (tid == 0) ticks[blockIdx.x] = __builtin_amdgcn_s_memtime() - t_start;
    // per-wave partials (the product kernel would sum the 8 waves through LDS first)
    float* o = out + ((size_t)blockIdx.x * NW + wave) * (64 * 148);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                o[((i * 4 + j) * 4 + q) * 64 + lane] = aW0[i][j][q];
                o[(64 + (i * 4 + j) * 4 + q) * 64 + lane] = aW1[i][j][q];
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) o[(128 + i * 4 + q) * 64 + lane] = aWo[i][q];
        o[(144 + i) * 64 + lane] = db0[i] + db1[i] + st;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    const long rows = 4096L * 8 * 128;
    float *x, *params, *adv, *out; int* action;
    CK(hipMalloc(&x, rows * H * sizeof(float)));
    CK(hipMalloc(&params, (4 * H * H) * sizeof(float)));
    CK(hipMalloc(&adv, rows * sizeof(float)));
    CK(hipMalloc(&action, rows * sizeof(int)));
    const int grid = 256;
    unsigned long long* ticks; CK(hipMalloc(&ticks, grid * sizeof(unsigned long long)));
    CK(hipMalloc(&out, (size_t)grid * NW * 64 * 148 * sizeof(float)));
    std::vector<float> h((size_t)4 * H * H);
    const bool zero = getenv("WPP_ZERO") != nullptr;  // all-zero operands: no bit toggling in the MFMA / LDS / register data paths
    for (size_t i = 0; i < h.size(); ++i) h[i] = zero ? 0.0f : ((float)rand() / RAND_MAX - 0.5f) * 0.25f;
    CK(hipMemcpy(params, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    std::vector<float> hx((size_t)1 << 24);
    for (size_t i = 0; i < hx.size(); ++i) hx[i] = zero ? 0.0f : (float)rand() / RAND_MAX - 0.5f;
    for (size_t off = 0; off < (size_t)rows * H; off += hx.size())
        CK(hipMemcpy(x + off, hx.data(), std::min(hx.size(), (size_t)rows * H - off) * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemcpy(adv, hx.data(), rows * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemset(action, 0, rows * sizeof(int)));
    const size_t lds = (size_t)LDS_FLOATS * sizeof(float);
    CK(hipFuncSetAttribute((const void*)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int warm = getenv("WPP_WARM") ? atoi(getenv("WPP_WARM")) : 3;
    for (int it = 0; it < warm; ++it) hipLaunchKernelGGL(k_probe, dim3(grid), dim3(NT), lds, 0, x, rows, params, action, adv, out, ticks);
    CK(hipDeviceSynchronize());
    const int N = 10;
    CK(hipEventRecord(e0));
    for (int it = 0; it < N; ++it) hipLaunchKernelGGL(k_probe, dim3(grid), dim3(NT), lds, 0, x, rows, params, action, adv, out, ticks);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= N;
    #if defined(NO_DW0) && defined(NO_DW1)
    const double mfma_flop = (double)rows / 16 * 240 * 2048;
#elif defined(NO_DW0)
    const double mfma_flop = (double)rows / 16 * 304 * 2048;
#else
    const double mfma_flop = (double)rows / 16 * 368 * 2048;  // issued (padded) MFMA FLOPs
#endif  // issued (padded) MFMA FLOPs
    static unsigned long long th[256];
    CK(hipMemcpy(th, ticks, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long tk = 0, tmin = ~0ull;
    for (int i = 0; i < grid; ++i) { tk = th[i] > tk ? th[i] : tk; tmin = th[i] < tmin ? th[i] : tmin; }
    printf("shader clock %.0f MHz (longest workgroup window in s_memtime ticks / wall; shortest window %.2f of it)\n", tk / (ms * 1e3), (double)tmin / tk);
    printf("wave-private skeleton: %.3f ms per pass over %ld rows, LDS %zu B; issued-MFMA rate %.1f TFLOP/s = %.1f%% of the 157.3 fp32 peak "
           "(k_mlp<1,M_ACTOR>: 1.87 ms, 103 TFLOP/s issued = 65%%)\n", ms, rows, lds, mfma_flop / ms / 1e9, mfma_flop / ms / 1e9 / 157.3 * 100);
    return 0;
}
