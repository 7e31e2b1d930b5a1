// mfma_split_probe.hip -- measurement behind DESIGN.md's plan for SURVEY 8(f)-4 (error-compensated bf16 MFMA).  NOT part of the
// product library.  One 64x64x64 tile product per workgroup iteration, operands resident in LDS with the product kernels' row
// stride (68 words), each of the 4 waves owning one 32x32 output tile -- exactly the inner loop of k_mlp's rowpar_nt -- in 3 forms:
//   V0  fp32 operands, v_mfma_f32_32x32x2_f32                       (what ships; exact fp32)
//   V1  operands stored as SPLIT WORDS (bf16 hi << 16 | bf16 lo of the residual), unpacked with v_perm_b32,
//       3 x v_mfma_f32_32x32x16_bf16 per K=16 (hi*hi + hi*lo + lo*hi)
//   V2  fp32 operands in LDS, split into hi/lo in registers right before the same 3 MFMAs
// Prints ms per launch, TFLOP/s (algorithmic 2*64*64*64 per tile) and the max relative error of one tile product vs fp64.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/probe tools/probes/mfma_split_probe.hip && /tmp/probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int LDT = 68, TM = 64;

__device__ __forceinline__ unsigned bf16_rne(float x) {  // top 16 bits of x rounded to nearest even
    unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ unsigned split_word(float x) {
    const unsigned hi = bf16_rne(x);
    const float r = x - __uint_as_float(hi << 16);
    return (hi << 16) | bf16_rne(r);
}
union Frag { u32x4 u; bf16x8 b; };

template <int V>
__global__ __launch_bounds__(256, 2) void k_probe(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned As[TM * LDT], Bs[TM * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < TM * 64; i += 256) {
        const int row = i >> 6, c = i & 63;
        const float a = A[i], b = B[i];
        As[row * LDT + c] = (V == 1) ? split_word(a) : __float_as_uint(a);
        Bs[row * LDT + c] = (V == 1) ? split_word(b) : __float_as_uint(b);
    }
    __syncthreads();
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 16; ++g) acc[g] = 0.0f;
    const unsigned* ap = As + (32 * wm + r) * LDT;
    const unsigned* bp = Bs + (32 * wn + r) * LDT;
    for (int it = 0; it < iters; ++it) {
        asm volatile("" ::: "memory");  // operands are re-read from LDS every iteration, like a new tile would be
        if (V == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 a = *reinterpret_cast<const float4*>(ap + 8 * j + 4 * h);
                const float4 b = *reinterpret_cast<const float4*>(bp + 8 * j + 4 * h);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned aw[8], bw[8];
                *reinterpret_cast<u32x4*>(aw) = *reinterpret_cast<const u32x4*>(ap + 16 * j + 8 * h);
                *reinterpret_cast<u32x4*>(aw + 4) = *reinterpret_cast<const u32x4*>(ap + 16 * j + 8 * h + 4);
                *reinterpret_cast<u32x4*>(bw) = *reinterpret_cast<const u32x4*>(bp + 16 * j + 8 * h);
                *reinterpret_cast<u32x4*>(bw + 4) = *reinterpret_cast<const u32x4*>(bp + 16 * j + 8 * h + 4);
                if (V == 2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { aw[e] = split_word(__uint_as_float(aw[e])); bw[e] = split_word(__uint_as_float(bw[e])); }
                }
                Frag ahi, alo, bhi, blo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {  // words 2e, 2e+1 -> one packed pair; element order is the same for A and B
                    ahi.u[e] = __builtin_amdgcn_perm(aw[2 * e + 1], aw[2 * e], 0x07060302u);
                    alo.u[e] = __builtin_amdgcn_perm(aw[2 * e + 1], aw[2 * e], 0x05040100u);
                    bhi.u[e] = __builtin_amdgcn_perm(bw[2 * e + 1], bw[2 * e], 0x07060302u);
                    blo.u[e] = __builtin_amdgcn_perm(bw[2 * e + 1], bw[2 * e], 0x05040100u);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo.b, bhi.b, acc, 0, 0, 0);  // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi.b, blo.b, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi.b, bhi.b, acc, 0, 0, 0);
            }
        }
    }
    float* out = C + ((size_t)blockIdx.x * 4 + wave) * 1024;
#pragma unroll
    for (int g = 0; g < 16; ++g) out[((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + r] = acc[g];
}

template <int V>
static void run(const char* name, const float* dA, const float* dB, float* dC, const std::vector<float>& A, const std::vector<float>& B, int grid) {
    std::vector<float> C(4 * 1024);
    hipLaunchKernelGGL(k_probe<V>, dim3(1), dim3(256), 0, 0, dA, dB, dC, 1);
    hipMemcpy(C.data(), dC, C.size() * sizeof(float), hipMemcpyDeviceToHost);
    double maxrel = 0, maxabs = 0, scale = 0;
    for (int w = 0; w < 4; ++w)
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0, mag = 0;
                const int row = 32 * (w >> 1) + i, col = 32 * (w & 1) + j;
                for (int k = 0; k < 64; ++k) { ref += (double)A[row * 64 + k] * B[col * 64 + k]; mag += fabs((double)A[row * 64 + k] * B[col * 64 + k]); }
                const double e = fabs(C[w * 1024 + i * 32 + j] - ref);
                if (e / mag > maxrel) maxrel = e / mag;   // relative to sum |a_k b_k| (the natural scale of a dot product's round-off)
                if (e > maxabs) maxabs = e;
                scale = fmax(scale, fabs(ref));
            }
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_probe<V>, dim3(grid), dim3(256), 0, 0, dA, dB, dC, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_probe<V>, dim3(grid), dim3(256), 0, 0, dA, dB, dC, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * 64 * 64 * 64 * (double)iters * grid;
    printf("%-44s %8.3f ms  %7.1f TFLOP/s   max|err|/sum|a b| = %.2e   max|err| = %.2e (|C| up to %.1f)\n", name, ms, flop / ms / 1e9, maxrel, maxabs, scale);
}

int main() {
    std::vector<float> A(64 * 64), B(64 * 64);
    srand(7);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2.0f - 1.0f;
    for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.0f - 1.0f) * 0.125f;
    float *dA, *dB, *dC;
    const int grid = 512;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)grid * 4 * 1024 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    run<0>("V0 fp32 MFMA 32x32x2 (ships)", dA, dB, dC, A, B, grid);
    run<1>("V1 split words in LDS + 3 x bf16 32x32x16", dA, dB, dC, A, B, grid);
    run<2>("V2 fp32 in LDS, split in registers + 3 x bf16", dA, dB, dC, A, B, grid);
    return 0;
}
