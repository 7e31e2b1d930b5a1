// Do two waves of ONE SIMD overlap?  (round 6: interval times of the split GRU sweep were the SUM of the recurrence waves' gate math and the
// helper waves' MFMA products, not the maximum -- docs/HISTORY.md.)  One workgroup of 512 threads = 8 waves = 2 per SIMD on one CU; waves 0-3
// run workload A, waves 4-7 workload B, each for a fixed instruction count; timed by wave 0 / wave 4 with s_memtime:
//   A alone (B waves exit at once), B alone, A and B together.  Perfect overlap: together = max(A, B); none: together = A + B.
// Workloads: M16 = dependent-pair chains of v_mfma_f32_16x16x4_f32 (2 accumulators: the GRU products' shape), M32 = v_mfma_f32_32x32x2_f32
// (k_mlp's), FMA = independent v_fma_f32 chains, TRN = v_exp_f32 / v_rcp_f32 chains (the gate math's transcendentals), LDS = ds_read_b128 stream.
//     hipcc -O3 --offload-arch=gfx950 cosimd_overlap.hip -o cosimd_overlap && ./cosimd_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
enum { M16 = 0, M32 = 1, FMA = 2, TRN = 3, LDS = 4, NONE = 5 };

template <int W>
__device__ __forceinline__ float work(int iters, float seed, const float* lds) {
    float r = 0.f;
    if (W == M16) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 1.0f, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(seed, 2.0f, a1, 0, 0, 0); }
        }
        r = a0[0] + a1[1];
    } else if (W == M32) {
        f32x16 a0;
        for (int g = 0; g < 16; ++g) a0[g] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(seed, 1.0f, a0, 0, 0, 0);
        }
        r = a0[0] + a0[7];
    } else if (W == FMA) {
        float x[8];
        for (int j = 0; j < 8; ++j) x[j] = seed + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], 0.999f, 0.001f);
        }
        for (int j = 0; j < 8; ++j) r += x[j];
    } else if (W == TRN) {
        float x[4];
        for (int j = 0; j < 4; ++j) x[j] = seed * 0.01f + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) x[j] = __builtin_amdgcn_rcpf(1.0f + __expf(-x[j]));
        }
        for (int j = 0; j < 4; ++j) r += x[j];
    } else if (W == LDS) {
        f32x4 s = {0, 0, 0, 0};
        const f32x4* p = reinterpret_cast<const f32x4*>(lds) + (threadIdx.x & 63);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) s += p[64 * ((i + u) & 7)];
        }
        r = s[0] + s[3];
    }
    return r;
}

template <int A, int B>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int ia, int ib) {
    __shared__ __attribute__((aligned(16))) float lds[8 * 64 * 4];
    for (int i = threadIdx.x; i < 8 * 64 * 4; i += 512) lds[i] = 1e-3f * i;
    __syncthreads();
    const bool second = threadIdx.x >= 256;
    const float seed = 1.0f + 1e-3f * (threadIdx.x & 63);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r = second ? work<B>(ib, seed, lds) : work<A>(ia, seed, lds);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if ((threadIdx.x & 255) == 0) cyc[2 * blockIdx.x + (second ? 1 : 0)] = t1 - t0;
}

template <int A, int B> void run3(const char* na, const char* nb, int ia, int ib) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 2 * 8);
    unsigned long long h[2];
    double t[3][2];
    for (int mode = 0; mode < 3; ++mode) {  // 0: A alone, 1: B alone, 2: together
        for (int rep = 0; rep < 3; ++rep) {
            if (mode == 0) hipLaunchKernelGGL((k<A, NONE>), dim3(1), dim3(512), 0, 0, out, cyc, ia, ib);
            else if (mode == 1) hipLaunchKernelGGL((k<NONE, B>), dim3(1), dim3(512), 0, 0, out, cyc, ia, ib);
            else hipLaunchKernelGGL((k<A, B>), dim3(1), dim3(512), 0, 0, out, cyc, ia, ib);
        }
        hipDeviceSynchronize();
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        t[mode][0] = (double)h[0]; t[mode][1] = (double)h[1];
    }
    const double a = t[0][0], b = t[1][1], ta = t[2][0], tb = t[2][1];
    const double both = ta > tb ? ta : tb, mx = a > b ? a : b;
    printf("A = %-4s %8.0f cycles alone | B = %-4s %8.0f alone | together: A %8.0f B %8.0f | max(A,B) %8.0f  A+B %8.0f  -> overlap %.2f (1 = perfect, 0 = serial)\n",
           na, a, nb, b, ta, tb, mx, a + b, (a + b - both) / (a + b - mx > 1 ? a + b - mx : 1));
    hipFree(out); hipFree(cyc);
}

// Within ONE wave: how many independent VALU instructions fit between two MFMAs without lengthening the MFMA stream?
template <int KIND, int NV>
__global__ void k_shadow(float* out, unsigned long long* cyc, int iters) {
    f32x16 a32; f32x4 a16 = {0, 0, 0, 0}, b16 = {0, 0, 0, 0};
    for (int g = 0; g < 16; ++g) a32[g] = 0.f;
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = 1.0f + j + threadIdx.x * 1e-3f;
    const float sd = 1.0f + 1e-3f * threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 32) a32 = __builtin_amdgcn_mfma_f32_32x32x2f32(sd, 1.0f, a32, 0, 0, 0);
            else { a16 = __builtin_amdgcn_mfma_f32_16x16x4f32(sd, 1.0f, a16, 0, 0, 0); b16 = __builtin_amdgcn_mfma_f32_16x16x4f32(sd, 2.0f, b16, 0, 0, 0); }
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j & 15]) : "v"(0.999f), "v"(0.001f));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = a32[0] + a16[0] + b16[1];
    for (int j = 0; j < 16; ++j) r += x[j];
    out[threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// ... and LDS reads (ds_read_b128, results summed afterwards) in the same place
template <int NL>
__global__ void k_shadow_lds(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4 * 16];
    for (int i = threadIdx.x; i < 64 * 4 * 16; i += 64) lds[i] = 1e-3f * i;
    __syncthreads();
    f32x16 a32;
    for (int g = 0; g < 16; ++g) a32[g] = 0.f;
    f32x4 v[NL > 0 ? NL : 1];
    for (int j = 0; j < (NL > 0 ? NL : 1); ++j) v[j] = f32x4{0, 0, 0, 0};
    const unsigned addr = (unsigned)reinterpret_cast<uintptr_t>(lds) + 16 * threadIdx.x;
    const float sd = 1.0f + 1e-3f * threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a32 = __builtin_amdgcn_mfma_f32_32x32x2f32(sd, 1.0f, a32, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NL; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(addr), "n"(1024 * (j & 15)));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = a32[0];
    for (int j = 0; j < (NL > 0 ? NL : 1); ++j) r += v[j][0];
    out[threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NL> void run_shadow_lds() {
    float* out; unsigned long long* cyc; unsigned long long h;
    hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_shadow_lds<NL>), dim3(1), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("one wave, 1 x v_mfma_f32_32x32x2 + %2d ds_read_b128 behind it: %.1f cycles per group\n", NL, (double)h / (iters * 8.0));
    hipFree(out); hipFree(cyc);
}
template <int KIND, int NV> void run_shadow() {
    float* out; unsigned long long* cyc; unsigned long long h;
    hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_shadow<KIND, NV>), dim3(1), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double per = (double)h / (iters * 8.0);
    printf("one wave, %s + %2d independent v_fma_f32 behind it: %.1f cycles per group\n", KIND == 32 ? "1 x v_mfma_f32_32x32x2 (64-cycle pipe time)" : "2 x v_mfma_f32_16x16x4 (2 x 32)", NV, per);
    hipFree(out); hipFree(cyc);
}

// the same question for a bf16 MFMA (v_mfma_f32_32x32x16_bf16: 8 passes, 16 x the fp32 rate per pass): does the real matrix core run beside the VALU?
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NV>
__global__ void k_shadow_bf16(float* out, unsigned long long* cyc, int iters) {
    f32x16 a32;
    for (int g = 0; g < 16; ++g) a32[g] = 0.f;
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(1.0f + 0.01f * j); b[j] = (__bf16)(0.5f + 0.01f * threadIdx.x); }
    float x[16];
    for (int j = 0; j < 16; ++j) x[j] = 1.0f + j + threadIdx.x * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            a32 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, a32, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j & 15]) : "v"(0.999f), "v"(0.001f));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = a32[0];
    for (int j = 0; j < 16; ++j) r += x[j];
    out[threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NV> void run_shadow_bf16() {
    float* out; unsigned long long* cyc; unsigned long long h;
    hipMalloc(&out, 64 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_shadow_bf16<NV>), dim3(1), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("one wave, 1 x v_mfma_f32_32x32x16_bf16 + %2d independent v_fma_f32 behind it: %.1f cycles per group\n", NV, (double)h / (iters * 8.0));
    hipFree(out); hipFree(cyc);
}

// two waves of one SIMD again, the YOUNGER wave (B: VALU) at s_setprio 3: does priority let it into the MFMA wave's stream?
template <int A>
__global__ __launch_bounds__(512) void k_prio(float* out, unsigned long long* cyc, int ia, int ib) {
    const bool second = threadIdx.x >= 256;
    const float seed = 1.0f + 1e-3f * (threadIdx.x & 63);
    if (second) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r = second ? work<FMA>(ib, seed, nullptr) : work<A>(ia, seed, nullptr);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = r;
    if ((threadIdx.x & 255) == 0) cyc[second ? 1 : 0] = t1 - t0;
}
template <int A> void run_prio(const char* na, int ia, int ib) {
    float* out; unsigned long long* cyc; unsigned long long h[2];
    hipMalloc(&out, 512 * 4); hipMalloc(&cyc, 16);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_prio<A>), dim3(1), dim3(512), 0, 0, out, cyc, ia, ib);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
    printf("A = %s beside B = FMA at s_setprio 3: A %llu cycles, B %llu cycles\n", na, h[0], h[1]);
    hipFree(out); hipFree(cyc);
}

int main() {
    run_shadow<32, 0>(); run_shadow<32, 4>(); run_shadow<32, 8>(); run_shadow<32, 12>(); run_shadow<32, 16>(); run_shadow<32, 24>();
    run_shadow<16, 0>(); run_shadow<16, 4>(); run_shadow<16, 8>(); run_shadow<16, 12>(); run_shadow<16, 16>();
    run_shadow_bf16<0>(); run_shadow_bf16<4>(); run_shadow_bf16<8>(); run_shadow_bf16<16>();
    run_shadow_lds<0>(); run_shadow_lds<1>(); run_shadow_lds<2>(); run_shadow_lds<4>(); run_shadow_lds<8>();
    run_prio<M16>("M16", 400, 200); run_prio<M32>("M32", 400, 400);
    // iteration counts chosen so that A and B take about the same time alone (the informative case)
    run3<M16, M16>("M16", "M16", 400, 400);
    run3<M32, M32>("M32", "M32", 400, 400);
    run3<M16, FMA>("M16", "FMA", 400, 200);
    run3<M32, FMA>("M32", "FMA", 400, 400);
    run3<M16, TRN>("M16", "TRN", 400, 100);
    run3<M32, TRN>("M32", "TRN", 400, 200);
    run3<M16, LDS>("M16", "LDS", 400, 400);
    run3<M32, LDS>("M32", "LDS", 400, 800);
    run3<FMA, FMA>("FMA", "FMA", 200, 200);
    run3<FMA, TRN>("FMA", "TRN", 200, 100);
    run3<FMA, LDS>("FMA", "LDS", 200, 400);
    return 0;
}
