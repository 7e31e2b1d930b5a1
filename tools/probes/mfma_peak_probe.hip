// mfma_peak_probe.hip -- what fp32 MFMA rate and shader clock does one MI355X sustain (NOT part of the product)?
//   R  operands in registers: nothing but dependent v_mfma_f32_32x32x2_f32 chains (4 accumulators per wave)
//   L  both operands re-read from LDS for every MFMA batch (ds_read_b128, the access pattern of cm_mlp_kernel.h's rowpar_nt)
// for 1, 2 and 4 waves per SIMD.  s_memtime counts shader clocks, so ticks / wall time is the clock the run sustained.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/peak tools/probes/mfma_peak_probe.hip && /tmp/peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool LDS>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* ticks, int iters, int pad_floats) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < 64 * 68; i += 256) sm[i] = 1e-3f * (i % 97);
    __syncthreads();
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int g = 0; g < 16; ++g) acc[q][g] = 0.f;
    float4 a = *reinterpret_cast<const float4*>(sm + r * 68 + 4 * h), b = *reinterpret_cast<const float4*>(sm + (32 + r) * 68 + 4 * h);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (LDS) {
                a = *reinterpret_cast<const float4*>(sm + r * 68 + 8 * j + 4 * h);
                b = *reinterpret_cast<const float4*>(sm + (32 + r) * 68 + 8 * j + 4 * h);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[q], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int g = 0; g < 16; ++g) s += acc[q][g];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s + (float)pad_floats;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <bool LDS>
static void run(const char* name, int wg_per_cu, float* out, unsigned long long* ticks) {
    // LDS padding so that exactly wg_per_cu workgroups (4 waves each = wg_per_cu waves per SIMD) fit a CU
    const size_t lds = wg_per_cu == 1 ? 100 * 1024 : (wg_per_cu == 2 ? 70 * 1024 : 36 * 1024);
    hipFuncSetAttribute((const void*)k<LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = 256 * wg_per_cu, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<LDS>, dim3(grid), dim3(256), lds, 0, out, ticks, iters, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<LDS>, dim3(grid), dim3(256), lds, 0, out, ticks, iters, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t;
    hipMemcpy(&t, ticks, sizeof(t), hipMemcpyDeviceToHost);
    const double flop = (double)grid * 4 * iters * 8 * 16 * 4096.0;
    printf("%s  %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s = %5.1f %% of 157.3   shader clock %6.0f MHz -> %5.1f %% of the MFMA rate at that clock\n", name,
           wg_per_cu, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100, t / (ms * 1e3), flop / ms / 1e9 / (256 * 256 * (t / (ms * 1e3)) * 1e-6) * 100);
}

int main() {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 1024 * 256 * sizeof(float)); hipMalloc(&ticks, 1024 * sizeof(unsigned long long));
    for (int w : {1, 2, 4}) run<false>("R registers", w, out, ticks);
    for (int w : {1, 2, 4}) run<true>("L LDS-fed  ", w, out, ticks);
    return 0;
}
