// Dependent-accumulator latency of v_mfma_f32_4x4x1_16b_f32 (and of 16x16x4 for reference): the question behind a wave-private
// rollout step (4 rows x 64 hidden columns per wave, k strictly sequential).  hipcc -O3 --offload-arch=gfx950 mfma4x4_chain.hip && ./a.out
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND, int CHAINS>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[c], 0, 0, 0);
                else acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND, int CHAINS> void run(const char* name) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 64 * 8);
    const int iters = 4096;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(1), dim3(64), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h;
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s chains %d: %.2f cycles per MFMA (one wave)\n", name, CHAINS, (double)h / ((double)iters * 16 * CHAINS));
}
int main() {
    run<0, 1>("v_mfma_f32_4x4x1_16b_f32"); run<0, 2>("v_mfma_f32_4x4x1_16b_f32"); run<0, 4>("v_mfma_f32_4x4x1_16b_f32");
    run<1, 1>("v_mfma_f32_16x16x4_f32"); run<1, 2>("v_mfma_f32_16x16x4_f32");
    return 0;
}
