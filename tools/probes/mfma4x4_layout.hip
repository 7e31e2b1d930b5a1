// Operand / result layout of v_mfma_f32_4x4x1_16b_f32 (16 independent 4x4 outer products per instruction), checked on the device:
//   hypothesis  D[reg r][lane l] = A[lane 4 (l / 4) + r] * B[lane l]        (block b = l / 4: row i = r in registers, column j = l % 4 in lanes)
// and the issue cost of the head's three products in that form (k_mlp's wave-private head, cm_mlp_kernel.h): cycles per 32 MFMAs with
// 2 / 4 accumulator chains, with b128 LDS operand reads in between.   hipcc -O3 --offload-arch=gfx950 mfma4x4_layout.hip -o _bin/mfma4x4_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_layout(float* out) {
    const int l = threadIdx.x;
    const float a = 1.0f + l, b = 100.0f + l;
    f32x4 d = {0, 0, 0, 0};
    d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[r * 64 + l] = d[r];
}
template <int CH>
__global__ void k_rate(float* out, unsigned long long* cyc, int iters) {
    __shared__ float lds[64 * 68];
    for (int i = threadIdx.x; i < 64 * 68; i += 64) lds[i] = 1e-3f * i;
    __syncthreads();
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = f32x4{0, 0, 0, 0};
    const float4* p = reinterpret_cast<const float4*>(lds + (threadIdx.x & 15) * 68 + 4 * (threadIdx.x >> 4));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {   // 8 x (2 b128 reads + 4 MFMAs) = the logits product of one 16-row slice
            const float4 a = p[4 * m], b = p[4 * m + 17 * 16];
            acc[0 % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.x, b.x, acc[0 % CH], 0, 0, 0);
            acc[1 % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.y, b.y, acc[1 % CH], 0, 0, 0);
            acc[2 % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.z, b.z, acc[2 % CH], 0, 0, 0);
            acc[3 % CH] = __builtin_amdgcn_mfma_f32_4x4x1f32(a.w, b.w, acc[3 % CH], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int CH> void rate() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 1024); hipMalloc(&cyc, 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_rate<CH>), dim3(1), dim3(64), 0, 0, out, cyc, 2048);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("32 x v_mfma_f32_4x4x1 + 16 ds_read_b128, %d chains: %.1f cycles (%.2f per MFMA)\n", CH, (double)h / 2048, (double)h / 2048 / 32);
}
int main() {
    float* out; hipMalloc(&out, 4 * 64 * 4);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, out);
    float h[256]; hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int r = 0; r < 4; ++r) for (int l = 0; l < 64; ++l) if (h[r * 64 + l] != (1.0f + 4 * (l / 4) + r) * (100.0f + l)) ++bad;
    printf("layout D[r][l] = A[4 (l / 4) + r] * B[l]: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    if (bad) for (int r = 0; r < 4; ++r) { for (int l = 0; l < 8; ++l) printf(" %9.0f", h[r * 64 + l]); printf("\n"); }
    rate<1>(); rate<2>(); rate<4>();
    return 0;
}
