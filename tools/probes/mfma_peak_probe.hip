// mfma_peak_probe.hip -- what fp32 MFMA rate and shader clock does one MI355X sustain (NOT part of the product)?
//   R  operands in registers: nothing but dependent v_mfma_f32_32x32x2_f32 chains (4 accumulators per wave)
//   L  both operands re-read from LDS once per 16 MFMAs (ds_read_b128; 4 accumulators share a fragment pair)
//   N  both operands re-read from LDS for every 4 MFMAs: 2 x ds_read_b128 per 4 MFMAs, exactly cm_mlp_kernel.h's rowpar_nt
//   W / V / G / A  = N plus, per 4 MFMAs: 2 ds_write_b32 (activation stores) / 12 VALU ops (epilogue math) / one 16-byte global
//       load per 16 MFMAs (tile prefetch) / all three -- the side activity of the fused training kernel
//   P  phases: 128 MFMAs from registers, then an LDS-only phase (32 ds_write_b32 + 32 ds_read_b32, wave-private slice): alone a wave keeps the
//      pipe ~2/3 busy, two waves per SIMD fill each other's LDS phases -- the structure of the fused kernels
//   C  both operands as scalar reads: 8 x ds_read_b32 per 4 MFMAs, exactly cm_mlp_kernel.h's colred / the B side of rowpar_tn
// for 1, 2 and 4 waves per SIMD.  s_memtime counts shader clocks, so ticks / wall time is the clock the run sustained.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/peak tools/probes/mfma_peak_probe.hip && /tmp/peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDS>
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* ticks, int iters, int pad_floats, const float4* __restrict__ gsrc) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    for (int i = threadIdx.x; i < 64 * 68; i += 256) {
        unsigned u = (unsigned)i * 2654435761u + blockIdx.x * 40503u; u ^= u >> 13; u *= 2246822519u; u ^= u >> 16;
        sm[i] = pad_floats ? ((float)(u & 0xFFFFFF) / 16777216.0f - 0.5f) : 1e-3f * (i % 97);  // pad_floats != 0: high-entropy operands
    }
    __syncthreads();
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int g = 0; g < 16; ++g) acc[q][g] = 0.f;
    float4 a = *reinterpret_cast<const float4*>(sm + r * 68 + 4 * h), b = *reinterpret_cast<const float4*>(sm + (32 + r) * 68 + 4 * h);
    float side = 1.0f + lane;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (LDS == 8) {
            float* mine = sm + 64 * 68 + (threadIdx.x >> 6) * 2048;
#pragma unroll
            for (int v = 0; v < 32; ++v) mine[64 * v + lane] = side + v;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int v = 0; v < 32; ++v) side += mine[64 * v + (lane ^ 1)];
            a.x += side * 1e-9f;
        }
        asm volatile("" ::: "memory");  // LDS contents are "unknown" again: the operand reads below stay inside the loop
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (LDS == 1) {
                a = *reinterpret_cast<const float4*>(sm + r * 68 + 8 * j + 4 * h);
                b = *reinterpret_cast<const float4*>(sm + (32 + r) * 68 + 8 * j + 4 * h);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (LDS >= 4) {  // side activity next to the rowpar_nt-style operand reads
                    if (LDS == 4 || LDS == 7) {
                        sm[64 * 68 + (threadIdx.x >> 6) * 2048 + 64 * ((j + q) & 15) + lane] = side;
                        sm[64 * 68 + (threadIdx.x >> 6) * 2048 + 1024 + 64 * ((j + q) & 15) + lane] = side * 0.5f;
                    }
                    if (LDS == 5 || LDS == 7) {
#pragma unroll
                        for (int v = 0; v < 6; ++v) side = fmaxf(fmaf(side, 1.0001f, 0.25f), -side);
                    }
                    if ((LDS == 6 || LDS == 7) && q == 0) {
                        const float4 gv = gsrc[(((size_t)blockIdx.x * iters + it) % 120000) * 2048 + j * 256 + threadIdx.x];
                        side += gv.x + gv.w;
                    }
                }
                if (LDS == 2 || LDS >= 4) {
                    a = *reinterpret_cast<const float4*>(sm + r * 68 + 8 * ((j + q) & 7) + 4 * h);
                    b = *reinterpret_cast<const float4*>(sm + (32 + r) * 68 + 8 * ((j + 2 * q) & 7) + 4 * h);
                }
                if (LDS == 3) {
                    const float* ap = sm + (8 * ((j + q) & 3) + h) * 68 + r;
                    const float* bp = sm + (32 + 8 * ((j + 2 * q) & 3) + h) * 68 + r;
                    a = make_float4(ap[0], ap[2 * 68], ap[4 * 68], ap[6 * 68]);
                    b = make_float4(bp[0], bp[2 * 68], bp[4 * 68], bp[6 * 68]);
                }
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
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s + (float)pad_floats + side;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

// the 16x16x4 form (same FLOP rate): 16 independent accumulator chains from registers
__global__ __launch_bounds__(256) void k16(float* out, unsigned long long* ticks, int iters) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 acc[16];
    for (int q = 0; q < 16; ++q) acc[q] = f32x4{0, 0, 0, 0};
    float a = 1e-3f * threadIdx.x, b = 2e-3f * threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int q = 0; q < 16; ++q) s += acc[q][0] + acc[q][3];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int LDS>
static void run(const char* name, int wg_per_cu, float* out, unsigned long long* ticks, const float4* gsrc, int entropy = 0) {
    // LDS padding so that exactly wg_per_cu workgroups (4 waves each = wg_per_cu waves per SIMD) fit a CU
    const size_t lds = wg_per_cu == 1 ? 100 * 1024 : 70 * 1024;  // >= 64*68*4 + 32 KB of store scratch
    hipFuncSetAttribute((const void*)k<LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const int grid = 256 * wg_per_cu, iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<LDS>, dim3(grid), dim3(256), lds, 0, out, ticks, iters, entropy, gsrc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<LDS>, dim3(grid), dim3(256), lds, 0, out, ticks, iters, entropy, gsrc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long th[1024];
    hipMemcpy(th, ticks, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    unsigned long long t = 0, tmin = ~0ull;
    for (int i = 0; i < grid; ++i) { t = th[i] > t ? th[i] : t; tmin = th[i] < tmin ? th[i] : tmin; }
    const double cover = (double)tmin / (double)t;  // < 1: some workgroups ran for only part of the launch
    const double flop = (double)grid * 4 * iters * 8 * 16 * 4096.0;
    printf("%s  %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s = %5.1f %% of 157.3   shader clock %6.0f MHz (min/max workgroup window %.2f) -> %5.1f %% of the MFMA rate at that clock\n", name,
           wg_per_cu, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100, t / (ms * 1e3), cover, flop / ms / 1e9 / (256 * 256 * (t / (ms * 1e3)) * 1e-6) * 100);
}

int main() {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 1024 * 256 * sizeof(float)); hipMalloc(&ticks, 1024 * sizeof(unsigned long long));
    float4* gsrc; hipMalloc(&gsrc, (size_t)120001 * 2048 * sizeof(float4));  // 3.9 GB window, indexed modulo
    for (int w : {1, 2}) run<0>("R registers          ", w, out, ticks, gsrc);
    for (int w : {1, 2}) run<1>("L b128 per 16 MFMAs  ", w, out, ticks, gsrc);
    for (int w : {1, 2}) run<2>("N 2 b128 per 4 MFMAs ", w, out, ticks, gsrc);
    for (int w : {1, 2}) run<3>("C 8 b32 per 4 MFMAs  ", w, out, ticks, gsrc);
    for (int w : {1, 2}) run<4>("W N + LDS stores     ", w, out, ticks, gsrc);
    for (int w : {1, 2}) run<5>("V N + VALU           ", w, out, ticks, gsrc);
    for (int w : {1, 2}) run<6>("G N + global loads   ", w, out, ticks, gsrc);
    for (int w : {1, 2}) run<7>("A N + all three      ", w, out, ticks, gsrc);
    for (int w : {1, 2}) run<8>("P MFMA / LDS phases  ", w, out, ticks, gsrc);
    for (int w : {1, 2}) run<2>("N, random operands   ", w, out, ticks, gsrc, 1);
    for (int w : {1, 2}) run<3>("C, random operands   ", w, out, ticks, gsrc, 1);
    for (int w : {1, 2}) run<8>("P, random operands   ", w, out, ticks, gsrc, 1);
    {
        const int grid = 256, iters = 4000;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k16, dim3(grid), dim3(256), 0, 0, out, ticks, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k16, dim3(grid), dim3(256), 0, 0, out, ticks, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long th[256];
        hipMemcpy(th, ticks, grid * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        unsigned long long t = 0; for (int i = 0; i < grid; ++i) t = th[i] > t ? th[i] : t;
        const double flop = (double)grid * 4 * iters * 256 * 2048.0;
        printf("R16 registers, 16x16x4 1 waves/SIMD: %8.3f ms  %7.1f TFLOP/s = %5.1f %% of 157.3   shader clock %6.0f MHz\n", ms, flop / ms / 1e9,
               flop / ms / 1e9 / 157.3 * 100, t / (ms * 1e3));
    }
    return 0;
}
