// Sustained MFMA issue rate and shader clock on gfx950 under a LONG load (DVFS settles): fp32 16x16x4 vs 32x32x2,
// bf16 16x16x32 vs 32x32x16, 2 waves per SIMD, pseudo-random operands.
// hipcc --offload-arch=gfx950 -O3 tools/src/mfma_peak.hip -o tools/bin/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((__vector_size__(16)));

__device__ __forceinline__ float rnd(unsigned s) { s = s * 1664525u + 1013904223u; s ^= s >> 13; return (float)(s & 0xffff) * (1.0f / 32768.0f) - 1.0f; }

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, long long* clk, int iters)
{
    f32x4 acc4[4] = {};
    f32x16 acc16[2] = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)rnd(threadIdx.x * 8 + i); b[i] = (__bf16)rnd(threadIdx.x * 8 + i + 77777); }
    float fa = rnd(threadIdx.x + 5), fb = rnd(threadIdx.x + 999);
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) { for (int c = 0; c < 4; ++c) acc4[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[c], 0, 0, 0); }
            if (MODE == 1) { for (int c = 0; c < 4; ++c) acc4[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc4[c], 0, 0, 0); }
            if (MODE == 2) { for (int c = 0; c < 2; ++c) acc16[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc16[c], 0, 0, 0); }
            if (MODE == 3) { for (int c = 0; c < 2; ++c) acc16[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc16[c], 0, 0, 0); }
        }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int c = 0; c < 4; ++c) s += acc4[c][0] + acc4[c][3];
    for (int c = 0; c < 2; ++c) s += acc16[c][0] + acc16[c][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE>
void run(const char* name, double flop_per_iter_per_wave, int iters)
{
    const int blocks = 256, threads = 512;
    float* out; long long* clk;
    (void)hipMalloc(&out, blocks * threads * 4); (void)hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, clk, 1000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double fl = (double)blocks * (threads / 64) * iters * flop_per_iter_per_wave;
    printf("%-22s %8.1f ms  %8.1f TFLOP/s  shader clock %.0f MHz\n", name, ms, fl / ms / 1e9, 100.0 * h[0] / h[1]);
    (void)hipFree(out); (void)hipFree(clk);
}

int main()
{
    // ~200-300 ms each: long enough for the power management to settle
    run<1>("f32 16x16x4", 32 * 2048.0, 1800000);
    run<2>("f32 32x32x2", 16 * 4096.0, 1800000);
    run<1>("f32 16x16x4 (again)", 32 * 2048.0, 1800000);
    run<0>("bf16 16x16x32", 32 * 16384.0, 3000000);
    run<3>("bf16 32x32x16", 16 * 32768.0, 3000000);
    return 0;
}
