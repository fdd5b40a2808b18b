// Sustained MFMA issue rate and shader clock on gfx950: bf16 16x16x32 vs f32 16x16x4, 1 or 2 waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 tools/src/mfma_peak.hip -o tools/bin/mfma_peak ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((__vector_size__(16)));

template <int MODE>
__global__ void k(float* out, long long* clk, int iters)
{
    f32x4 acc[4] = {};
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f - i * 0.01f); }
    float fa = threadIdx.x * 0.001f, fb = 0.5f;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 6; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (MODE == 0) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[c], 0, 0, 0);
                else acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[c], 0, 0, 0);
            }
    }
    long long t1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int MODE>
void run(const char* name, int threads, double flop_per_mfma)
{
    const int blocks = 256, iters = 20000;
    float* out; long long* clk;
    hipMalloc(&out, blocks * threads * 4); hipMalloc(&clk, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, clk, 1000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, clk, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    double mf = (double)blocks * (threads / 64) * iters * 24;
    printf("%-28s %d waves/CU: %.3f ms  %.1f TFLOP/s  clock64/wall = %.3f (x100 MHz => %.0f MHz), cycles per MFMA per SIMD %.2f\n", name, threads / 64, ms,
           mf * flop_per_mfma / ms / 1e9, (double)h[0] / h[1], 100.0 * h[0] / h[1], (double)h[0] / (iters * 24.0 * (threads / 256)));
    hipFree(out); hipFree(clk);
}

int main()
{
    run<0>("bf16 16x16x32", 256, 16384.0);
    run<0>("bf16 16x16x32", 512, 16384.0);
    run<1>("f32 16x16x4", 256, 2048.0);
    run<1>("f32 16x16x4", 512, 2048.0);
    return 0;
}
