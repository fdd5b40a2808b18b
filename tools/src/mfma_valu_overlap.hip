// Does VALU work hide behind fp32 MFMAs on gfx950?  A wave issues groups of 8 independent v_mfma_f32_16x16x4_f32 (4 rotating
// accumulators) with K plain / packed fp32 adds between the groups; 1 or 2 waves per SIMD; time per MFMA vs K.
// hipcc --offload-arch=gfx950 -O3 tools/src/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((__vector_size__(16)));

template <int K, int PK, int BF, int NACC = 4>
__global__ void k(float* out, int iters)
{
    f32x4 acc[4] = {};
    float fa = threadIdx.x * 0.001f, fb = 1.0f - threadIdx.x * 0.002f;
    bf16x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(fa + i); b8[i] = (__bf16)(fb - i); }
    float v[16];
    f32x2 p[8];
    for (int i = 0; i < 16; ++i) v[i] = fa * i;
    for (int i = 0; i < 8; ++i) p[i] = (f32x2){fa * i, fb * i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if (BF) acc[c % NACC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[c % NACC], 0, 0, 0);
                else acc[c % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[c % NACC], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (PK) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j & 7]) : "v"(p[(j + 3) & 7])); }
                else { asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j & 15]) : "v"(v[(j + 5) & 15])); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][3];
    for (int i = 0; i < 16; ++i) s += v[i];
    for (int i = 0; i < 8; ++i) s += p[i][0] + p[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int K, int PK, int BF, int NACC = 4>
void run(int threads)
{
    const int blocks = 256, iters = 200000;
    float* out;
    (void)hipMalloc(&out, blocks * threads * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<K, PK, BF, NACC>), dim3(blocks), dim3(threads), 0, 0, out, 1000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<K, PK, BF, NACC>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 32 * (threads / 256);
    printf("%s waves/SIMD %d  %d accumulators  %2d %s adds per 8 MFMAs: %7.2f ms  %6.1f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)\n", BF ? "bf16 16x16x32" : "f32 16x16x4 ",
           threads / 256, NACC, K, PK ? "packed" : "plain ", ms, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.4);
    (void)hipFree(out);
}

int main()
{
    for (int threads = 256; threads <= 512; threads += 256) {
        run<0, 0, 0>(threads); run<4, 0, 0>(threads); run<8, 0, 0>(threads); run<16, 0, 0>(threads); run<32, 0, 0>(threads);
        run<4, 1, 0>(threads); run<8, 1, 0>(threads); run<16, 1, 0>(threads);
        run<0, 0, 0, 2>(threads); run<0, 0, 0, 1>(threads); run<8, 0, 0, 2>(threads);     // dependent chains: 2 / 1 rotating accumulators
        run<0, 0, 1>(threads); run<8, 0, 1>(threads); run<16, 0, 1>(threads); run<8, 1, 1>(threads);
    }
    return 0;
}
