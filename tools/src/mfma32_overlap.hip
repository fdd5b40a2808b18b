// What does one filler instruction between two v_mfma_f32_32x32x2_f32 cost?  One wave per SIMD (256 threads per CU), 16 MFMAs per loop
// body on four rotating accumulators, K fillers of one kind after every MFMA.  Prints cycles per MFMA (64 = the matrix pipe's pace).
// hipcc --offload-arch=gfx950 -O3 tools/src/mfma32_overlap.hip -o tools/bin/mfma32_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int K>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* gsrc, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[16384];
    f32x16 acc[4] = {};
    float fa = threadIdx.x * 0.001f, fb = 1.0f - threadIdx.x * 0.002f;
    float v[8];
    f32x4 q[4];
    for (int i = 0; i < 8; ++i) v[i] = fa * i;
    for (int i = 0; i < 4; ++i) q[i] = (f32x4){fa, fb, fa * i, fb * i};
    const unsigned laddr = (threadIdx.x & 255) * 16;
    const float* gp = gsrc + (threadIdx.x & 63) * 4 + (blockIdx.x & 7) * 1024;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            acc[c & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[c & 3], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < K; ++j) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[(c + j) & 7]) : "v"(v[(c + j + 3) & 7]));
                if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&q[(c + j) & 3]) : "v"(*(double*)&q[(c + j + 1) & 3]));
                if (KIND == 2) asm volatile("v_accvgpr_read_b32 %0, a255" : "=v"(v[(c + j) & 7]));
                if (KIND == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(q[(c + j) & 3]) : "v"(laddr));
                if (KIND == 4) asm volatile("ds_write_b128 %0, %1" :: "v"(laddr), "v"(q[(c + j) & 3]));
                if (KIND == 5) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(q[(c + j) & 3]) : "v"(gp));
                if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[(c + j) & 7]) : "v"(v[(c + j + 3) & 7]));
                if (KIND == 7) asm volatile("s_nop 0");
            }
        }
        if (KIND == 3 || KIND == 4 || KIND == 5) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][7];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += q[i][0] + q[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + lds[threadIdx.x];
}

template <int KIND, int K>
void run(float* out, float* src, const char* name)
{
    const int blocks = 256, iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, K>), dim3(blocks), dim3(256), 0, 0, out, src, 2000);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, K>), dim3(blocks), dim3(256), 0, 0, out, src, iters);
    (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 16);
    printf("%-22s K=%d per MFMA: %6.1f cycles per MFMA at 2.4 GHz  (+%.1f per filler)\n", name, K, cyc, K ? (cyc - 64.0) / K : 0.0);
}

int main()
{
    float *out, *src;
    (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&src, 1 << 20); (void)hipMemset(src, 0, 1 << 20);
    run<0, 0>(out, src, "none");
    run<0, 1>(out, src, "v_add_f32"); run<0, 2>(out, src, "v_add_f32"); run<0, 4>(out, src, "v_add_f32"); run<0, 8>(out, src, "v_add_f32");
    run<6, 2>(out, src, "v_fma_f32"); run<6, 4>(out, src, "v_fma_f32");
    run<1, 1>(out, src, "v_pk_add_f32"); run<1, 2>(out, src, "v_pk_add_f32"); run<1, 4>(out, src, "v_pk_add_f32");
    run<2, 1>(out, src, "v_accvgpr_read"); run<2, 2>(out, src, "v_accvgpr_read"); run<2, 4>(out, src, "v_accvgpr_read");
    run<3, 1>(out, src, "ds_read_b128"); run<3, 2>(out, src, "ds_read_b128");
    run<4, 1>(out, src, "ds_write_b128"); run<4, 2>(out, src, "ds_write_b128");
    run<5, 1>(out, src, "global_load_dwordx4 L1"); run<5, 2>(out, src, "global_load_dwordx4 L1");
    run<7, 4>(out, src, "s_nop 0"); run<7, 8>(out, src, "s_nop 0");
    return 0;
}
