// conv1x1.hip -- 1x1 convolution (stride 1 | 2) of an NHWC map with the folded BatchNorm, the residual add and the ReLU in its
// epilogue, on gfx950 fp32 MFMA: the bottleneck convolutions of the semantic branch's ResNet (hybrid_models/resnet_encoder.py:40-51
// over torchvision's Bottleneck: conv1 / conv3 / downsample[0], each followed by BatchNorm2d, conv3's sum with the shortcut and
// the block's ReLU).  SURVEY.md §8(f) rank 3.  One launch replaces a library GEMM + a separate BatchNorm / add / ReLU pass over the
// output map.
//
// A GEMM D[cout][pixel] = W[cout][cin] . X[pixel][cin]^T on v_mfma_f32_16x16x4_f32, TRANSPOSED like the other kernels of this
// library (weights as the A operand, pixels as B): lane (g, i) of an accumulator holds output channels n0 + 4g .. 4g+3 of pixel
// m0 + i -- the epilogue is one 16-byte residual load, four FMAs / maxes and one 16-byte store per accumulator.
//   * operands come STRAIGHT from L1 / L2 in MFMA layout, no LDS and no barrier: the k index of lane group g at k-step e of a
//     16-channel chunk is channel 4g + e on both sides, so a lane's operand quad is the 16 bytes at X[pixel][k0 + 4g] (NHWC rows
//     are contiguous in the channel) resp. W[cout][k0 + 4g] (the Conv2d weight exactly as it lies in memory: nothing to pack);
//   * a wave owns a (16 TM pixels) x (16 TN channels) block: per 16-channel chunk TM + TN 16-byte loads feed 4 TM TN MFMAs
//     (TM = TN = 4: 64 MFMAs per 8 loads); the next chunk's quads are requested before the current chunk's MFMAs;
//   * the four waves of a workgroup take neighbouring channel blocks of the same pixels (their pixel loads meet in L1); maps with
//     few pixels (layer3 / layer4 at 30x40 / 15x20) take 32 x 32 blocks so that there are enough waves to go round;
//   * stride 2 (the downsample convolutions): output pixel (y, x) reads input pixel (2y, 2x) -- an address computation, no gather pass.
// The layers with few input channels at full resolution (64 -> 256 at 120x160) are HBM-bound: 74 MB per launch of 3 images.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "estd_hip.h"
#include "estd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;

__device__ __forceinline__ float4 as_float4(u32x4 v) { float4 f; __builtin_memcpy(&f, &v, 16); return f; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}

// SK = 1: the four waves of a workgroup own four neighbouring blocks.  SK = 4 (K-heavy layers on small maps: few blocks, long K): the
// four waves of a workgroup share ONE block and split its input channels into four contiguous ranges; waves 1..3 hand their partial
// sums to wave 0 through LDS, which adds them in a fixed order (deterministic) and runs the epilogue.  Larger blocks per wave = fewer
// operand bytes per MFMA from L2, which is what bounds this kernel.
template <int TM, int TN, int PF, int SK>
__global__ __launch_bounds__(256) void conv1x1_nhwc_kernel(const estd_conv1x1_desc p, int Ho, int Wo, int tiles_m, int tiles_n)
{
    __shared__ float4 red[SK > 1 ? (SK - 1) * TN * TM * 64 : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int wt = SK > 1 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;      // wave tile: channel block fastest
    if (wt >= tiles_m * tiles_n) return;                        // (workgroup-uniform when SK > 1)
    const int tn = wt % tiles_n, tm = wt / tiles_n;
    const int m0 = tm * 16 * TM, n0 = tn * 16 * TN;
    const int Mtot = p.N * Ho * Wo;
    const int cin = p.cin, cout = p.cout;

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.in, (size_t)p.N * p.H * p.W * cin * 4);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (size_t)cout * cin * 4);
    unsigned xoff[TM], woff[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int m = m0 + 16 * t + i;
        const int x = m % Wo, r = m / Wo;
        const int y = r % Ho, n = r / Ho;
        xoff[t] = m < Mtot ? (unsigned)((((size_t)n * p.H + y * p.stride) * p.W + x * p.stride) * cin + 4 * g) * 4u : OOB_OFFSET;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) woff[t] = (unsigned)((n0 + 16 * t + i) * cin + 4 * g) * 4u;       // cout is a multiple of 16 TN

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ring of PF chunks: PF - 1 chunks of operand quads are in flight while one is multiplied (small blocks have few MFMAs per chunk
    // and few waves per SIMD: they need several L2 round trips of cover; the registers are there)
    float4 xq[PF][TM], wq[PF][TN];
    auto load_chunk = [&](int k0, float4 (&xd)[TM], float4 (&wd)[TN]) {
#pragma unroll
        for (int t = 0; t < TN; ++t) wd[t] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, woff[t], k0 * 4, 0));
#pragma unroll
        for (int t = 0; t < TM; ++t) xd[t] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff[t], k0 * 4, 0));
    };
    auto mfma_chunk = [&](const float4 (&xs)[TM], const float4 (&ws)[TN]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const float wv = e == 0 ? ws[a].x : e == 1 ? ws[a].y : e == 2 ? ws[a].z : ws[a].w;
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const float xv = e == 0 ? xs[b].x : e == 1 ? xs[b].y : e == 2 ? xs[b].z : xs[b].w;
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, acc[a][b], 0, 0, 0);
                }
            }
    };
    const int nchunks = (cin >> 4) / SK;                        // this wave's share of the input channels: chunks kbase .. kbase + nchunks
    const int kbase = SK > 1 ? wave * nchunks : 0;
#pragma unroll
    for (int j = 0; j < PF - 1; ++j)
        if (j < nchunks) load_chunk((kbase + j) * 16, xq[j], wq[j]);
    for (int c = 0; c < nchunks; c += PF) {                     // PF chunks per trip: the ring slots are compile-time constants
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int cc = c + j;                                // (wave-uniform conditions)
            if (cc + PF - 1 < nchunks) load_chunk((kbase + cc + PF - 1) * 16, xq[(j + PF - 1) % PF], wq[(j + PF - 1) % PF]);
            if (cc < nchunks) mfma_chunk(xq[j], wq[j]);
        }
    }
    if (SK > 1) {
        if (wave > 0) {
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    red[(((wave - 1) * TN + a) * TM + b) * 64 + lane] = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < SK - 1; ++w)
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const float4 r = red[((w * TN + a) * TM + b) * 64 + lane];
                    acc[a][b] += (f32x4){r.x, r.y, r.z, r.w};
                }
    }

    // ---- epilogue: folded BatchNorm, + residual, ReLU; 16-byte stores ----
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, (size_t)Mtot * cout * 4);
    const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.residual ? p.residual : p.out, (size_t)Mtot * cout * 4);
    const float floor_ = p.relu ? 0.0f : ESTD_NO_FLOOR;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int cb = n0 + 16 * a + 4 * g;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + cb);
        if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + cb);
        unsigned ooff[TM];
        float4 res[TM];
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const int m = m0 + 16 * b + i;
            ooff[b] = m < Mtot ? (unsigned)((size_t)m * cout + cb) * 4u : OOB_OFFSET;
        }
        if (p.residual) {
#pragma unroll
            for (int b = 0; b < TM; ++b) res[b] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_r, ooff[b], 0, 0));
        }
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            float4 v;
            v.x = fmaf(acc[a][b][0], sc.x, sh.x); v.y = fmaf(acc[a][b][1], sc.y, sh.y);
            v.z = fmaf(acc[a][b][2], sc.z, sh.z); v.w = fmaf(acc[a][b][3], sc.w, sh.w);
            if (p.residual) { v.x += res[b].x; v.y += res[b].y; v.z += res[b].z; v.w += res[b].w; }
            v.x = fmaxf(v.x, floor_); v.y = fmaxf(v.y, floor_); v.z = fmaxf(v.z, floor_); v.w = fmaxf(v.w, floor_);
            u32x4 bits;
            __builtin_memcpy(&bits, &v, 16);
            __builtin_amdgcn_raw_buffer_store_b128(bits, rs_o, ooff[b], 0, 0);
        }
    }
}

template <int TM, int TN, int PF, int SK>
int launch1x1(const estd_conv1x1_desc& d, int Ho, int Wo, hipStream_t stream)
{
    const long long Mtot = (long long)d.N * Ho * Wo;
    const int tiles_m = (int)((Mtot + 16 * TM - 1) / (16 * TM)), tiles_n = d.cout / (16 * TN);
    const long long wts = (long long)tiles_m * tiles_n;
    const unsigned grid = SK > 1 ? (unsigned)wts : (unsigned)((wts + 3) / 4);
    hipLaunchKernelGGL((conv1x1_nhwc_kernel<TM, TN, PF, SK>), dim3(grid), dim3(256), 0, stream, d, Ho, Wo, tiles_m, tiles_n);
    return ESTD_LAUNCH_CHECK();
}

}  // namespace

extern "C" int estd_conv1x1_nhwc(const estd_conv1x1_desc* dp, estd_stream_t s)
{
    if (!dp) return ESTD_ERR_ARG;
    const estd_conv1x1_desc& d = *dp;
    if (d.N <= 0 || d.H <= 0 || d.W <= 0 || !d.in || !d.w || !d.out) return ESTD_ERR_ARG;
    if (d.stride != 1 && d.stride != 2) return ESTD_ERR_UNSUPPORTED;
    if (d.cin < 16 || (d.cin & 15) || d.cout < 32 || (d.cout & 31)) return ESTD_ERR_UNSUPPORTED;
    const int Ho = (d.H - 1) / d.stride + 1, Wo = (d.W - 1) / d.stride + 1;           // kernel 1, padding 0
    const long long Mtot = (long long)d.N * Ho * Wo;
    // 32-bit byte offsets inside the input map, the output map and the weight matrix
    if ((long long)d.N * d.H * d.W * d.cin * 4 >= 0x7fffff00LL || Mtot * d.cout * 4 >= 0x7fffff00LL || (long long)d.cin * d.cout * 4 >= 0x7fffff00LL)
        return ESTD_ERR_UNSUPPORTED;
    hipStream_t stream = estd_stream(s);
    // Block per wave: the largest one that still gives the device ~one wave per SIMD (tools/conv1x1_bench.py).  ESTD_C1X1_CFG = 100 TM + 10 TN + SK
    // forces one (A/B).
    static const int cfg_env = [] { const char* e = getenv("ESTD_C1X1_CFG"); return e ? atoi(e) : 0; }();
    const long long want = (long long)estd_device_cus() * 7 / 2;                        // 896 on 256 CUs
    auto tiles = [&](int tm, int tn) { return ((Mtot + 16 * tm - 1) / (16 * tm)) * (d.cout / (16 * tn)); };
    const bool sk4 = (d.cin & 63) == 0 && d.cin >= 256;                                 // four K ranges of whole chunks, long enough to be worth the exchange
    int cfg = cfg_env;
    if (cfg == 0) {
        const bool c64 = (d.cout & 63) == 0;
        if (c64 && tiles(4, 4) >= want) cfg = 441;
        else if (sk4 && d.cin >= 1024 && c64 && tiles(4, 4) >= want / 4) cfg = 444;     // few blocks, long K: split K over the four waves
        else if (tiles(4, 2) >= want) cfg = 421;
        else if (sk4 && c64 && tiles(4, 4) >= want / 4) cfg = 444;
        else if (sk4 && c64 && tiles(2, 4) >= want / 4) cfg = 244;
        else if (tiles(2, 2) >= want) cfg = 221;
        else if (sk4 && tiles(2, 2) >= want / 4) cfg = 224;
        else cfg = 121;
    }
    if ((cfg % 10) == 4 && !((d.cin & 63) == 0)) cfg = cfg - 3;                         // SK needs cin % 64 == 0
    if ((cfg / 10) % 10 == 4 && (d.cout & 63)) cfg -= 20;                               // TN = 4 needs cout % 64 == 0
    switch (cfg) {
    case 441: return launch1x1<4, 4, 2, 1>(d, Ho, Wo, stream);
    case 421: return launch1x1<4, 2, 3, 1>(d, Ho, Wo, stream);
    case 241: return launch1x1<2, 4, 3, 1>(d, Ho, Wo, stream);
    case 221: return launch1x1<2, 2, 4, 1>(d, Ho, Wo, stream);
    case 444: return launch1x1<4, 4, 2, 4>(d, Ho, Wo, stream);
    case 424: return launch1x1<4, 2, 3, 4>(d, Ho, Wo, stream);
    case 244: return launch1x1<2, 4, 3, 4>(d, Ho, Wo, stream);
    case 224: return launch1x1<2, 2, 4, 4>(d, Ho, Wo, stream);
    case 124: return launch1x1<1, 2, 4, 4>(d, Ho, Wo, stream);
    default: return launch1x1<1, 2, 8, 1>(d, Ho, Wo, stream);
    }
}
