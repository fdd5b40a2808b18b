// 2D refinement tail of DepthHybridDecoder (hybrid_depth_decoder.py:267-290 / :392-415): the glue around its 3x3 convolutions,
// HBM bound, one pass each (gfx950 only).
//
//   estd_planes_cat_nhwc      cat([semantic_vs, relu(all_fused_logits)], 1) (:268) of two NCHW plane stacks, written directly as the
//                             NHWC map the next convolution reads (was: cat kernel + channels-last copy)
//   estd_upsample2_cat_nhwc   cat([upsample(x), skip], 1) (:269-272): nearest x2 of an NHWC map beside its full-resolution skip
//   estd_disp_head_nhwc       depth_max * sigmoid(Conv2d(C, 1, 3, padding 1, bias)(x)) (:274, :279), optionally nearest x2 (:274
//                             F.interpolate(scale_factor=2)): C -> 1 channels is a GEMV per pixel -- one thread per pixel on the VALU
//   estd_normalise_nhwc       imgs -> 2 * (imgs / 255) - 1 (model_hybrid.py:119) written as the NHWC image batch the 2D networks read
//                             (was: three elementwise kernels + a channels-last copy); same three roundings, contraction off
//   estd_conv2d_k3_to16_nhwc  ConvBlocks of the full-resolution end of the decoder (hybrid_depth_decoder.py:276-278: upconv_0_0 32 -> 16 at
//                             1/2 resolution, upconv_0_1 16 -> 16 on the nearest-x2 upsampled map): 16 output channels are one MFMA N
//                             tile; the upsampling is an address computation in the operand loads (no 59 MB intermediate), BN + ReLU
//                             in the epilogue, operands straight from L1/L2 (the whole input is 15-30 MB)
//   estd_stem3x3s2_nhwc       first layer of the PSM extractor (networks/psm_submodule.py:47: convbn(3, 32, 3, 2, 1, 1) + ReLU): 3 input
//                             channels are 27 multiplies per output channel -- one thread per output pixel on the VALU, the padded
//                             copy, library convolution and BatchNorm pass it replaces moved 5x the bytes
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "estd_hip.h"

#include "estd_common.h"

namespace {

// ---- [N][Ca][HW] and [N][Cb][HW] planes -> [N][HW][Ca + Cb] records, through an LDS tile so both sides coalesce ----
template <int PIX>            // pixels per block
__global__ __launch_bounds__(256) void planes_cat_nhwc_kernel(const float* __restrict__ a, int Ca, const float* __restrict__ b, int Cb,
                                                              int relu_b, float* __restrict__ out, long long HW)
{
    extern __shared__ float tile[];                 // [C][PIX + 1]
    const int C = Ca + Cb;
    const long long blocks_per_img = (HW + PIX - 1) / PIX;
    const long long n = blockIdx.x / blocks_per_img;
    const long long p0 = (blockIdx.x % blocks_per_img) * PIX;
    for (int e = threadIdx.x; e < C * PIX; e += 256) {
        const int c = e / PIX, k = e % PIX;
        float v = 0.f;
        if (p0 + k < HW) {
            if (c < Ca) v = a[(n * Ca + c) * HW + p0 + k];
            else {
                v = b[(n * Cb + (c - Ca)) * HW + p0 + k];
                if (relu_b) v = v > 0.f ? v : 0.f;
            }
        }
        tile[c * (PIX + 1) + k] = v;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * PIX; e += 256) {
        const int k = e / C, c = e % C;
        if (p0 + k < HW) out[(n * HW + p0 + k) * C + c] = tile[c * (PIX + 1) + k];
    }
}

// ---- [N][HW][C] records -> [N][C][HW] planes (the inverse direction), through the same LDS tile ----
template <int PIX>
__global__ __launch_bounds__(256) void nhwc_to_planes_kernel(const float* __restrict__ in, int C, float* __restrict__ out, long long HW)
{
    extern __shared__ float tile[];                 // [C][PIX + 1]
    const long long blocks_per_img = (HW + PIX - 1) / PIX;
    const long long n = blockIdx.x / blocks_per_img;
    const long long p0 = (blockIdx.x % blocks_per_img) * PIX;
    for (int e = threadIdx.x; e < C * PIX; e += 256) {
        const int k = e / C, c = e % C;
        tile[c * (PIX + 1) + k] = p0 + k < HW ? in[(n * HW + p0 + k) * C + c] : 0.f;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < C * PIX; e += 256) {
        const int c = e / PIX, k = e % PIX;
        if (p0 + k < HW) out[(n * C + c) * HW + p0 + k] = tile[c * (PIX + 1) + k];
    }
}

// ---- out[n][y][x] = cat(x[n][y/2][x/2][0..Cx), skip[n][y][x][0..Cs)); one thread per 16-byte chunk ----
__global__ __launch_bounds__(256) void upsample2_cat_nhwc_kernel(const float4* __restrict__ x, int Cx4, const float4* __restrict__ skip, int Cs4,
                                                                 float4* __restrict__ out, int N, int H, int W)
{
    const int C4 = Cx4 + Cs4;
    const long long total = (long long)N * H * W * C4;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C4);
    const long long pix = e / C4;
    const int xx = (int)(pix % W), yy = (int)((pix / W) % H);
    const long long n = pix / ((long long)W * H);
    float4 v;
    if (c < Cx4) v = x[((n * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1)) * Cx4 + c];
    else v = skip[pix * Cs4 + (c - Cx4)];
    out[e] = v;
}

// ---- depth head: one thread per input-resolution pixel; weights [9][C] in LDS ----
template <int C>
__global__ __launch_bounds__(256) void disp_head_nhwc_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ bias, float depth_max,
                                                             float* __restrict__ out, int N, int H, int W, int up)
{
    __shared__ float ws[9 * C];
    for (int e = threadIdx.x; e < 9 * C; e += 256) {           // w is [1][C][3][3] (Conv2d layout) -> [tap][c]
        const int t = e / C, c = e % C;
        ws[e] = w[c * 9 + t];
    }
    __syncthreads();
    const long long total = (long long)N * H * W;
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pix >= total) return;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const long long n = pix / ((long long)W * H);
    float acc = bias[0];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x + kx - 1;
            if ((unsigned)xx >= (unsigned)W) continue;
            const float4* src = reinterpret_cast<const float4*>(in + ((n * H + yy) * W + xx) * C);
            const float* wt = ws + (ky * 3 + kx) * C;
#pragma unroll
            for (int q = 0; q < C / 4; ++q) {
                const float4 v = src[q];
                acc = fmaf(v.x, wt[4 * q + 0], acc);
                acc = fmaf(v.y, wt[4 * q + 1], acc);
                acc = fmaf(v.z, wt[4 * q + 2], acc);
                acc = fmaf(v.w, wt[4 * q + 3], acc);
            }
        }
    }
    const float d = depth_max * (1.0f / (1.0f + expf(-acc)));
    if (up == 1) {
        out[pix] = d;
    } else {                                                     // nearest x2: the 2x2 block of the output
        const int W2 = 2 * W;
        float* o = out + (n * (2 * H) + 2 * y) * (long long)W2 + 2 * x;
        *reinterpret_cast<float2*>(o) = make_float2(d, d);
        *reinterpret_cast<float2*>(o + W2) = make_float2(d, d);
    }
}

// ---- 3x3 / stride 1 / padding 1, CIN (16 | 32) -> 16 channels, optional nearest x2 upsampling of the input, folded BN, ReLU ----
// One wave = a strip of ROWS output rows x 16 output pixels.  MFMA 16x16x4 f32, issued transposed (weights as A, pixels as B):
// lane (g, i) loads channels 16 q + 4 g .. + 3 of input pixel i (one 16-byte load per tap and 16-channel half) as the k-steps
// ks = 0..3 (K order permuted, weights packed to match) and ends up holding output channels 4 g .. 4 g + 3 of pixel i.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((__vector_size__(16)));
constexpr int TO16_ROWS = 8;

template <int CIN, bool UP>
__global__ __launch_bounds__(256) void conv2d_k3_to16_kernel(const float* __restrict__ in, const float4* __restrict__ wpk, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float* __restrict__ out, int N, int H, int W,
                                                             int strips_y, int segs_x)
{
    constexpr int Q = CIN / 16;
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);          // wave id -> (n, strip, segment)
    const long long nwaves = (long long)N * strips_y * segs_x;
    if (wid >= nwaves) return;
    const int seg = (int)(wid % segs_x);
    const int strip = (int)((wid / segs_x) % strips_y);
    const int n = (int)(wid / ((long long)segs_x * strips_y));
    const int Hin = UP ? H >> 1 : H, Win = UP ? W >> 1 : W;

    float4 wr[9][Q];                                          // this lane's weights: [tap][half] x 4 k-steps
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < Q; ++q) wr[t][q] = wpk[(t * Q + q) * 64 + lane];
    const float4 sc = reinterpret_cast<const float4*>(scale)[g], sh = reinterpret_cast<const float4*>(shift)[g];

    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (size_t)n * Hin * Win * CIN), 0, (int)((size_t)Hin * Win * CIN * 4), 0x00020000);
    const int x = seg * 16 + i;
    unsigned coloff[3];                                       // byte offset of the source column of tap kx (OOB: beyond the buffer)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int xx = x + kx - 1;
        coloff[kx] = ((unsigned)xx < (unsigned)W) ? (unsigned)((UP ? xx >> 1 : xx) * CIN + 4 * g) * 4u : 0xFFFFFF00u;
    }
    for (int r = 0; r < TO16_ROWS; ++r) {
        const int y = strip * TO16_ROWS + r;
        if (y >= H) break;                                    // wave-uniform
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            if ((unsigned)yy >= (unsigned)H) continue;        // wave-uniform: zero padding rows
            const int rowoff = (UP ? yy >> 1 : yy) * Win * CIN * 4;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(rs_in, coloff[kx], rowoff + q * 64, 0);
                    float4 a;
                    __builtin_memcpy(&a, &raw, 16);
                    const float4 wq = wr[ky * 3 + kx][q];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.x, a.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.y, a.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.z, a.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.w, a.w, acc, 0, 0, 0);
                }
            }
        }
        if (x < W) {
            float4 o;
            o.x = fmaxf(acc[0] * sc.x + sh.x, 0.f);
            o.y = fmaxf(acc[1] * sc.y + sh.y, 0.f);
            o.z = fmaxf(acc[2] * sc.z + sh.z, 0.f);
            o.w = fmaxf(acc[3] * sc.w + sh.w, 0.f);
            *reinterpret_cast<float4*>(out + (((size_t)n * H + y) * W + x) * 16 + 4 * g) = o;
        }
    }
}

// ---- the small convolutions of the PSM extractor that are neither 3x3 / stride 1 nor wide enough for the tiled kernels:
//      3x3 stride 2 (layer2[0].conv1, psm_submodule.py:52), 1x1 stride 1 | 2 (the downsample paths :78-83, the SPP branch
//      convolutions :100-110, lastconv's 1x1 :72-74).  Same scheme as conv2d_k3_to16_kernel: a wave owns 16 output pixels of a row x
//      one 16-channel output tile, weights of that tile in registers, operands straight from L1/L2 with one 16-byte load per tap and
//      16 input channels, transposed MFMAs (a lane stores 16 bytes), folded BN (+ ReLU) epilogue.  NHWC in / out.
template <int CIN, int KS, int STRIDE, int NTW>
__global__ __launch_bounds__(256) void conv2d_small_kernel(const float* __restrict__ in, const float4* __restrict__ wpk, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float* __restrict__ out, int N, int Hin, int Win,
                                                           int Ho, int Wo, int cout, int relu, int strips_y, int segs_x, int rows)
{
    // NTW = 16-channel output tiles per wave: every operand fetched serves NTW tiles (the 3x3 stride-2 layer re-reads its taps per
    // tile otherwise and is bound by the L1 / texture-address path)
    constexpr int Q = CIN / 16, TAPS = KS * KS, PAD = KS / 2;
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
    const int ngroups = cout / (16 * NTW);
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);          // wave id -> (n, strip, segment, channel group)
    const long long nwaves = (long long)N * strips_y * segs_x * ngroups;
    if (wid >= nwaves) return;
    const int ng = (int)(wid % ngroups);
    const int seg = (int)((wid / ngroups) % segs_x);
    const int strip = (int)((wid / ((long long)ngroups * segs_x)) % strips_y);
    const int n = (int)(wid / ((long long)ngroups * segs_x * strips_y));

    float4 wr[NTW][TAPS][Q];                                  // this lane's weights: [tile][tap][16-channel group] x 4 k-steps
    float4 sc[NTW], sh[NTW];
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
        const int nt = ng * NTW + u;
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int q = 0; q < Q; ++q) wr[u][t][q] = wpk[((nt * TAPS + t) * Q + q) * 64 + lane];
        sc[u] = reinterpret_cast<const float4*>(scale + 16 * nt)[g];
        sh[u] = reinterpret_cast<const float4*>(shift + 16 * nt)[g];
    }

    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (size_t)n * Hin * Win * CIN), 0, (int)((size_t)Hin * Win * CIN * 4), 0x00020000);
    const int x = seg * 16 + i;
    unsigned coloff[KS];                                      // byte offset of the source column of tap kx (OOB: beyond the buffer)
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) {
        const int xx = x * STRIDE + kx - PAD;
        coloff[kx] = (x < Wo && (unsigned)xx < (unsigned)Win) ? (unsigned)(xx * CIN + 4 * g) * 4u : 0xFFFFFF00u;
    }
    for (int r = 0; r < rows; ++r) {
        const int y = strip * rows + r;
        if (y >= Ho) break;                                   // wave-uniform
        f32x4_t acc[NTW][2];                                  // two chains per tile: no MFMA waits for its own predecessor
#pragma unroll
        for (int u = 0; u < NTW; ++u) { acc[u][0] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; acc[u][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            const int yy = y * STRIDE + ky - PAD;
            if ((unsigned)yy >= (unsigned)Hin) continue;      // wave-uniform: zero padding rows
            const int rowoff = yy * Win * CIN * 4;
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(rs_in, coloff[kx], rowoff + q * 64, 0);
                    float4 a;
                    __builtin_memcpy(&a, &raw, 16);
#pragma unroll
                    for (int u = 0; u < NTW; ++u) {
                        const float4 wq = wr[u][ky * KS + kx][q];
                        acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.x, a.x, acc[u][0], 0, 0, 0);
                        acc[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.y, a.y, acc[u][1], 0, 0, 0);
                        acc[u][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.z, a.z, acc[u][0], 0, 0, 0);
                        acc[u][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq.w, a.w, acc[u][1], 0, 0, 0);
                    }
                }
            }
        }
        if (x < Wo) {
#pragma unroll
            for (int u = 0; u < NTW; ++u) {
                float4 o;
                o.x = (acc[u][0][0] + acc[u][1][0]) * sc[u].x + sh[u].x;
                o.y = (acc[u][0][1] + acc[u][1][1]) * sc[u].y + sh[u].y;
                o.z = (acc[u][0][2] + acc[u][1][2]) * sc[u].z + sh[u].z;
                o.w = (acc[u][0][3] + acc[u][1][3]) * sc[u].w + sh[u].w;
                if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
                *reinterpret_cast<float4*>(out + (((size_t)n * Ho + y) * Wo + x) * cout + 16 * (ng * NTW + u) + 4 * g) = o;
            }
        }
    }
}

// ---- [N][3][HW] image planes in 0..255 -> [N][HW][3] records in -1..1 ----
__global__ __launch_bounds__(256) void normalise_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, long long HW, long long total)
{
#pragma clang fp contract(off)
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;         // one thread per pixel
    if (e >= total) return;
    const long long n = e / HW, p = e - n * HW;
    const float* src = in + n * 3 * HW + p;
    float* dst = out + e * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float q = src[c * HW] / 255.0f;                               // three roundings, as the reference's three ATen ops
        const float d = 2.0f * q;
        dst[c] = d - 1.0f;
    }
}

// ---- 3x3 / stride 2 / padding 1, 3 -> 32 channels, folded BatchNorm, ReLU; NHWC in [N][H][W][3] -> out [N][Ho][Wo][32] ----
__global__ __launch_bounds__(256) void stem3x3s2_nhwc_kernel(const float* __restrict__ in, const float* __restrict__ w, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float* __restrict__ out, int N, int H, int W, int Ho,
                                                             int Wo)
{
    __shared__ __attribute__((aligned(16))) float ws[27 * 32];          // [tap = (ky, kx, ci)][co]
    for (int e = threadIdx.x; e < 27 * 32; e += 256) {                  // w is [32][3][3][3] (Conv2d layout)
        const int t = e >> 5, co = e & 31;
        const int ky = t / 9, kx = (t / 3) % 3, ci = t % 3;
        ws[e] = w[((co * 3 + ci) * 3 + ky) * 3 + kx];
    }
    __syncthreads();
    const long long total = (long long)N * Ho * Wo;
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long pc = pix < total ? pix : total - 1;     // lanes past the end recompute the last pixel and store nothing
    const int xo = (int)(pc % Wo), yo = (int)((pc / Wo) % Ho);
    const long long n = pc / ((long long)Wo * Ho);
    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.f;
#pragma clang loop unroll(disable)
    for (int t = 0; t < 9; ++t) {                          // rolled: fully unrolled the 216 weight reads are hoisted (356 spills)
        const int ky = t / 3, kx = t - 3 * ky;
        const int y = 2 * yo - 1 + ky, x = 2 * xo - 1 + kx;
        const bool ok = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        const float* src = in + ((n * H + (ok ? y : 0)) * W + (ok ? x : 0)) * 3;
        const float v3[3] = {ok ? src[0] : 0.f, ok ? src[1] : 0.f, ok ? src[2] : 0.f};
#pragma unroll
        for (int ci = 0; ci < 3; ++ci) {
            const float v = v3[ci];
            const float4* wt = reinterpret_cast<const float4*>(ws + (t * 3 + ci) * 32);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float4 wq = wt[q];
                acc[4 * q + 0] = fmaf(v, wq.x, acc[4 * q + 0]);
                acc[4 * q + 1] = fmaf(v, wq.y, acc[4 * q + 1]);
                acc[4 * q + 2] = fmaf(v, wq.z, acc[4 * q + 2]);
                acc[4 * q + 3] = fmaf(v, wq.w, acc[4 * q + 3]);
            }
        }
    }
    // BN + ReLU, then through LDS so that a store instruction writes whole 128-byte records (lane l: chunk l % 8 of pixel l / 8)
    __shared__ __attribute__((aligned(16))) float4 stage[4][64][9];            // [wave][pixel lane][8 chunks + pad]
    const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 sc = reinterpret_cast<const float4*>(scale)[q], sh = reinterpret_cast<const float4*>(shift)[q];
        float4 r;
        r.x = fmaxf(acc[4 * q + 0] * sc.x + sh.x, 0.f);
        r.y = fmaxf(acc[4 * q + 1] * sc.y + sh.y, 0.f);
        r.z = fmaxf(acc[4 * q + 2] * sc.z + sh.z, 0.f);
        r.w = fmaxf(acc[4 * q + 3] * sc.w + sh.w, 0.f);
        stage[wv][ln][q] = r;
    }
    __syncthreads();
    const long long wave_pix0 = pix - ln;                  // first pixel of this wave
    float4* o = reinterpret_cast<float4*>(out);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int e = it * 64 + ln, pl = e >> 3, ch = e & 7;
        if (wave_pix0 + pl < total) o[(wave_pix0 + pl) * 8 + ch] = stage[wv][pl][ch];
    }
}

}  // namespace

extern "C" int estd_conv2d_k3_to16_nhwc(const float* in, const float* w_packed, const float* scale, const float* shift, float* out, int N, int H,
                                        int W, int cin, int upsample, estd_stream_t s)
{
    if (!in || !w_packed || !scale || !shift || !out || N <= 0 || H <= 0 || W <= 0) return ESTD_ERR_ARG;
    if ((cin != 16 && cin != 32) || (upsample != 0 && upsample != 1)) return ESTD_ERR_UNSUPPORTED;
    if (upsample && ((H | W) & 1)) return ESTD_ERR_ARG;
    const long long hin = upsample ? H / 2 : H, win = upsample ? W / 2 : W;
    if (hin * win * cin * 4 >= 0x7fffff00LL) return ESTD_ERR_UNSUPPORTED;       // one image through a 32-bit buffer descriptor
    const int strips_y = (H + TO16_ROWS - 1) / TO16_ROWS, segs_x = (W + 15) / 16;
    const long long waves = (long long)N * strips_y * segs_x;
    const long long blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffLL) return ESTD_ERR_ARG;
    const float4* wp = reinterpret_cast<const float4*>(w_packed);
#define ESTD_TO16(C, U) hipLaunchKernelGGL((conv2d_k3_to16_kernel<C, U>), dim3((unsigned)blocks), dim3(256), 0, estd_stream(s), in, wp, scale, shift, \
                                           out, N, H, W, strips_y, segs_x)
    if (cin == 16) { if (upsample) ESTD_TO16(16, true); else ESTD_TO16(16, false); }
    else { if (upsample) ESTD_TO16(32, true); else ESTD_TO16(32, false); }
#undef ESTD_TO16
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_conv2d_small_nhwc(const float* in, const float* w_packed, const float* scale, const float* shift, float* out, int N, int Hin,
                                      int Win, int cin, int cout, int ksize, int stride, int relu, estd_stream_t s)
{
    if (!in || !w_packed || !scale || !shift || !out || N <= 0 || Hin <= 0 || Win <= 0) return ESTD_ERR_ARG;
    if (cout <= 0 || (cout & 15) || (ksize != 1 && ksize != 3) || (stride != 1 && stride != 2)) return ESTD_ERR_ARG;
    if ((long long)Hin * Win * cin * 4 >= 0x7fffff00LL) return ESTD_ERR_UNSUPPORTED;
    const int pad = ksize / 2;
    const int Ho = (Hin + 2 * pad - ksize) / stride + 1, Wo = (Win + 2 * pad - ksize) / stride + 1;
    // channel tiles per wave: 2 where the register budget allows and the layer has an even tile count (3x3: 144 weight registers)
    const int ntw = ((cout & 31) == 0 && cin <= 64) ? 2 : 1;
    // rows of 16 pixels per wave: 8 amortise the weight fetch; fewer when the map is small, so that every SIMD has several waves to
    // cover the load latency with (these kernels are latency / bandwidth bound)
    const int segs_x = (Wo + 15) / 16, ngroups = cout / (16 * ntw);
    int rows = TO16_ROWS;
    while (rows > 1 && (long long)N * ((Ho + rows - 1) / rows) * segs_x * ngroups < 8192) rows >>= 1;
    const int strips_y = (Ho + rows - 1) / rows;
    const long long waves = (long long)N * strips_y * segs_x * ngroups;
    const long long blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffLL) return ESTD_ERR_ARG;
    const float4* wp = reinterpret_cast<const float4*>(w_packed);
#define ESTD_SMALL(C, K, S, T)                                                                                                                    \
    hipLaunchKernelGGL((conv2d_small_kernel<C, K, S, T>), dim3((unsigned)blocks), dim3(256), 0, estd_stream(s), in, wp, scale, shift, out, N, Hin, \
                       Win, Ho, Wo, cout, relu, strips_y, segs_x, rows)
    if (ksize == 3 && stride == 2 && cin == 32) { if (ntw == 2) ESTD_SMALL(32, 3, 2, 2); else ESTD_SMALL(32, 3, 2, 1); }
    else if (ksize == 1 && stride == 2 && cin == 32) { if (ntw == 2) ESTD_SMALL(32, 1, 2, 2); else ESTD_SMALL(32, 1, 2, 1); }
    else if (ksize == 1 && stride == 1 && cin == 32) { if (ntw == 2) ESTD_SMALL(32, 1, 1, 2); else ESTD_SMALL(32, 1, 1, 1); }
    else if (ksize == 1 && stride == 1 && cin == 64) { if (ntw == 2) ESTD_SMALL(64, 1, 1, 2); else ESTD_SMALL(64, 1, 1, 1); }
    else if (ksize == 1 && stride == 1 && cin == 128) ESTD_SMALL(128, 1, 1, 1);
    else return ESTD_ERR_UNSUPPORTED;
#undef ESTD_SMALL
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_normalise_nhwc(const float* in, float* out, int N, int64_t HW, estd_stream_t s)
{
    if (!in || !out || N <= 0 || HW <= 0) return ESTD_ERR_ARG;
    const long long total = (long long)N * HW;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return ESTD_ERR_ARG;
    hipLaunchKernelGGL(normalise_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, estd_stream(s), in, out, (long long)HW, total);
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_stem3x3s2_nhwc(const float* in, const float* w, const float* scale, const float* shift, float* out, int N, int H, int W,
                                   estd_stream_t s)
{
    if (!in || !w || !scale || !shift || !out || N <= 0 || H <= 0 || W <= 0) return ESTD_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)N * Ho * Wo;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return ESTD_ERR_ARG;
    hipLaunchKernelGGL(stem3x3s2_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, estd_stream(s), in, w, scale, shift, out, N, H, W, Ho, Wo);
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_planes_cat_nhwc(const float* a, int Ca, const float* b, int Cb, int relu_b, float* out, int N, int64_t HW, estd_stream_t s)
{
    if (!a || !b || !out || Ca <= 0 || Cb <= 0 || N <= 0 || HW <= 0) return ESTD_ERR_ARG;
    const int C = Ca + Cb;
    const int pix = (size_t)C * 65 * sizeof(float) <= 64 * 1024 ? 64 : 32;                   // LDS tile [C][pix + 1] within 64 KB
    const size_t lds = (size_t)C * (pix + 1) * sizeof(float);
    if (lds > 64 * 1024) return ESTD_ERR_UNSUPPORTED;                                        // <= 496 channels
    const long long blocks = (long long)N * ((HW + pix - 1) / pix);
    if (blocks > 0x7fffffffLL) return ESTD_ERR_ARG;
    if (pix == 64)
        hipLaunchKernelGGL(planes_cat_nhwc_kernel<64>, dim3((unsigned)blocks), dim3(256), lds, estd_stream(s), a, Ca, b, Cb, relu_b, out, (long long)HW);
    else
        hipLaunchKernelGGL(planes_cat_nhwc_kernel<32>, dim3((unsigned)blocks), dim3(256), lds, estd_stream(s), a, Ca, b, Cb, relu_b, out, (long long)HW);
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_nhwc_to_planes(const float* in, int C, float* out, int N, int64_t HW, estd_stream_t s)
{
    if (!in || !out || C <= 0 || N <= 0 || HW <= 0) return ESTD_ERR_ARG;
    const int pix = (size_t)C * 65 * sizeof(float) <= 64 * 1024 ? 64 : 32;
    const size_t lds = (size_t)C * (pix + 1) * sizeof(float);
    if (lds > 64 * 1024) return ESTD_ERR_UNSUPPORTED;                                        // <= 496 channels
    const long long blocks = (long long)N * ((HW + pix - 1) / pix);
    if (blocks > 0x7fffffffLL) return ESTD_ERR_ARG;
    if (pix == 64) hipLaunchKernelGGL(nhwc_to_planes_kernel<64>, dim3((unsigned)blocks), dim3(256), lds, estd_stream(s), in, C, out, (long long)HW);
    else hipLaunchKernelGGL(nhwc_to_planes_kernel<32>, dim3((unsigned)blocks), dim3(256), lds, estd_stream(s), in, C, out, (long long)HW);
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_upsample2_cat_nhwc(const float* x, int Cx, const float* skip, int Cs, float* out, int N, int H, int W, estd_stream_t s)
{
    if (!x || !skip || !out || Cx <= 0 || Cs <= 0 || (Cx & 3) || (Cs & 3) || N <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1)) return ESTD_ERR_ARG;
    const long long total = (long long)N * H * W * ((Cx + Cs) / 4);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return ESTD_ERR_ARG;
    hipLaunchKernelGGL(upsample2_cat_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, estd_stream(s), reinterpret_cast<const float4*>(x), Cx / 4,
                       reinterpret_cast<const float4*>(skip), Cs / 4, reinterpret_cast<float4*>(out), N, H, W);
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_disp_head_nhwc(const float* in, const float* w, const float* bias, float depth_max, float* out, int N, int H, int W, int C,
                                   int upscale, estd_stream_t s)
{
    if (!in || !w || !bias || !out || N <= 0 || H <= 0 || W <= 0 || (upscale != 1 && upscale != 2)) return ESTD_ERR_ARG;
    if (C != 16 && C != 32) return ESTD_ERR_UNSUPPORTED;
    const long long total = (long long)N * H * W;
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return ESTD_ERR_ARG;
    if (C == 16)
        hipLaunchKernelGGL(disp_head_nhwc_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, estd_stream(s), in, w, bias, depth_max, out, N, H, W, upscale);
    else
        hipLaunchKernelGGL(disp_head_nhwc_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, estd_stream(s), in, w, bias, depth_max, out, N, H, W, upscale);
    return ESTD_LAUNCH_CHECK();
}
