// conv2d_taps.hip -- the layers of the semantic branch's ResNet trunk that are neither 1x1 nor "3x3 / stride 1 on a large map"
// (hybrid_models/resnet_encoder.py:40-51 over torchvision's ResNet: conv1 7x7 / stride 2 + bn1 + relu, maxpool 3x3 / stride 2,
// the stride-2 3x3 convolution of layer2..4's first block; the long-K 3x3 of the 2D decoder on the 15x20 map,
// hybrid_depth_decoder.py:17-30), and the k x k / stride-k average pooling of the PSM extractor's SPP branches
// (networks/psm_submodule.py:56-70).  SURVEY.md §8(f) rank 3: with these no library convolution / pooling kernel is left in either
// 2D branch.  NHWC in, NHWC out, folded BatchNorm + residual + ReLU in the convolutions' epilogues, fp32 MFMA (16x16x4).
//
//   estd_conv2d_taps_nhwc    k x k convolution (k odd, <= 5), stride 1 | 2, any zero padding: an implicit GEMM over (tap, input
//                            channel) in the operand scheme of csrc/conv1x1.hip -- operands straight from L1 / L2 in MFMA layout, a
//                            wave owns a (16 TM pixels) x (16 TN channels) block, ring of prefetched 16-channel chunks; the tap of a
//                            chunk moves the pixel address by a wave-uniform delta and selects one bit of a per-lane validity mask
//                            (zero padding = a buffer load beyond the descriptor's range, which returns 0).  SK = 4: the four waves
//                            of a workgroup split the (tap, channel) range and wave 0 adds the partial sums in a fixed order.
//   estd_stem7x7s2_nhwc      3 -> 64, 7 x 7, stride 2, padding 3 + BN + ReLU: K = 7 rows x 8 pixels x 3 channels (the 8th pixel is a
//                            zero tap), lane group g of an MFMA k-step owns pixels 2g, 2g + 1 of the window: two 12-byte loads per
//                            lane and row feed 6 k-steps x 4 channel tiles; the 168 weights of a lane stay in registers over a strip.
//   estd_maxpool3x3s2_nhwc   MaxPool2d(3, 2, 1) (padding = -inf, i.e. ignored).
//   estd_avgpool_nhwc        k x k / stride k average (floor output size), window sum in row-major order then one division, as ATen.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "estd_hip.h"
#include "estd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
typedef unsigned int u32x3 __attribute__((__vector_size__(12)));
constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;

__device__ __forceinline__ float4 as_float4(u32x4 v) { float4 f; __builtin_memcpy(&f, &v, 16); return f; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}

template <int TM, int TN, int PF, int SK>
__global__ __launch_bounds__(256) void conv2d_taps_kernel(const estd_conv2d_taps_desc p, int Ho, int Wo, int tiles_m, int tiles_n)
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
    const int cin = p.cin, cout = p.cout, ks = p.ksize;

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.in, (size_t)p.N * p.H * p.W * cin * 4);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (size_t)ks * ks * cout * cin * 4);
    int xbase[TM];                                              // byte offset of the window's first pixel (negative in the padding: used with a valid tap only)
    unsigned xmask[TM], woff[TN];                               // bit (ky * ks + kx): that tap reads a pixel inside the map
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int m = m0 + 16 * t + i;
        const int x = m % Wo, r = m / Wo;
        const int y = r % Ho, n = r / Ho;
        const int y0 = y * p.stride - p.pad, x0 = x * p.stride - p.pad;
        xbase[t] = (((n * p.H + y0) * p.W + x0) * cin + 4 * g) * 4;
        unsigned mk = 0;
        if (m < Mtot)
            for (int ky = 0; ky < ks; ++ky)
                for (int kx = 0; kx < ks; ++kx)
                    if ((unsigned)(y0 + ky) < (unsigned)p.H && (unsigned)(x0 + kx) < (unsigned)p.W) mk |= 1u << (ky * ks + kx);
        xmask[t] = mk;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) woff[t] = (unsigned)((n0 + 16 * t + i) * cin + 4 * g) * 4u;       // cout is a multiple of 16 TN

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int cpt = cin >> 4;                                   // 16-channel chunks per tap
    const int nchunks = ks * ks * cpt / SK;                     // this wave's share of the (tap, chunk) range: kbase .. kbase + nchunks
    const int kbase = SK > 1 ? wave * nchunks : 0;
    // load cursor (wave-uniform, scalar registers): the next chunk to request
    int l_tap = kbase / cpt, l_kc = kbase - l_tap * cpt;
    int l_ky = l_tap / ks, l_kx = l_tap - l_ky * ks;

    float4 xq[PF][TM], wq[PF][TN];
    auto load_next = [&](float4 (&xd)[TM], float4 (&wd)[TN]) {
        const int xdelta = ((l_ky * p.W + l_kx) * cin + l_kc * 16) * 4;
        const int wdelta = (l_tap * cout * cin + l_kc * 16) * 4;
#pragma unroll
        for (int t = 0; t < TN; ++t) wd[t] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, woff[t], wdelta, 0));
#pragma unroll
        for (int t = 0; t < TM; ++t) {
            const unsigned off = ((xmask[t] >> l_tap) & 1u) ? (unsigned)(xbase[t] + xdelta) : OOB_OFFSET;
            xd[t] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_x, off, 0, 0));
        }
        if (++l_kc == cpt) {
            l_kc = 0; ++l_tap;
            if (++l_kx == ks) { l_kx = 0; ++l_ky; }
        }
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
#pragma unroll
    for (int j = 0; j < PF - 1; ++j)
        if (j < nchunks) load_next(xq[j], wq[j]);
    for (int c = 0; c < nchunks; c += PF) {                     // PF chunks per trip: the ring slots are compile-time constants
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int cc = c + j;                                // (wave-uniform conditions)
            if (cc + PF - 1 < nchunks) load_next(xq[(j + PF - 1) % PF], wq[(j + PF - 1) % PF]);
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
int launch_taps(const estd_conv2d_taps_desc& d, int Ho, int Wo, hipStream_t stream)
{
    const long long Mtot = (long long)d.N * Ho * Wo;
    const int tiles_m = (int)((Mtot + 16 * TM - 1) / (16 * TM)), tiles_n = d.cout / (16 * TN);
    const long long wts = (long long)tiles_m * tiles_n;
    const unsigned grid = SK > 1 ? (unsigned)wts : (unsigned)((wts + 3) / 4);
    hipLaunchKernelGGL((conv2d_taps_kernel<TM, TN, PF, SK>), dim3(grid), dim3(256), 0, stream, d, Ho, Wo, tiles_m, tiles_n);
    return ESTD_LAUNCH_CHECK();
}

// ---- 7x7 / stride 2 / padding 3, 3 -> 64 channels, folded BatchNorm, ReLU; in [N][H][W][3] -> out [N][Ho][Wo][64] ----
// A wave owns 16 neighbouring output pixels of a row over STEM_ROWS rows; lane (g, i): output pixel i, k slots 6g .. 6g+5 of a
// window row = channels of window pixels 2g and 2g + 1 (input columns 2 x - 3 + 2g, + 1; pixel 7 of group 3 is the zero tap).
constexpr int STEM_ROWS = 8;
__global__ __launch_bounds__(256) void stem7x7s2_nhwc_kernel(const float* __restrict__ in, const float* __restrict__ wpk, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float* __restrict__ out, int N, int H, int W, int Ho,
                                                             int Wo, int strips_y, int segs_x)
{
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long nwaves = (long long)N * strips_y * segs_x;
    if (wid >= nwaves) return;
    const int seg = (int)(wid % segs_x);
    const int strip = (int)((wid / segs_x) % strips_y);
    const int n = (int)(wid / ((long long)segs_x * strips_y));

    float wr[7][6][4];                                        // [window row][k-step][channel tile]: A operand of lane (k = 6g + step, channel 16 tile + i)
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int s = 0; s < 6; ++s)
#pragma unroll
            for (int u = 0; u < 4; ++u) wr[ky][s][u] = wpk[((ky * 6 + s) * 4 + u) * 64 + lane];
    float4 sc[4], sh[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        sc[u] = reinterpret_cast<const float4*>(scale + 16 * u)[g];
        sh[u] = reinterpret_cast<const float4*>(shift + 16 * u)[g];
    }
    const __amdgpu_buffer_rsrc_t rs_in =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in + (size_t)n * H * W * 3), 0, (int)((size_t)H * W * 3 * 4), 0x00020000);
    const int x = seg * 16 + i;
    const int c0 = 2 * x - 3 + 2 * g, c1 = c0 + 1;            // the two window pixels of this lane
    const unsigned off0 = (x < Wo && (unsigned)c0 < (unsigned)W) ? (unsigned)c0 * 12u : OOB_OFFSET;
    const unsigned off1 = (x < Wo && g < 3 && (unsigned)c1 < (unsigned)W) ? (unsigned)c1 * 12u : OOB_OFFSET;
    for (int r = 0; r < STEM_ROWS; ++r) {
        const int y = strip * STEM_ROWS + r;
        if (y >= Ho) break;                                   // wave-uniform
        f32x4 acc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 7; ++ky) {
            const int yy = 2 * y - 3 + ky;
            if ((unsigned)yy >= (unsigned)H) continue;        // wave-uniform: zero padding rows
            const int rowoff = yy * W * 12;
            const u32x3 r0 = __builtin_amdgcn_raw_buffer_load_b96(rs_in, off0, rowoff, 0);
            const u32x3 r1 = __builtin_amdgcn_raw_buffer_load_b96(rs_in, off1, rowoff, 0);
            float a[6];
            __builtin_memcpy(&a[0], &r0, 12);
            __builtin_memcpy(&a[3], &r1, 12);
#pragma unroll
            for (int s = 0; s < 6; ++s)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[ky][s][u], a[s], acc[u], 0, 0, 0);
        }
        if (x < Wo) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 o;
                o.x = fmaxf(fmaf(acc[u][0], sc[u].x, sh[u].x), 0.f);
                o.y = fmaxf(fmaf(acc[u][1], sc[u].y, sh[u].y), 0.f);
                o.z = fmaxf(fmaf(acc[u][2], sc[u].z, sh[u].z), 0.f);
                o.w = fmaxf(fmaf(acc[u][3], sc[u].w, sh[u].w), 0.f);
                *reinterpret_cast<float4*>(out + (((size_t)n * Ho + y) * Wo + x) * 64 + 16 * u + 4 * g) = o;
            }
        }
    }
}

// ---- MaxPool2d(3, 2, 1) of an NHWC map; one thread per (output pixel, 4-channel group) ----
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc_kernel(const float4* __restrict__ in, float4* __restrict__ out, int N, int H, int W, int C4,
                                                                int Ho, int Wo, long long total)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C4);
    const long long pix = e / C4;
    const int x = (int)(pix % Wo), y = (int)((pix / Wo) % Ho);
    const long long n = pix / ((long long)Wo * Ho);
    const float ninf = -__builtin_inff();
    float4 m = make_float4(ninf, ninf, ninf, ninf);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = 2 * y - 1 + ky;
        if ((unsigned)yy >= (unsigned)H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = 2 * x - 1 + kx;
            if ((unsigned)xx >= (unsigned)W) continue;
            const float4 v = in[((n * H + yy) * W + xx) * C4 + c];
            // ATen's comparison (val > max || isnan(val)): a NaN in the window is the result
            m.x = (v.x > m.x || v.x != v.x) ? v.x : m.x;
            m.y = (v.y > m.y || v.y != v.y) ? v.y : m.y;
            m.z = (v.z > m.z || v.z != v.z) ? v.z : m.z;
            m.w = (v.w > m.w || v.w != v.w) ? v.w : m.w;
        }
    }
    out[e] = m;
}

// ---- AvgPool2d(k, k) of an NHWC map (floor output size); one thread per (output pixel, 4-channel group) ----
__global__ __launch_bounds__(256) void avgpool_nhwc_kernel(const float4* __restrict__ in, float4* __restrict__ out, int H, int W, int C4, int k,
                                                           int Ho, int Wo, long long total)
{
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int c = (int)(e % C4);
    const long long pix = e / C4;
    const int x = (int)(pix % Wo), y = (int)((pix / Wo) % Ho);
    const long long n = pix / ((long long)Wo * Ho);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ky = 0; ky < k; ++ky) {
        const float4* row = in + ((n * H + (long long)y * k + ky) * W + (long long)x * k) * C4 + c;
        for (int kx = 0; kx < k; ++kx) {
            const float4 v = row[(long long)kx * C4];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    const float div = (float)(k * k);
    out[e] = make_float4(s.x / div, s.y / div, s.z / div, s.w / div);
}

}  // namespace

extern "C" int estd_conv2d_taps_nhwc(const estd_conv2d_taps_desc* dp, estd_stream_t s)
{
    if (!dp) return ESTD_ERR_ARG;
    const estd_conv2d_taps_desc& d = *dp;
    if (d.N <= 0 || d.H <= 0 || d.W <= 0 || !d.in || !d.w || !d.out || d.pad < 0) return ESTD_ERR_ARG;
    if (d.stride != 1 && d.stride != 2) return ESTD_ERR_UNSUPPORTED;
    if (d.ksize != 1 && d.ksize != 3 && d.ksize != 5) return ESTD_ERR_UNSUPPORTED;
    if (d.cin < 16 || (d.cin & 15) || d.cout < 32 || (d.cout & 31)) return ESTD_ERR_UNSUPPORTED;
    if (d.H + 2 * d.pad < d.ksize || d.W + 2 * d.pad < d.ksize) return ESTD_ERR_ARG;
    const int Ho = (d.H + 2 * d.pad - d.ksize) / d.stride + 1, Wo = (d.W + 2 * d.pad - d.ksize) / d.stride + 1;
    const long long Mtot = (long long)d.N * Ho * Wo;
    const int taps = d.ksize * d.ksize;
    // 32-bit byte offsets inside the input map (plus one padded window), the output map and the weight array
    if (((long long)d.N * d.H + d.ksize) * d.W * d.cin * 4 >= 0x7fffff00LL || Mtot * d.cout * 4 >= 0x7fffff00LL ||
        (long long)taps * d.cin * d.cout * 4 >= 0x7fffff00LL)
        return ESTD_ERR_UNSUPPORTED;
    hipStream_t stream = estd_stream(s);
    // block per wave: as csrc/conv1x1.hip (the largest that still gives the device about one wave per SIMD); ESTD_CTAPS_CFG = 100 TM + 10 TN + SK forces one
    static const int cfg_env = [] { const char* e = getenv("ESTD_CTAPS_CFG"); return e ? atoi(e) : 0; }();
    const long long want = (long long)estd_device_cus() * 7 / 2;
    auto tiles = [&](int tm, int tn) { return ((Mtot + 16 * tm - 1) / (16 * tm)) * (d.cout / (16 * tn)); };
    const bool sk4 = (d.cin & 63) == 0 && (long long)taps * d.cin >= 256;
    const bool c64 = (d.cout & 63) == 0;
    int cfg = cfg_env;
    if (cfg == 0) {
        if (c64 && tiles(4, 4) >= want) cfg = 441;
        else if (tiles(4, 2) >= want) cfg = 421;
        else if (sk4 && c64 && tiles(4, 4) >= want / 4) cfg = 444;
        else if (tiles(2, 2) >= want) cfg = 221;
        else if (sk4 && c64 && tiles(2, 4) >= want / 4) cfg = 244;
        else if (sk4 && tiles(2, 2) >= want / 4) cfg = 224;
        else if (sk4) cfg = 124;
        else cfg = 121;
    }
    if ((cfg % 10) == 4 && (d.cin & 63)) cfg = cfg - 3;                                 // SK needs whole chunks per range: cin % 64 == 0
    if ((cfg / 10) % 10 == 4 && !c64) cfg -= 20;                                        // TN = 4 needs cout % 64 == 0
    switch (cfg) {
    case 441: return launch_taps<4, 4, 2, 1>(d, Ho, Wo, stream);
    case 421: return launch_taps<4, 2, 3, 1>(d, Ho, Wo, stream);
    case 241: return launch_taps<2, 4, 3, 1>(d, Ho, Wo, stream);
    case 221: return launch_taps<2, 2, 4, 1>(d, Ho, Wo, stream);
    case 444: return launch_taps<4, 4, 2, 4>(d, Ho, Wo, stream);
    case 424: return launch_taps<4, 2, 3, 4>(d, Ho, Wo, stream);
    case 244: return launch_taps<2, 4, 3, 4>(d, Ho, Wo, stream);
    case 224: return launch_taps<2, 2, 4, 4>(d, Ho, Wo, stream);
    case 124: return launch_taps<1, 2, 4, 4>(d, Ho, Wo, stream);
    default: return launch_taps<1, 2, 8, 1>(d, Ho, Wo, stream);
    }
}

extern "C" int estd_stem7x7s2_nhwc(const float* in, const float* w_packed, const float* scale, const float* shift, float* out, int N, int H, int W,
                                   estd_stream_t s)
{
    if (!in || !w_packed || !scale || !shift || !out || N <= 0 || H <= 0 || W <= 0) return ESTD_ERR_ARG;
    if ((long long)H * W * 12 >= 0x7fffff00LL) return ESTD_ERR_UNSUPPORTED;           // one image through a 32-bit buffer descriptor
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;                             // kernel 7, stride 2, padding 3
    const int strips_y = (Ho + STEM_ROWS - 1) / STEM_ROWS, segs_x = (Wo + 15) / 16;
    const long long waves = (long long)N * strips_y * segs_x;
    const long long blocks = (waves + 3) / 4;
    if (blocks > 0x7fffffffLL) return ESTD_ERR_ARG;
    hipLaunchKernelGGL(stem7x7s2_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, estd_stream(s), in, w_packed, scale, shift, out, N, H, W, Ho, Wo,
                       strips_y, segs_x);
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_maxpool3x3s2_nhwc(const float* in, float* out, int N, int H, int W, int C, estd_stream_t s)
{
    if (!in || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return ESTD_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long long total = (long long)N * Ho * Wo * (C / 4);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return ESTD_ERR_ARG;
    hipLaunchKernelGGL(maxpool3x3s2_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, estd_stream(s), reinterpret_cast<const float4*>(in),
                       reinterpret_cast<float4*>(out), N, H, W, C / 4, Ho, Wo, total);
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_avgpool_nhwc(const float* in, float* out, int N, int H, int W, int C, int k, estd_stream_t s)
{
    if (!in || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || k <= 0 || k > H || k > W) return ESTD_ERR_ARG;
    const int Ho = H / k, Wo = W / k;
    const long long total = (long long)N * Ho * Wo * (C / 4);
    const long long blocks = (total + 255) / 256;
    if (blocks > 0x7fffffffLL) return ESTD_ERR_ARG;
    hipLaunchKernelGGL(avgpool_nhwc_kernel, dim3((unsigned)blocks), dim3(256), 0, estd_stream(s), reinterpret_cast<const float4*>(in),
                       reinterpret_cast<float4*>(out), H, W, C / 4, k, Ho, Wo, total);
    return ESTD_LAUNCH_CHECK();
}
