// conv2d_mfma.hip -- 3x3 (stride 1, dilation 1 or 2) 2D convolution on NHWC maps as an implicit GEMM on gfx950 fp32
// MFMA, with folded BatchNorm / ReLU / residual epilogue.
//
// the PSMNet matching-feature extractor (networks/psm_submodule.py:40-116) is 20 % of a
// forward pass and sits directly in front of the plane sweep; its 3x3 convolutions (Conv2d bias=False + BatchNorm2d
// [+ReLU], networks/layers_op.py:10-27, residual add of BasicBlock psm_submodule.py:26-37) run here instead of on
// MIOpen's Winograd kernels.  Same building blocks as conv3d_mfma.hip:
//   * v_mfma_f32_16x16x4_f32 (exact fp32), M = 16 pixels along W, N = 16 output channels, K = 4 input channels;
//   * 256-thread persistent workgroups, 8x16-pixel output tiles, 2 rows per wave;
//   * the input is consumed in 32-channel chunks: one chunk of the haloed brick ((8+2d)x(16+2d) pixels x 32 ch) per LDS
//     slot, two slots, ONE LDS-only barrier per chunk; the next chunk (possibly of the next tile) is fetched into
//     registers during the MFMAs of the current one;
//   * XOR-swizzled 128-byte LDS records, permuted K order, packed weight fragments, buffer-descriptor loads/stores;
//   * output channels of one launch group: 16*NT (NT = 2 or 4), channel = NT*col + tile so a lane owns NT adjacent
//     channels (8- or 16-byte stores).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "estd_hip.h"
#include "estd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
typedef unsigned int u32x2 __attribute__((__vector_size__(8)));
constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;

constexpr int TH = 8, TW = 16, MT = 2;

__device__ __forceinline__ float4 as_float4(u32x4 v) { float4 f; __builtin_memcpy(&f, &v, 16); return f; }
__device__ __forceinline__ u32x4 as_u32x4(float4 f) { u32x4 v; __builtin_memcpy(&v, &f, 16); return v; }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, size_t elems)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(elems * 4), 0x00020000);
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int lds_chunk_off(int v, int c) { return v * 128 + ((c ^ ((v >> 1) & 7)) << 4); }

template <int NT, int DIL>
__global__ __launch_bounds__(256, 2) void conv2d_k3_kernel(const estd_conv2d_desc p, int tiles_w, int tiles_h, int total_items)
{
    constexpr int IN_H = TH + 2 * DIL, IN_W = TW + 2 * DIL;
    constexpr int NVOX = IN_H * IN_W;                 // 180 (dil 1) / 240 (dil 2)
    constexpr int NEL = NVOX * 8;                     // 16-byte chunks per 32-channel brick
    constexpr int SIT = (NEL + 255) / 256;            // 6 / 8 loads per thread per brick
    constexpr int SLOT_BYTES = NVOX * 128;
    constexpr int QN = 2 * NT;                        // weight quads per lane per tap
    static_assert(SIT <= 9, "prefetch must fit in the 9 taps");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    // MFMA row <-> pixel of a tile row (csrc/conv3d_wino.hip): rows {0-3,12-15} = even pixels, rows {4-11} = odd pixels, matched to
    // the ds_read_b128 lane groups -- conflict-free A reads for every tap parity
    const int pi = i < 4 ? 2 * i : i < 12 ? 2 * i - 7 : 2 * i - 16;
    const int px0 = g == 0 ? 0 : g == 1 ? 1 : g == 2 ? 9 : 8;             // pixel of D row 4g + r = px0 + 2r
    const int H = p.H, W = p.W, Cin = p.cin, Cout = p.cout;
    const int nchunks = Cin >> 5;
    const int tiles_per_group = p.N * tiles_h * tiles_w;
    const int row0 = wave * MT;
    const int wlane = lane * 16;

    int u, u_end;
    {
        const int G = gridDim.x, bid = blockIdx.x;
        const int r = ((G & 7) == 0) ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;
        u = (int)((long long)total_items * r / G);
        u_end = (int)((long long)total_items * (r + 1) / G);
    }
    if (u >= u_end) return;

    const size_t img_in = (size_t)H * W * Cin, img_out = (size_t)H * W * Cout;

    // ---- decode of a work item (output-channel group, image, tile) ----
    auto decode = [&](int item, int& grp, int& n, int& th0, int& tw0) {
        grp = item / tiles_per_group;
        int t = item - grp * tiles_per_group;
        const int twi = t % tiles_w; t /= tiles_w;
        const int thi = t % tiles_h; n = t / tiles_h;
        th0 = thi * TH; tw0 = twi * TW;
    };
    // per-thread source offsets (bytes inside one image, chunk 0) of the brick of a tile; OOB -> zeros
    auto brick_offsets = [&](int th0, int tw0, unsigned (&voff)[SIT]) {
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int e = tid + it * 256;
            const int vs = e >> 3, c = e & 7;
            const int zy = vs / IN_W, zx = vs - zy * IN_W;
            const int gy = th0 - DIL + zy, gx = tw0 - DIL + zx;
            const bool ok = e < NEL && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            voff[it] = ok ? (unsigned)((gy * W + gx) * Cin + c * 4) * 4u : OOB_OFFSET;
        }
    };

    int loff[SIT];
#pragma unroll
    for (int it = 0; it < SIT; ++it) {
        const int e = tid + it * 256;
        loff[it] = e < NEL ? lds_chunk_off(e >> 3, e & 7) : -1;
    }

    // current item
    int grp, n, th0, tw0;
    decode(u, grp, n, th0, tw0);
    unsigned voff[SIT];
    brick_offsets(th0, tw0, voff);
    __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in + (size_t)n * img_in, img_in);
    float4 pf[SIT];
#pragma unroll
    for (int it = 0; it < SIT; ++it) pf[it] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[it], 0, 0));

    int k = 0;                                              // global chunk counter -> LDS slot
    while (true) {
        const size_t wgrp_elems = (size_t)nchunks * 10 * QN * 256;          // packed floats per output group
        const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w + (size_t)grp * wgrp_elems, wgrp_elems);

        f32x4 acc[MT][NT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) acc[m][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // next item (for the cross-tile prefetch in the last chunk)
        const bool has_next_item = (u + 1 < u_end);
        int ngrp = grp, nn_ = n, nth0 = th0, ntw0 = tw0;
        if (has_next_item) decode(u + 1, ngrp, nn_, nth0, ntw0);

        for (int c = 0; c < nchunks; ++c, ++k) {
            char* slot = smem + (k & 1) * SLOT_BYTES;
#pragma unroll
            for (int it = 0; it < SIT; ++it)
                if (it < SIT - 1 || loff[it] >= 0) *reinterpret_cast<float4*>(slot + loff[it]) = pf[it];
            lds_barrier();

            // what to prefetch while this chunk computes
            const bool last_chunk = (c + 1 == nchunks);
            const bool do_pf = !last_chunk || has_next_item;
            unsigned nvoff[SIT];
            int pf_soff = 0;
            if (!last_chunk) {
#pragma unroll
                for (int it = 0; it < SIT; ++it) nvoff[it] = voff[it];
                pf_soff = (c + 1) * 128;                                   // next 32-channel chunk of the same pixels
            } else if (has_next_item) {
                brick_offsets(nth0, ntw0, nvoff);
                if (nn_ != n) rs_in = make_rsrc(p.in + (size_t)nn_ * img_in, img_in);
            }

            float4 bcur[QN], bnext[QN];
            const int wbase = c * 10 * QN;                                 // quads of this chunk's first tap
#pragma unroll
            for (int q = 0; q < QN; ++q) bcur[q] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, (wbase + q) * 1024, 0));

#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap % 3;
#pragma unroll
                for (int q = 0; q < QN; ++q)
                    bnext[q] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, (wbase + (tap + 1) * QN + q) * 1024, 0));
                if (do_pf && tap < SIT)
                    pf[tap] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, nvoff[tap], pf_soff, 0));
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int vs = (row0 + m + kh * DIL) * IN_W + kw * DIL + pi;
                    const int off0 = lds_chunk_off(vs, g);
                    const float4 a0 = *reinterpret_cast<const float4*>(slot + off0);
                    const float4 a1 = *reinterpret_cast<const float4*>(slot + (off0 ^ 64));
                    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
                        for (int nn = 0; nn < NT; ++nn) {
                            const int idx = ks * NT + nn;
                            const float4 bq = bcur[idx >> 2];
                            const float b = (idx & 3) == 0 ? bq.x : (idx & 3) == 1 ? bq.y : (idx & 3) == 2 ? bq.z : bq.w;
                            acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], b, acc[m][nn], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < QN; ++q) bcur[q] = bnext[q];
                __builtin_amdgcn_sched_barrier(0);
            }
            if (last_chunk && has_next_item) {
#pragma unroll
                for (int it = 0; it < SIT; ++it) voff[it] = nvoff[it];
            }
        }

        // ---- epilogue: lane = column j (N index) x rows 4g..4g+3; its channels are grp*16*NT + NT*j .. +NT-1 ----
        {
            const int cb = grp * 16 * NT + NT * i;
            float sc[NT], sh[NT];
#pragma unroll
            for (int nn = 0; nn < NT; ++nn) { sc[nn] = p.scale[cb + nn]; sh[nn] = p.shift[cb + nn]; }
            const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(p.out + (size_t)n * img_out, img_out);
            const __amdgpu_buffer_rsrc_t rs_res = make_rsrc((p.residual ? p.residual : p.out) + (size_t)n * img_out, img_out);
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int y = th0 + row0 + m;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int x = tw0 + px0 + 2 * r;
                    const unsigned eo = (y < H && x < W) ? (unsigned)((y * W + x) * Cout + cb) * 4u : OOB_OFFSET;
                    float v[NT];
#pragma unroll
                    for (int nn = 0; nn < NT; ++nn) v[nn] = acc[m][nn][r] * sc[nn] + sh[nn];
                    if (p.relu_before_residual) {
#pragma unroll
                        for (int nn = 0; nn < NT; ++nn) v[nn] = v[nn] > 0.f ? v[nn] : 0.f;
                    }
                    if (p.residual) {
                        if (NT == 4) {
                            const float4 rr = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_res, eo, 0, 0));
                            v[0] += rr.x; v[1] += rr.y; v[2] += rr.z; v[3] += rr.w;
                        } else {
                            const u32x2 rv = __builtin_amdgcn_raw_buffer_load_b64(rs_res, eo, 0, 0);
                            float2 rr; __builtin_memcpy(&rr, &rv, 8);
                            v[0] += rr.x; v[1] += rr.y;
                        }
                    }
                    if (p.relu_after_residual) {
#pragma unroll
                        for (int nn = 0; nn < NT; ++nn) v[nn] = v[nn] > 0.f ? v[nn] : 0.f;
                    }
                    if (NT == 4) {
                        __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(make_float4(v[0], v[1], v[2], v[3])), rs_out, eo, 0, 0);
                    } else {
                        const float2 ov = make_float2(v[0], v[1]);
                        u32x2 od; __builtin_memcpy(&od, &ov, 8);
                        __builtin_amdgcn_raw_buffer_store_b64(od, rs_out, eo, 0, 0);
                    }
                }
            }
        }

        if (!has_next_item) break;
        ++u;
        grp = ngrp; n = nn_; th0 = nth0; tw0 = ntw0;
    }
}

template <int NT, int DIL>
int launch2d(const estd_conv2d_desc& d, hipStream_t stream)
{
    constexpr int IN_H = TH + 2 * DIL, IN_W = TW + 2 * DIL;
    const int tiles_w = (d.W + TW - 1) / TW, tiles_h = (d.H + TH - 1) / TH;
    const int groups = d.cout / (16 * NT);
    const int total = groups * d.N * tiles_h * tiles_w;
    const size_t lds = (size_t)2 * IN_H * IN_W * 128;
    const int slots = estd_persistent_wgs(2);
    int grid = total < slots ? total : slots;
    if (grid >= 8) grid &= ~7;
    estd_allow_dynamic_lds<conv2d_k3_kernel<NT, DIL>>((int)lds);
    hipLaunchKernelGGL((conv2d_k3_kernel<NT, DIL>), dim3(grid), dim3(256), lds, stream, d, tiles_w, tiles_h, total);
    return hipGetLastError() == hipSuccess ? ESTD_OK : ESTD_ERR_LAUNCH;
}

}  // namespace

extern "C" int estd_conv2d_k3(const estd_conv2d_desc* dp, estd_stream_t s)
{
    if (!dp) return ESTD_ERR_ARG;
    const estd_conv2d_desc& d = *dp;
    if (d.N <= 0 || d.H <= 0 || d.W <= 0 || !d.in || !d.w || !d.scale || !d.shift || !d.out) return ESTD_ERR_ARG;
    if (d.cin < 32 || (d.cin & 31) || d.cout < 32 || (d.cout & 31)) return ESTD_ERR_ARG;
    if (d.dilation != 1 && d.dilation != 2) return ESTD_ERR_UNSUPPORTED;
    if (d.group_tiles != 2 && d.group_tiles != 4) return ESTD_ERR_ARG;
    if (d.cout % (16 * d.group_tiles)) return ESTD_ERR_ARG;
    const long long widest = (long long)d.H * d.W * (d.cin > d.cout ? d.cin : d.cout) * 4;
    if (widest >= 0x7fffff00LL) return ESTD_ERR_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(s);
    if (d.group_tiles == 4 && d.dilation == 1) return launch2d<4, 1>(d, stream);
    if (d.group_tiles == 4 && d.dilation == 2) return launch2d<4, 2>(d, stream);
    if (d.group_tiles == 2 && d.dilation == 1) return launch2d<2, 1>(d, stream);
    return launch2d<2, 2>(d, stream);
}
