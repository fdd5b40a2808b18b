// conv3d_xout.hip -- ONE output channel of a 3x3x3 convolution over 32 + 1 input channels (+ folded BatchNorm + activation): the 33rd OUTPUT channel
// of `dres2` = convbnrelu_3d(33, 33) (hybrid_models/hybrid_depth_decoder.py:93-95 through networks/layers_op.py:16-39, called at :196), written as
// the scalar volume the key || value convolution reads as its 33rd input (:198-199).  Round 6: with this pass beside it, dres2's 32 main output
// channels run on the three-axis Winograd kernel's scalar-channel instance (csrc/conv3d_wino3.hip, 33 -> 32) instead of the two-axis kernel's 33 -> 33
// instance, whose 33rd channel is a VALU GEMV over the MFMA fragments (16 accumulator registers and 32 spilled ones).
//
// A convolution with ONE output channel has no N dimension for a matrix core -- but its 27 taps are one: with
//       P[u][t] = sum over the 33 input channels c of w[c][t] * x[u][c]            (a POINTWISE product: [voxels x 33] . [33 x 27])
//       y[v]    = sum over the 27 taps t of P[v + offset(t)][t]                    (a shifted sum of scalars)
// the channel contraction -- 891 of the 918 operations per voxel -- is a dense GEMM with 27 (of 32) useful columns on v_mfma_f32_16x16x4_f32
// (1/26 of a 32 -> 32 3x3x3 convolution's MFMA work), and what is left is 27 LDS reads and adds per output voxel.
//   * a workgroup (256 threads) walks a column of 16 x 16-pixel tiles through a depth segment, one INPUT plane per step: the 18 x 18 halo plane
//     (324 voxels = 21 groups of 16 MFMA columns) is multiplied straight from L1 / L2 -- lane (g, i) of a group reads the 16 bytes at channels
//     16 c + 4 g of voxel i of each 16-channel chunk (the k index a lane group multiplies at step e is channel 4 g + e on both sides, as in
//     csrc/conv1x1.hip), the scalar channel as one more k-step with three zero lanes groups; weights [2 chunks][2 tap tiles] = 4 quads + 2 floats
//     per lane live in registers for the whole launch;
//   * the accumulators (taps 4 g .. 4 g + 3 and 16 + 4 g .. of voxel i) go to LDS as P[voxel][33-float pitch] (odd pitch: the stencil reads of
//     a row of output pixels fall on 16 different banks), one barrier, then thread (y, x) adds its 9 in-plane taps for each of the three depth
//     taps: plane d contributes tap kd to output plane d + 1 - kd, so two running sums per thread carry a column and one plane is finished per step;
//   * halo voxels outside the volume read through an out-of-range buffer offset (zeros: the convolution's zero padding), planes outside it are skipped.
// HBM: the input is read once (x 1.27 for the in-plane halo, from L2), 4 bytes per voxel are written.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "estd_hip.h"
#include "estd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;
#ifndef ESTD_XOUT_TW
#define ESTD_XOUT_TW 16      // tile width: 16 (256 threads, 21 column groups over 4 waves) | 32 (512 threads, 39 groups over 8 waves; A/B: 0.170 vs 0.175 ms, within noise in dres2)
#endif
constexpr int TH = 16, TW = ESTD_XOUT_TW, IN_H = TH + 2, IN_W = TW + 2, HALO = IN_H * IN_W;       // 324 (612) halo voxels per plane
constexpr int NT = TH * TW, NWAVES = NT / 64;                                           // one thread per output pixel
constexpr int GROUPS = (HALO + 15) / 16;                                                // 21 (39) groups of 16 MFMA columns
constexpr int PITCH = 33;                                                               // floats per voxel of P (27 taps used)
constexpr int DSEG = 32;                                                                // output planes per workgroup (+ 2 halo planes)

__device__ __forceinline__ float4 as_float4(u32x4 v) { float4 f; __builtin_memcpy(&f, &v, 16); return f; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, size_t elems)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(elems * 4), 0x00020000);
}

__global__ __launch_bounds__(NT) void conv3d_xout_kernel(const estd_conv3d_desc p, int tiles_w, int tiles_h, int dsegs)
{
    extern __shared__ __attribute__((aligned(16))) float P[];     // [GROUPS * 16][PITCH]: 44 352 (82 368) bytes
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int D = p.D, H = p.H, W = p.W, HW = H * W;
    // workgroup -> (volume, tile row, tile column, depth segment): depth segments of a column are neighbours (their halo planes meet in L2)
    int b = blockIdx.x;
    const int ds = b % dsegs; b /= dsegs;
    const int twi = b % tiles_w; b /= tiles_w;
    const int thi = b % tiles_h, n = b / tiles_h;
    const int h0 = thi * TH, w0 = twi * TW, d0 = ds * DSEG, d1 = min(D, d0 + DSEG);

    const size_t vol = (size_t)D * HW;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in_main + (size_t)n * vol * p.in_stride, vol * p.in_stride);
    const __amdgpu_buffer_rsrc_t rs_ex = make_rsrc(p.in_extra + (size_t)n * vol, vol);
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(p.out_extra + (size_t)n * vol, vol);

    // weights of this lane: [chunk][tap tile] quads (taps 16 t + i, channels 16 c + 4 g ..) + the scalar channel's (lane group 0 only)
    const float4* wq = reinterpret_cast<const float4*>(p.w_xout);
    float4 wm[2][2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t) wm[c][t] = wq[(c * 2 + t) * 64 + lane];
    const float wx0 = p.w_xout[1024 + lane], wx1 = p.w_xout[1024 + 64 + lane];
    const float sc = p.scale[32], sh = p.shift[32];
    const float floor_ = (32 < p.act_split ? p.act_a : p.act_b) == ESTD_ACT_RELU ? 0.0f : ESTD_NO_FLOOR;

    // this wave's MFMA column groups of a plane (wave w takes groups w, w + 4, ...) and, per group, this lane's halo voxel -> in-plane offset
    constexpr int GPW = (GROUPS + NWAVES - 1) / NWAVES;   // 6 (5)
    unsigned voff[GPW];                                   // voxel index inside a plane (y * W + x), or OOB
#pragma unroll
    for (int k = 0; k < GPW; ++k) {
        const int grp = wave + NWAVES * k, u = grp * 16 + i;
        const int hy = u / IN_W, hx = u - hy * IN_W;
        const int y = h0 + hy - 1, x = w0 + hx - 1;
        voff[k] = (grp < GROUPS && u < HALO && y >= 0 && y < H && x >= 0 && x < W) ? (unsigned)(y * W + x) : 0xFFFFFFFFu;
    }
    // the output pixel of this thread and its 9 in-plane stencil offsets into P
    const int oy = tid / TW, ox = tid % TW;
    const bool ovalid = h0 + oy < H && w0 + ox < W;
    const int pbase = (oy * IN_W + ox) * PITCH;

    // one plane's operands of this lane: two quads of the 32 main channels + the scalar channel per column group
    struct Plane {
        float4 x0[GPW], x1[GPW];
        float xe[GPW];
    };
    auto request = [&](int d, Plane& q) {                 // input plane d (absent planes: nothing is requested, nothing is multiplied)
        if (d < 0 || d >= D) return;
        const unsigned plane = (unsigned)d * (unsigned)HW;
#pragma unroll
        for (int k = 0; k < GPW; ++k) {
            if (wave + NWAVES * k < GROUPS) {                  // (wave-uniform)
                const bool in = voff[k] != 0xFFFFFFFFu;
                const unsigned vo = in ? (plane + voff[k]) : 0u;
                const unsigned bo = in ? vo * (unsigned)p.in_stride * 4u + (unsigned)g * 16u : OOB_OFFSET;
                q.x0[k] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, bo, 0, 0));
                q.x1[k] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, bo, 64, 0));
                q.xe[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ex, (in && g == 0) ? vo * 4u : OOB_OFFSET, 0, 0));
            }
        }
    };
    float A = 0.0f, B = 0.0f;                             // running sums: output plane d - 1 (depth taps 0, 1 so far) and d (depth tap 0)
    auto step = [&](int d, const Plane& q) {              // multiply input plane d, fold it into the running sums, finish output plane d - 1
        float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f;
        if (d >= 0 && d < D) {                            // (workgroup-uniform)
#pragma unroll
            for (int k = 0; k < GPW; ++k) {
                const int grp = wave + NWAVES * k;
                if (grp < GROUPS) {
                    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xv0 = e == 0 ? q.x0[k].x : e == 1 ? q.x0[k].y : e == 2 ? q.x0[k].z : q.x0[k].w;
                        const float xv1 = e == 0 ? q.x1[k].x : e == 1 ? q.x1[k].y : e == 2 ? q.x1[k].z : q.x1[k].w;
                        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(e == 0 ? wm[0][0].x : e == 1 ? wm[0][0].y : e == 2 ? wm[0][0].z : wm[0][0].w, xv0, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(e == 0 ? wm[0][1].x : e == 1 ? wm[0][1].y : e == 2 ? wm[0][1].z : wm[0][1].w, xv0, a1, 0, 0, 0);
                        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(e == 0 ? wm[1][0].x : e == 1 ? wm[1][0].y : e == 2 ? wm[1][0].z : wm[1][0].w, xv1, a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(e == 0 ? wm[1][1].x : e == 1 ? wm[1][1].y : e == 2 ? wm[1][1].z : wm[1][1].w, xv1, a1, 0, 0, 0);
                    }
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wx0, q.xe[k], a0, 0, 0, 0);     // the scalar channel: k-step (s, 0, 0, 0)
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wx1, q.xe[k], a1, 0, 0, 0);
                    float* pv = P + (grp * 16 + i) * PITCH + 4 * g;                             // taps 4 g .. and 16 + 4 g .. of voxel i
                    pv[0] = a0[0]; pv[1] = a0[1]; pv[2] = a0[2]; pv[3] = a0[3];
                    pv[16] = a1[0]; pv[17] = a1[1]; pv[18] = a1[2]; pv[19] = a1[3];
                }
            }
            __syncthreads();
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float* r = P + pbase + (kh * IN_W + kw) * PITCH + kh * 3 + kw;
                    s0 += r[0];
                    s1 += r[9];
                    s2 += r[18];
                }
            __syncthreads();                              // P is rewritten by the next plane
        }
        // plane d carries depth tap kd into output plane d + 1 - kd
        const float done = A + s2;                        // output plane d - 1 is complete
        A = B + s1;
        B = s0;
        const int od = d - 1;
        if (od >= d0 && od < d1 && ovalid) {
            const float v = fmaxf(fmaf(done, sc, sh), floor_);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, (unsigned)(((size_t)od * HW + (size_t)(h0 + oy) * W + (w0 + ox)) * 4u), 0, 0);
        }
    };
    // two register sets in turn: the next plane is requested before the current one is multiplied
    Plane qa, qb;
    request(d0 - 1, qa);
    for (int d = d0 - 1; d <= d1; d += 2) {               // INPUT planes d0 - 1 .. d1
        request(d + 1, qb);
        step(d, qa);
        if (d + 1 <= d1) {
            request(d + 2, qa);
            step(d + 1, qb);
        }
    }
}

}  // namespace

extern "C" int estd_conv3d_k3_xout(const estd_conv3d_desc* dp, estd_stream_t s)
{
    if (!dp) return ESTD_ERR_ARG;
    const estd_conv3d_desc& d = *dp;
    if (d.N <= 0 || d.D <= 0 || d.H <= 0 || d.W <= 0) return ESTD_ERR_ARG;
    if (!d.in_main || !d.in_extra || !d.w_xout || !d.scale || !d.shift || !d.out_extra) return ESTD_ERR_ARG;
    if (d.cin_main != 32 || d.in_stride < 32 || (d.in_stride & 3)) return ESTD_ERR_UNSUPPORTED;
    if ((long long)d.D * d.H * d.W * d.in_stride * 4 >= 0x7fffff00LL) return ESTD_ERR_UNSUPPORTED;      // 32-bit byte offsets inside one volume
    const int tiles_w = (d.W + TW - 1) / TW, tiles_h = (d.H + TH - 1) / TH, dsegs = (d.D + DSEG - 1) / DSEG;
    const long long grid = (long long)d.N * tiles_h * tiles_w * dsegs;
    if (grid > 0x7fffffffLL) return ESTD_ERR_ARG;
    constexpr int LDS_BYTES = GROUPS * 16 * PITCH * 4;
    estd_allow_dynamic_lds<conv3d_xout_kernel>(LDS_BYTES);
    hipLaunchKernelGGL(conv3d_xout_kernel, dim3((unsigned)grid), dim3(NT), LDS_BYTES, estd_stream(s), d, tiles_w, tiles_h, dsegs);
    return ESTD_LAUNCH_CHECK();
}
