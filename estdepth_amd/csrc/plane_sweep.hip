// plane_sweep.hip -- camera algebra, plane-sweep homography warp and the fused cost-volume front.
//
// Reference semantics restated (file:line into the reference repo):
//   utils/homo_utils.py:458-504  homo_warping            (bilinear, zeros, align_corners=False, |xn|>1 -> 2)
//   hybrid_models/model_hybrid.py:62-102 get_costvolume  (pre0 = 1x1x1 conv 64->32 + BN on cat[ref, warped])
// HBM-bound: the D x H x W x 32 volume is written exactly once, as whole 128-byte voxel records
// (8 lanes x float4), the 2.5 MB source map stays in L2.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <math.h>
#include <stdint.h>

#include "estd_hip.h"
#include "estd_common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// tiny fp64 matrix helpers (single thread)
__device__ void inv4(const double* a, double* out)
{
    double m[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) { m[r][c] = a[r * 4 + c]; m[r][4 + c] = (r == c) ? 1.0 : 0.0; }
    for (int col = 0; col < 4; ++col) {
        int piv = col;
        double best = fabs(m[col][col]);
        for (int r = col + 1; r < 4; ++r) if (fabs(m[r][col]) > best) { best = fabs(m[r][col]); piv = r; }
        if (piv != col) for (int c = 0; c < 8; ++c) { double t = m[col][c]; m[col][c] = m[piv][c]; m[piv][c] = t; }
        const double inv = 1.0 / m[col][col];
        for (int c = 0; c < 8; ++c) m[col][c] *= inv;
        for (int r = 0; r < 4; ++r) if (r != col) {
            const double f = m[r][col];
            for (int c = 0; c < 8; ++c) m[r][c] -= f * m[col][c];
        }
    }
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) out[r * 4 + c] = m[r][4 + c];
}

__device__ void inv3(const double* a, double* out)
{
    double e[16] = {a[0], a[1], a[2], 0, a[3], a[4], a[5], 0, a[6], a[7], a[8], 0, 0, 0, 0, 1};
    double o[16];
    inv4(e, o);
    out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
    out[3] = o[4]; out[4] = o[5]; out[5] = o[6];
    out[6] = o[8]; out[7] = o[9]; out[8] = o[10];
}

__device__ void mul4(const double* a, const double* b, double* o)
{
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) {
        double s = 0.0;
        for (int k = 0; k < 4; ++k) s += a[r * 4 + k] * b[k * 4 + c];
        o[r * 4 + c] = s;
    }
}

// proj = K-projected extrinsic (model_hybrid.py:85-88): rows 0..2 = K @ E[:3,:4], row 3 = E[3,:]
__device__ void pose_to_proj(const float* pose, const float* K, double* proj)
{
    double p[16], e[16];
    for (int k = 0; k < 16; ++k) p[k] = pose[k];
    inv4(p, e);                                     // extrinsic = inverse(cam_pose)  (:74,:83)
    for (int k = 0; k < 16; ++k) e[k] = (double)(float)e[k];   // the reference holds it in fp32
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 4; ++c) {
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s += (double)K[r * 3 + k] * e[k * 4 + c];
        proj[r * 4 + c] = (double)(float)s;
    }
    for (int c = 0; c < 4; ++c) proj[12 + c] = e[12 + c];
}

__device__ void pair_proj(const double* sp, const double* rp, float* out12)
{
    double ri[16], pr[16];
    inv4(rp, ri);
    for (int k = 0; k < 16; ++k) ri[k] = (double)(float)ri[k];
    mul4(sp, ri, pr);                               // homo_utils.py:469
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) out12[r * 3 + c] = (float)pr[r * 4 + c];   // rot  (:470)
        out12[9 + r] = (float)pr[r * 4 + 3];                                   // trans (:471)
    }
}

__global__ void cam_pair_proj_kernel(const float* sp, const float* rp, float* out12)
{
    if (threadIdx.x || blockIdx.x) return;
    double a[16], b[16];
    for (int k = 0; k < 16; ++k) { a[k] = sp[k]; b[k] = rp[k]; }
    pair_proj(a, b, out12);
}

__global__ void cam_sweep_proj_kernel(const float* ref_pose, const float* src_pose, const float* K, float* out12)
{
    if (threadIdx.x || blockIdx.x) return;
    double sp[16], rp[16];
    pose_to_proj(src_pose, K, sp);
    pose_to_proj(ref_pose, K, rp);
    pair_proj(sp, rp, out12);
}

__global__ void cam_volume_mats_kernel(const float* pose_j, const float* pose_i, const float* K, float* out30)
{
    if (threadIdx.x || blockIdx.x) return;
    double rel[16];
    if (pose_i) {
        double pi[16], pj[16], ii[16];
        for (int k = 0; k < 16; ++k) { pi[k] = pose_i[k]; pj[k] = pose_j[k]; }
        inv4(pi, ii);
        for (int k = 0; k < 16; ++k) ii[k] = (double)(float)ii[k];
        mul4(pj, ii, rel);                          // hybrid_depth_decoder.py:235  (Q8: P_j @ P_i^-1)
        for (int k = 0; k < 16; ++k) rel[k] = (double)(float)rel[k];
    } else {
        for (int k = 0; k < 16; ++k) rel[k] = pose_j[k];
    }
    double m[16], kd[9], ki[9];
    inv4(rel, m);                                   // homo_utils.py:258
    for (int k = 0; k < 9; ++k) kd[k] = K[k];
    inv3(kd, ki);                                   // homo_utils.py:51
    for (int k = 0; k < 9; ++k) out30[k] = (float)ki[k];
    for (int k = 0; k < 12; ++k) out30[9 + k] = (float)m[k];
    for (int k = 0; k < 9; ++k) out30[21 + k] = K[k];
}

// ------------------------------------------------------------------------------------------------
struct Bilin {
    int o00, o01, o10, o11;     // pixel offsets (y*W+x), 0 when masked
    float w00, w01, w10, w11;   // weights, 0 when the corner is out of bounds
};

// homo_utils.py:479-501 for one (x, y, depth plane)
__device__ __forceinline__ Bilin sweep_coords(const float* __restrict__ P, float dv, int x, int y, int H, int W)
{
    // Rounding sequence of the reference's torch-CPU composition, op for op (the |norm| > 1 mask is discontinuous, so the
    // coordinates must be bit-identical): rot @ [x, y, 1] is a matmul (homo_utils.py:479) whose GEMM kernel accumulates
    // k = 0,1,2 in order with fused multiply-adds; the rest are elementwise ATen ops with their own rounding (:480-485).
#pragma clang fp contract(off)
    const float fx = (float)x, fy = (float)y;
    const float r0 = fmaf(P[1], fy, P[0] * fx) + P[2];
    const float r1 = fmaf(P[4], fy, P[3] * fx) + P[5];
    const float r2 = fmaf(P[7], fy, P[6] * fx) + P[8];
    const float p0 = r0 * dv + P[9];
    const float p1 = r1 * dv + P[10];
    const float p2 = r2 * dv + P[11];
    const float den = p2 + 1e-8f;
    const float px = p0 / den, py = p1 / den;
    float xn = px / ((float)(W - 1) * 0.5f) - 1.0f;
    float yn = py / ((float)(H - 1) * 0.5f) - 1.0f;
    if (xn > 1.0f || xn < -1.0f) xn = 2.0f;
    if (yn > 1.0f || yn < -1.0f) yn = 2.0f;
    const float ix = ((xn + 1.0f) * (float)W - 1.0f) * 0.5f;
    const float iy = ((yn + 1.0f) * (float)H - 1.0f) * 0.5f;
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float tx = ix - fx0, ty = iy - fy0;
    // NaN coordinates fail every comparison -> all four corners invalid (ATen behaviour)
    const bool finite = (ix == ix) && (iy == iy);
    const int x0 = finite ? (int)fx0 : -2, y0 = finite ? (int)fy0 : -2;
    const int x1 = x0 + 1, y1 = y0 + 1;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W;
    const bool vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    Bilin b;
    b.w00 = (vx0 && vy0) ? (1.0f - tx) * (1.0f - ty) : 0.0f;
    b.w01 = (vx1 && vy0) ? tx * (1.0f - ty) : 0.0f;
    b.w10 = (vx0 && vy1) ? (1.0f - tx) * ty : 0.0f;
    b.w11 = (vx1 && vy1) ? tx * ty : 0.0f;
    b.o00 = (vx0 && vy0) ? y0 * W + x0 : 0;
    b.o01 = (vx1 && vy0) ? y0 * W + x1 : 0;
    b.o10 = (vx0 && vy1) ? y1 * W + x0 : 0;
    b.o11 = (vx1 && vy1) ? y1 * W + x1 : 0;
    return b;
}

// Level-1 operator: NCHW in, NCDHW out.  One thread per (d,y,x), loop over channels.
__global__ __launch_bounds__(256) void homo_warping_kernel(const float* __restrict__ src, const float* __restrict__ P,
                                                           const float* __restrict__ dvals, int per_pixel, float* __restrict__ out,
                                                           int C, int D, int H, int W)
{
    // per_pixel: dvals = [D][H][W] depth hypotheses per pixel (homo_utils.py:462: "depth_values: [B, Ndepth] o [B, Ndepth, H, W]")
    const long long HW = (long long)H * W;
    const long long total = (long long)D * HW;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(idx % W);
        const int y = (int)((idx / W) % H);
        const int d = (int)(idx / HW);
        const Bilin b = sweep_coords(P, per_pixel ? dvals[idx] : dvals[d], x, y, H, W);
        for (int c = 0; c < C; ++c) {
            const float* s = src + (long long)c * HW;
            const float v = s[b.o00] * b.w00 + s[b.o01] * b.w01 + s[b.o10] * b.w10 + s[b.o11] * b.w11;
            out[(long long)c * total + idx] = v;
        }
    }
}

// out[p][o] = sum_c w[o][c] * in[c][p] + bias[o]; weights cached in LDS; one thread per pixel.
__global__ __launch_bounds__(256) void mix1x1_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                     int Cin, int Cout, int HW)
{
    __shared__ float sw[64 * 64 + 64];
    for (int k = threadIdx.x; k < Cin * Cout; k += blockDim.x) sw[k] = w[k];
    for (int k = threadIdx.x; k < Cout; k += blockDim.x) sw[64 * 64 + k] = bias ? bias[k] : 0.0f;
    __syncthreads();
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= HW) return;
    for (int o = 0; o < Cout; o += 4) {
        float a[4] = {sw[64 * 64 + o], sw[64 * 64 + o + 1], sw[64 * 64 + o + 2], sw[64 * 64 + o + 3]};
        for (int c = 0; c < Cin; ++c) {
            const float xv = in[(long long)c * HW + p];   // L1/L2 resident: the map is 2.5 MB
            a[0] += sw[o * Cin + c] * xv;
            a[1] += sw[(o + 1) * Cin + c] * xv;
            a[2] += sw[(o + 2) * Cin + c] * xv;
            a[3] += sw[(o + 3) * Cin + c] * xv;
        }
        *reinterpret_cast<float4*>(out + (long long)p * Cout + o) = make_float4(a[0], a[1], a[2], a[3]);
    }
}

// Fused warp + pre0.  Two phases per 256-voxel block:
//   1. one thread per voxel evaluates the homography (4 offsets + 4 weights) ONCE and parks them in LDS
//      (the coordinate math is ~90 VALU ops: doing it in each of the 8 channel lanes made the kernel VALU-bound);
//   2. 8 lanes per voxel (one float4 of the 32 channels each) fetch the record from LDS (16-byte broadcast reads),
//      gather the four 128-byte source records from the L2-resident map and write whole 128-byte voxel records.
constexpr int SWEEP_VOX_PER_BLOCK = 256;

// Block -> voxel mapping: hardware hands consecutive workgroups to the eight XCDs round-robin.  With the linear mapping every XCD
// sweeps the whole image in every depth plane, so each 4 MB L2 has to hold both 2.5 MB feature maps under a 157 MB output stream:
// measured 2 x 96 MiB fetched per launch for 4.9 MB of maps.  BANDED (H a multiple of 8): XCD k = blockIdx % 8 owns image rows
// [k H/8, (k+1) H/8) of every plane -- its share of the reference map and the band of the source map those rows project into.
template <bool BANDED>
__global__ __launch_bounds__(256) void homo_warp_costvol_kernel(const float* __restrict__ src, const float* __restrict__ ref,
                                                                const float* __restrict__ P, const float* __restrict__ dvals,
                                                                float* __restrict__ out, int D, int H, int W)
{
    __shared__ __attribute__((aligned(16))) int s_off[SWEEP_VOX_PER_BLOCK][4];
    __shared__ __attribute__((aligned(16))) float s_w[SWEEP_VOX_PER_BLOCK][4];
    __shared__ int s_idx[SWEEP_VOX_PER_BLOCK];           // linear voxel index (d, y, x) of the block's voxels; -1: none
    const int HW = H * W;
    {
        int x, y, d;
        bool valid;
        if (BANDED) {
            const int k = blockIdx.x & 7, j = blockIdx.x >> 3;
            const int Hb = H >> 3, band_plane = Hb * W;
            const long long m = (long long)j * SWEEP_VOX_PER_BLOCK + threadIdx.x;     // index inside the band's (d, y, x) order
            valid = m < (long long)D * band_plane;
            const long long mm = valid ? m : 0;
            d = (int)(mm / band_plane);
            const int rem = (int)(mm % band_plane);
            y = k * Hb + rem / W;
            x = rem % W;
        } else {
            const long long raw = (long long)blockIdx.x * SWEEP_VOX_PER_BLOCK + threadIdx.x;
            valid = raw < (long long)D * HW;
            const long long idx = valid ? raw : 0;
            x = (int)(idx % W);
            y = (int)((idx / W) % H);
            d = (int)(idx / HW);
        }
        const Bilin b = sweep_coords(P, dvals[d], x, y, H, W);
        *reinterpret_cast<int4*>(s_off[threadIdx.x]) = make_int4(b.o00, b.o01, b.o10, b.o11);
        *reinterpret_cast<float4*>(s_w[threadIdx.x]) = make_float4(b.w00, b.w01, b.w10, b.w11);
        s_idx[threadIdx.x] = valid ? (d * H + y) * W + x : -1;
    }
    __syncthreads();
    const int sub = threadIdx.x & 7;
    const int grp = threadIdx.x >> 3;
    const float4* s4 = reinterpret_cast<const float4*>(src) + sub;
    const float4* r4 = reinterpret_cast<const float4*>(ref) + sub;
    float4* o4 = reinterpret_cast<float4*>(out) + sub;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int v = s * 32 + grp;
        const int idx = s_idx[v];
        if (idx >= 0) {
            const int4 of = *reinterpret_cast<const int4*>(s_off[v]);
            const float4 w = *reinterpret_cast<const float4*>(s_w[v]);
            const float4 c00 = s4[(long long)of.x * 8], c01 = s4[(long long)of.y * 8];
            const float4 c10 = s4[(long long)of.z * 8], c11 = s4[(long long)of.w * 8];
            const float4 rf = r4[(long long)(idx % HW) * 8];
            float4 o;
            o.x = rf.x + (c00.x * w.x + c01.x * w.y + c10.x * w.z + c11.x * w.w);
            o.y = rf.y + (c00.y * w.x + c01.y * w.y + c10.y * w.z + c11.y * w.w);
            o.z = rf.z + (c00.z * w.x + c01.z * w.y + c10.z * w.z + c11.z * w.w);
            o.w = rf.w + (c00.w * w.x + c01.w * w.y + c10.w * w.z + c11.w * w.w);
            typedef float nt_f4 __attribute__((ext_vector_type(4)));
            nt_f4 ov = {o.x, o.y, o.z, o.w};
            __builtin_nontemporal_store(ov, reinterpret_cast<nt_f4*>(&o4[(long long)idx * 8]));   // streamed once, read later by the conv
        }
    }
}

}  // namespace

extern "C" int estd_cam_pair_proj(const float* sp, const float* rp, float* out12, estd_stream_t s)
{
    if (!sp || !rp || !out12) return ESTD_ERR_ARG;
    hipLaunchKernelGGL(cam_pair_proj_kernel, dim3(1), dim3(1), 0, estd_stream(s), sp, rp, out12);
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_cam_sweep_proj(const float* ref_pose, const float* src_pose, const float* K, float* out12, estd_stream_t s)
{
    if (!ref_pose || !src_pose || !K || !out12) return ESTD_ERR_ARG;
    hipLaunchKernelGGL(cam_sweep_proj_kernel, dim3(1), dim3(1), 0, estd_stream(s), ref_pose, src_pose, K, out12);
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_cam_volume_mats(const float* pose_j, const float* pose_i, const float* K, float* out30, estd_stream_t s)
{
    if (!pose_j || !K || !out30) return ESTD_ERR_ARG;
    hipLaunchKernelGGL(cam_volume_mats_kernel, dim3(1), dim3(1), 0, estd_stream(s), pose_j, pose_i, K, out30);
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_homo_warping(const float* src, const float* P, const float* dvals, float* out,
                                 int C, int D, int H, int W, estd_stream_t s)
{
    if (!src || !P || !dvals || !out || C <= 0 || D <= 0 || H <= 0 || W <= 0) return ESTD_ERR_ARG;
    const long long total = (long long)D * H * W;
    const int grid = (int)((total + 255) / 256 > 65535 * 8 ? 65535 * 8 : (total + 255) / 256);
    hipLaunchKernelGGL(homo_warping_kernel, dim3(grid), dim3(256), 0, estd_stream(s), src, P, dvals, 0, out, C, D, H, W);
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_homo_warping_px(const float* src, const float* P, const float* depth_dhw, float* out,
                                    int C, int D, int H, int W, estd_stream_t s)
{
    if (!src || !P || !depth_dhw || !out || C <= 0 || D <= 0 || H <= 0 || W <= 0) return ESTD_ERR_ARG;
    const long long total = (long long)D * H * W;
    const int grid = (int)((total + 255) / 256 > 65535 * 8 ? 65535 * 8 : (total + 255) / 256);
    hipLaunchKernelGGL(homo_warping_kernel, dim3(grid), dim3(256), 0, estd_stream(s), src, P, depth_dhw, 1, out, C, D, H, W);
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_mix1x1_chw_to_hwc(const float* in, const float* w, const float* bias, float* out,
                                      int Cin, int Cout, int HW, estd_stream_t s)
{
    if (!in || !w || !out || Cin <= 0 || Cin > 64 || Cout <= 0 || Cout > 64 || (Cout & 3) || HW <= 0) return ESTD_ERR_ARG;
    hipLaunchKernelGGL(mix1x1_kernel, dim3((HW + 255) / 256), dim3(256), 0, estd_stream(s), in, w, bias, out, Cin, Cout, HW);
    return ESTD_LAUNCH_CHECK();
}

extern "C" int estd_homo_warp_costvol(const float* src, const float* ref, const float* P, const float* dvals,
                                      float* out, int D, int H, int W, estd_stream_t s)
{
    if (!src || !ref || !P || !dvals || !out || D <= 0 || H <= 0 || W <= 0) return ESTD_ERR_ARG;
    const long long total = (long long)D * H * W;
    if (total >= 0x7fffffffLL) return ESTD_ERR_UNSUPPORTED;                 // 32-bit voxel indices
    const long long per_block = SWEEP_VOX_PER_BLOCK;
    if ((H & 7) == 0 && !getenv("ESTD_SWEEP_LINEAR")) {
        const long long per_band = (total / 8 + per_block - 1) / per_block;          // blocks per XCD band
        hipLaunchKernelGGL(homo_warp_costvol_kernel<true>, dim3((unsigned)(8 * per_band)), dim3(256), 0, estd_stream(s), src, ref, P, dvals,
                           out, D, H, W);
    } else {
        hipLaunchKernelGGL(homo_warp_costvol_kernel<false>, dim3((unsigned)((total + per_block - 1) / per_block)), dim3(256), 0,
                           estd_stream(s), src, ref, P, dvals, out, D, H, W);
    }
    return ESTD_LAUNCH_CHECK();
}
