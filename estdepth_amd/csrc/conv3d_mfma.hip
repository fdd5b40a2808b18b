// conv3d_mfma.hip -- 3x3x3 convolution as an implicit GEMM on gfx950 fp32 MFMA.
//
// Replaces the Conv3d(+BatchNorm3d eval)(+ReLU/Tanh) stacks of the reference
// (networks/layers_op.py:16-39; instantiated at hybrid_models/model_hybrid.py:59-60 and
// hybrid_models/hybrid_depth_decoder.py:84-112) and the biased Conv3d pair of
// transformer/epipolar_transformer.py:21,:26.  >70 % of the FLOPs of a forward pass run here.
//
// Design (CDNA4):
//   * GEMM view per tap: M = 16 consecutive voxels along W, N = 16 output channels, K = 4 input
//     channels -> v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain; gfx950 has no TF32).
//   * One 256-thread workgroup (4 waves) owns an output tile of 1 x 8 x 16 voxels; its input brick
//     (3 x 10 x 18 voxels x C channels, zero padded) is staged ONCE in LDS (69 KB for C=32, so two
//     workgroups fit the 160 KB of a CU and one stages while the other computes).
//   * LDS voxel records are XOR-swizzled in 16-byte chunks so the per-tap ds_read_b128 of
//     "16 voxels x 4 channels" is (nearly) bank-conflict free for any tap shift.
//   * The K order inside a tap is permuted (lane group g reads channels 4g..4g+3 and 16+4g..16+4g+3)
//     so that each lane fetches its A operands with two 16-byte LDS reads; the packed weights use the
//     same permutation (estdepth_amd/packing.py).
//   * Weights (B fragments) stream from L2 in a pre-packed [tap][quad][lane] order: one coalesced
//     1 KB load per quad, prefetched one tap ahead.
//   * Output channels are interleaved across the two N tiles (channel = 2*col + tile) so that a
//     lane owns two adjacent channels and the epilogue stores full 128-byte voxel records.
//   * Epilogue fuses: folded BatchNorm / bias, ReLU / tanh (split per channel range), residual add,
//     running mean over source views, the 33rd channel, the 1x1x1 stereo head, and the partial
//     sums for GroupNorm(1 group).
//   * blockIdx -> tile mapping is XCD-aware: each of the 8 XCDs walks a contiguous range of tiles
//     so neighbouring bricks (shared halos) hit the same private L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "estd_hip.h"
#include "estd_common.h"

#ifdef ESTD_TIMELINE
#define ESTD_STATS_ON 0
#else
#define ESTD_STATS_ON 1
#endif
#ifndef ESTD_STORE_AUX
#define ESTD_STORE_AUX 0   // buffer-store cache policy bits (2 = nt)
#endif
#ifndef ESTD_ABL
#define ESTD_ABL 0   // timing ablations only (tools/ablate_conv.sh); results are wrong when != 0
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8;    // tile rows
constexpr int TW = 16;   // tile columns (= one MFMA M tile)
constexpr int IN_D = 3, IN_H = TH + 2, IN_W = TW + 2;
constexpr int NVOX_IN = IN_D * IN_H * IN_W;   // 540
constexpr int MT = 2;    // M tiles (rows) per wave: 4 waves x 2 = 8 rows

// sum over the 16 lanes of a DPP row (every lane gets the total): two quad permutations + two row rotations, all folded into v_add_f32
template <int CTRL>
__device__ __forceinline__ float dpp_add(float v)
{
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v)
{
    v = dpp_add<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add<0x124>(v);     // row_ror:4
    v = dpp_add<0x128>(v);     // row_ror:8
    return v;
}

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == ESTD_ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == ESTD_ACT_TANH) return tanhf(v);
    return v;
}

// byte offset of 16-byte chunk c of LDS voxel record v
template <int CM>
__device__ __forceinline__ int lds_chunk_off(int v, int c) {
    if (CM == 32) return v * 128 + ((c ^ ((v >> 1) & 7)) << 4);
    else          return v * 64 + ((c ^ ((v >> 2) & 3)) << 4);
}

typedef unsigned int u32x4 __attribute__((__vector_size__(16)));   // the type raw_buffer_load_b128 returns (GCC vector)
typedef unsigned int u32x2 __attribute__((__vector_size__(8)));
constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;    // beyond num_records of any descriptor: loads return 0, stores are dropped

__device__ __forceinline__ float4 as_float4(u32x4 v)
{
    // NB: __builtin_bit_cast(float, v[i]) on a vector element is miscompiled by this clang (every lane reads
    // element 0); a whole-object copy is the safe form.
    float4 f;
    __builtin_memcpy(&f, &v, 16);
    return f;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, size_t elems)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(elems * 4), 0x00020000);
}

// Workgroup barrier that only orders LDS traffic.  __syncthreads() would also drain vmcnt, i.e. wait for the
// previous tile's output stores (and the in-flight prefetch) to complete on every tile.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

constexpr int SL_VOX = IN_H * IN_W;     // 180 voxels per input slice (one depth plane of the brick, with halo)

// Persistent workgroups with a SLIDING DEPTH WINDOW.
//   * The flattened tile list (column-major: tiles of one (n, h-tile, w-tile) column are consecutive in d) is cut
//     into gridDim.x equal ranges; XCD x owns a contiguous block of ranges (L2 locality of the shared halos).
//   * Inside a range the workgroup walks d upwards keeping the three input slices d-1, d, d+1 in a 3-slot LDS
//     ring: per tile only ONE new slice (180 voxels) is fetched instead of the whole 540-voxel brick.
//   * The next slice is fetched into registers while the 864 MFMAs of the current tile run (one 16-byte load in
//     each of the first taps), then written to the free ring slot between two barriers.
//   * All global traffic uses wave-uniform buffer descriptors: per-lane 32-bit offsets are computed once per
//     column segment, the per-tile part is a scalar offset, out-of-volume lanes read zeros / drop stores.
template <int CM, int NT, bool EXTRA, bool XOUT>
__global__ __launch_bounds__(256, 2) void conv3d_k3_kernel(const estd_conv3d_desc p, int tiles_w, int tiles_h, int total_tiles)
{
    constexpr int CH = CM / 4;          // 16-byte chunks per voxel
    constexpr int KS = CM / 4;          // MFMA k-steps per tap
    constexpr int QN = (KS * NT) / 4;   // float4 weight quads per lane per tap
    constexpr int XS = 7;               // k-steps of the extra (scalar) input channel: 27 taps padded to 28
    constexpr int XQ = (XS * NT + 3) / 4;
    constexpr int SL_EL = SL_VOX * CH;                 // 16-byte chunks per slice: 1440 (CM=32) / 720 (CM=16)
    constexpr int SIT = (SL_EL + 255) / 256;           // chunk loads per thread per slice: 6 / 3
    constexpr int SLICE_BYTES = SL_VOX * CM * 4;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_main = smem;                                                  // [3][SLICE_BYTES]
    float* lds_extra = reinterpret_cast<float*>(smem + 3 * SLICE_BYTES);    // [3][SL_VOX]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;            // k index inside an MFMA
    const int i = lane & 15;            // M row (A) / N column (B, D)
    // MFMA row <-> voxel of a tile row: rows {0-3,12-15} = even voxels, rows {4-11} = odd voxels, so that the two halves of a
    // ds_read_b128 lane group ({0-3,12-15} of lane group g with {4-11} of g^1) never meet in a 256-byte bank row (csrc/conv3d_wino.hip)
    const int pi = i < 4 ? 2 * i : i < 12 ? 2 * i - 7 : 2 * i - 16;
    const int D = p.D, H = p.H, W = p.W;

    // ---- range of the flattened tile list owned by this workgroup ----
    int u, u_end;
    {
        const int G = gridDim.x, bid = blockIdx.x;
        const int r = ((G & 7) == 0) ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;
        u = (int)((long long)total_tiles * r / G);
        u_end = (int)((long long)total_tiles * (r + 1) / G);
    }
    if (u >= u_end) return;
    // per-lane constants of the epilogue
    const int cbase = (NT == 1) ? i : 2 * i;
    float sc[2], sh[2];
    sc[0] = p.scale[cbase]; sh[0] = p.shift[cbase];
    sc[1] = sc[0]; sh[1] = sh[0];
    if (NT >= 2) { sc[1] = p.scale[cbase + 1]; sh[1] = p.shift[cbase + 1]; }
    const int act0 = cbase < p.act_split ? p.act_a : p.act_b;   // both channels of a lane share the range (split is even)
    float sc2 = 0.f, sh2 = 0.f;
    if (XOUT) { sc2 = p.scale[32]; sh2 = p.shift[32]; }
    float hw = 0.f, hb = 0.f;
    if (NT == 1 && p.head_w) { hw = p.head_w[i]; hb = p.head_b[0]; }
    // stereo heads (the only NT = 1 callers with a head): nothing but the 1x1x1 head leaves the kernel and the activation is one of
    // none / ReLU for all 16 channels -> a straight-line epilogue (the generic one below branches per output element)
    const bool act_uniform = p.act_split <= 0 || p.act_split >= 16 || p.act_a == p.act_b;
    const int act_u = p.act_split <= 0 ? p.act_b : p.act_a;
    const bool head_only = NT == 1 && !XOUT && p.head_w && !p.out_main && !(ESTD_STATS_ON && p.stats_partials) && act_uniform &&
                           act_u != ESTD_ACT_TANH && !(ESTD_ABL & 128);
    const float head_floor = act_u == ESTD_ACT_RELU ? 0.f : ESTD_NO_FLOOR;

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w_main, (size_t)28 * QN * 256);
    // 16 -> 16 (stereo heads): one weight quad per lane and tap, 27 quads = 108 registers -- the whole filter stays in registers, no
    // weight stream at all (a tap is only 8 MFMAs = 256 matrix cycles, less than an L2 round trip next to another stream's kernels)
    constexpr bool WREG = (CM == 16 && NT == 1 && !EXTRA && !XOUT && !(ESTD_ABL & 64));
    float4 wreg[WREG ? 27 : 1];
    if (WREG) {
#pragma unroll
        for (int t = 0; t < 27; ++t) wreg[t] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, lane * 16, t * 1024, 0));
    }
    // 33rd OUTPUT channel (dres2): a GEMV, 1/16 efficient on MFMA -> computed on the VALU in the MFMA shadow from the
    // A fragments already in registers: w_xout = [28 taps][2 quads][64 lanes][4] + [2][64][4] extra-input taps
    const __amdgpu_buffer_rsrc_t rs_wxo = make_rsrc(XOUT ? p.w_xout : p.w_main, (size_t)(28 * 2 + 2) * 256);
    const size_t vol = (size_t)D * H * W;
    const int HW = H * W;
    const int row0 = wave * MT;   // first tile row of this wave
    const int wlane = lane * 16;

    while (u < u_end) {
        // ---- column segment [u, seg_end): same (n, h-tile, w-tile), consecutive d ----
        const int col = u / D;
        int d = u - col * D;
        const int twi = col % tiles_w, c2 = col / tiles_w;
        const int thi = c2 % tiles_h, n = c2 / tiles_h;
        const int tw0 = twi * TW, th0 = thi * TH;
        const int seg_end = min(u_end, (col + 1) * D);

        const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in_main + (size_t)n * vol * p.in_stride, vol * p.in_stride);
        __amdgpu_buffer_rsrc_t rs_ex = rs_in;
        if (EXTRA) rs_ex = make_rsrc(p.in_extra + (size_t)n * vol, vol);

        // per-thread slice-element constants of this segment (validity in y/x does not depend on d)
        unsigned voff[SIT];
        int loff[SIT];
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int e = tid + it * 256;
            const int vs = e / CH, c = e % CH;
            const int zy = vs / IN_W, zx = vs % IN_W;
            const int gy = th0 - 1 + zy, gx = tw0 - 1 + zx;
            const bool ok = e < SL_EL && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            voff[it] = ok ? (unsigned)((gy * W + gx) * p.in_stride + c * 4) * 4u : OOB_OFFSET;
            loff[it] = e < SL_EL ? lds_chunk_off<CM>(vs, c) : -1;
        }
        unsigned voffx = OOB_OFFSET;
        if (EXTRA) {
            const int zy = tid / IN_W, zx = tid % IN_W;
            const int gy = th0 - 1 + zy, gx = tw0 - 1 + zx;
            const bool ok = tid < SL_VOX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            voffx = ok ? (unsigned)(gy * W + gx) * 4u : OOB_OFFSET;
        }
        const int in_slice_bytes = HW * p.in_stride * 4;

        // epilogue lane offsets (bytes inside one depth plane); rows/columns past the volume are dropped
        const int ey0 = th0 + row0, ex0 = tw0 + (g == 0 ? 0 : g == 1 ? 1 : g == 2 ? 9 : 8);    // D row 4g + r = voxel pi(4g + r) = ex0 + 2r

        // output descriptors and per-lane output offsets of this segment (bytes inside one depth plane)
        __amdgpu_buffer_rsrc_t rs_out = rs_in, rs_res = rs_in, rs_res2 = rs_in;
        if (p.out_main) rs_out = make_rsrc(p.out_main + (size_t)n * vol * p.out_stride, vol * p.out_stride);
        if (p.residual) rs_res = make_rsrc(p.residual + (size_t)n * vol * p.out_stride, vol * p.out_stride);
        if (p.residual2) rs_res2 = make_rsrc(p.residual2 + (size_t)n * vol * p.out_stride, vol * p.out_stride);
        const int out_plane_bytes = HW * p.out_stride * 4;
        unsigned eoff[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int y = ey0 + m, x = ex0 + 2 * r;
                eoff[m][r] = (y < H && x < W) ? (unsigned)((y * W + x) * p.out_stride + cbase) * 4u : OOB_OFFSET;
            }

        // ---- epilogue of one finished tile (depth plane dd of this column) ----
        // head-only epilogue: lane (g, i) stores element i of its row group's 8 head outputs ((m, r) = (i >> 2, i & 3))
        __amdgpu_buffer_rsrc_t rs_head = rs_in;
        unsigned hoff = OOB_OFFSET;
        if (NT == 1 && head_only) {
            rs_head = make_rsrc(p.out_head + (size_t)n * vol, vol);
            const int y = ey0 + ((i >> 2) & 1), x = ex0 + 2 * (i & 3);
            hoff = (i < 4 * MT && y < H && x < W) ? (unsigned)(y * W + x) * 4u : OOB_OFFSET;
        }
        auto epilogue = [&](const f32x4 (&a)[MT][NT], const float (&ax)[MT], int dd) {
            if (NT == 1 && head_only) {
                float outv = 0.f;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = __builtin_fmaxf(__builtin_fmaf(a[m][0][r], sc[0], sh[0]), head_floor) * hw;
                        const float tot = row16_sum(v) + hb;
                        outv = (i == m * 4 + r) ? tot : outv;
                    }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, outv), rs_head, hoff, dd * HW * 4, 0);
                return;
            }
            // D layout: lane holds column j = i (N index) and rows 4g..4g+3 (M index = voxel along W).
            // channel of (tile nn, column j): NT==1 -> j ; NT>=2 -> 2j+nn for nn<2 ; nn==2 -> 32 (only j==0).
            double s_sum = 0.0, s_sq = 0.0;     // GroupNorm partials of this lane (its channels are in one group)
            const size_t plane = (size_t)n * vol + (size_t)dd * HW;     // voxel index of (n, d, 0, 0)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int y = ey0 + m;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int x = ex0 + 2 * r;
                    const bool valid = (y < H) && (x < W);
                    const size_t vox = plane + (size_t)y * W + x;
                    float v0 = a[m][0][r] * sc[0] + sh[0];
                    float v1 = 0.f;
                    if (NT >= 2) v1 = a[m][1][r] * sc[1] + sh[1];
                    if (ESTD_STATS_ON && p.stats_partials && valid) {
                        s_sum += (double)v0; s_sq += (double)v0 * (double)v0;
                        if (NT >= 2) { s_sum += (double)v1; s_sq += (double)v1 * (double)v1; }
                    }
                    v0 = act_apply(v0, act0);
                    if (NT >= 2) v1 = act_apply(v1, act0);
                    if (NT == 1 && p.head_w) {
                        // 1x1x1 head: reduce over the 16 channel lanes of this row group
                        float hsum = v0 * hw;
                        hsum += __shfl_xor(hsum, 1);
                        hsum += __shfl_xor(hsum, 2);
                        hsum += __shfl_xor(hsum, 4);
                        hsum += __shfl_xor(hsum, 8);
                        if (valid && i == 0) p.out_head[vox] = hsum + hb;
                    }
                    if (p.out_main && !(ESTD_ABL & 1)) {
                        // bounds-checked buffer ops: lanes outside the volume carry OOB_OFFSET (loads give 0, stores drop)
                        const unsigned eo = eoff[m][r];
                        const int so = dd * out_plane_bytes;
                        if (NT == 1) {
                            if (p.residual) v0 += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res, eo, so, 0));
                            if (p.residual2) v0 += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res2, eo, so, 0));
                            v0 *= p.out_scale;
                            if (p.accumulate) v0 += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_out, eo, so, 0));
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rs_out, eo, so, ESTD_STORE_AUX);
                        } else {
                            if (p.residual) {
                                const u32x2 rv = __builtin_amdgcn_raw_buffer_load_b64(rs_res, eo, so, 0);
                                float2 rr; __builtin_memcpy(&rr, &rv, 8);
                                v0 += rr.x; v1 += rr.y;
                            }
                            if (p.residual2) {
                                const u32x2 rv = __builtin_amdgcn_raw_buffer_load_b64(rs_res2, eo, so, 0);
                                float2 rr; __builtin_memcpy(&rr, &rv, 8);
                                v0 += rr.x; v1 += rr.y;
                            }
                            v0 *= p.out_scale; v1 *= p.out_scale;
                            if (p.accumulate) {
                                const u32x2 pv = __builtin_amdgcn_raw_buffer_load_b64(rs_out, eo, so, 0);
                                float2 pr; __builtin_memcpy(&pr, &pv, 8);
                                v0 += pr.x; v1 += pr.y;
                            }
                            const float2 ov = make_float2(v0, v1);
                            u32x2 od; __builtin_memcpy(&od, &ov, 8);
                            __builtin_amdgcn_raw_buffer_store_b64(od, rs_out, eo, so, ESTD_STORE_AUX);
                        }
                    }
                }
                if (XOUT) {
                    // ax[m] = channel 32 of voxel (row m, column i), identical in the four lane groups
                    const int xx = tw0 + pi;
                    if (g == 0 && y < H && xx < W) p.out_extra[plane + (size_t)y * W + xx] = act_apply(ax[m] * sc2 + sh2, p.act_b);
                }
            }

            if (ESTD_STATS_ON && p.stats_partials) {
                // group 0 = channels 0..15, group 1 = channels 16..31.  Lane's channels: cbase(,+1).
                const int grp = (cbase >= 16) ? 1 : 0;
                double a0 = grp == 0 ? s_sum : 0.0, q0 = grp == 0 ? s_sq : 0.0;
                double a1 = grp == 1 ? s_sum : 0.0, q1 = grp == 1 ? s_sq : 0.0;
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    a0 += __shfl_xor(a0, o); q0 += __shfl_xor(q0, o);
                    a1 += __shfl_xor(a1, o); q1 += __shfl_xor(q1, o);
                }
                // cross-wave reduction through a small LDS scratch placed after the ring
                double* red = reinterpret_cast<double*>(smem + 3 * SLICE_BYTES + (EXTRA ? 3 * SL_VOX * 4 : 0));
                if (lane == 0) { red[wave * 4 + 0] = a0; red[wave * 4 + 1] = q0; red[wave * 4 + 2] = a1; red[wave * 4 + 3] = q1; }
                __syncthreads();
                if (tid < 4) {
                    const double tot = red[tid] + red[4 + tid] + red[8 + tid] + red[12 + tid];
                    // partial index = canonical tile id (n, d, thi, twi) so the finalize order is launch-independent
                    const size_t tile_id = (((size_t)n * D + dd) * tiles_h + thi) * tiles_w + twi;
                    p.stats_partials[tile_id * 4 + tid] = tot;
                }
            }
        };
        f32x4 pend[MT][NT];
        float pend_x[MT] = {};
        int pend_d = 0;
        bool have_pend = false;

        float4 pf[SIT];
        float pfx = 0.f;
        bool first = true;

        for (; u < seg_end; ++u, ++d) {
            const int dm = d % 3;                                 // ring slot of slice d-1
            const int sb0 = dm * SLICE_BYTES;
            const int sb1 = (dm == 2 ? 0 : dm + 1) * SLICE_BYTES;
            const int sb2 = (dm == 0 ? 2 : dm - 1) * SLICE_BYTES;  // (dm+2)%3 : slot of slice d+1
            const int xb0 = dm * SL_VOX, xb1 = (dm == 2 ? 0 : dm + 1) * SL_VOX, xb2 = (dm == 0 ? 2 : dm - 1) * SL_VOX;

            lds_barrier();                        // every wave is done with the previous tile's slices
            if (first) {
                // prime the ring: slices d-1 and d straight to LDS, slice d+1 into the prefetch registers
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int sd = d - 1 + s;
                    const bool sv = (unsigned)sd < (unsigned)D;
                    float4 tmp[SIT];
#pragma unroll
                    for (int it = 0; it < SIT; ++it)
                        tmp[it] = sv ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[it], sd * in_slice_bytes, 0))
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
                    float tx = 0.f;
                    if (EXTRA && sv) tx = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ex, voffx, sd * HW * 4, 0));
#pragma unroll
                    for (int it = 0; it < SIT; ++it)
                        if (it < SIT - 1 || loff[it] >= 0)
                            *reinterpret_cast<float4*>(lds_main + (s == 0 ? sb0 : sb1) + loff[it]) = tmp[it];
                    if (EXTRA && tid < SL_VOX) lds_extra[(s == 0 ? xb0 : xb1) + tid] = tx;
                }
                {
                    const int sd = d + 1;
                    const bool sv = sd < D;
#pragma unroll
                    for (int it = 0; it < SIT; ++it)
                        pf[it] = sv ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[it], sd * in_slice_bytes, 0))
                                    : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (EXTRA) pfx = sv ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ex, voffx, sd * HW * 4, 0)) : 0.f;
                }
                first = false;
            }
            // slice d+1 (prefetched during the previous tile) -> its ring slot
#pragma unroll
            for (int it = 0; it < SIT; ++it)
                if ((it < SIT - 1 || loff[it] >= 0) && !(ESTD_ABL & 8))
                    *reinterpret_cast<float4*>(lds_main + sb2 + loff[it]) = pf[it];
            if (EXTRA && tid < SL_VOX) lds_extra[xb2 + tid] = pfx;
            lds_barrier();
#ifdef ESTD_TIMELINE
            if (tid == 0 && p.stats_partials) {     // debug build only: per-tile start stamps instead of GroupNorm sums
                const size_t tile_id = (((size_t)n * D + d) * tiles_h + thi) * tiles_w + twi;
                p.stats_partials[tile_id * 4 + 0] = (double)__builtin_amdgcn_s_memtime();
                p.stats_partials[tile_id * 4 + 1] = (double)blockIdx.x;
                p.stats_partials[tile_id * 4 + 2] = (double)wall_clock64();
            }
#endif
            if (have_pend) epilogue(pend, pend_x, pend_d);

            const bool has_next = (u + 1 < seg_end);            // wave-uniform
            // The XOUT variant is register-tight: make the lane ids opaque per tile so the 54 loop-invariant LDS offsets
            // are recomputed in the MFMA shadow instead of being hoisted out of the tile loop (which spills).
            int gi = g, ii = pi;
            if (XOUT) asm volatile("" : "+v"(gi), "+v"(ii));
            const bool next_valid = (d + 2 < D);
            const int next_soff = (d + 2) * in_slice_bytes;

            // ---- main loop: 27 taps x KS k-steps ----
            f32x4 acc[MT][NT];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < NT; ++nn) acc[m][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};

            float4 bcur[QN], bnext[QN];
            if (WREG) bcur[0] = wreg[0];
            float4 xo_cur[2], xo_next[2];
            float xacc[MT] = {};
            if (XOUT) {
#pragma unroll
                for (int q = 0; q < 2; ++q) xo_cur[q] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_wxo, wlane, q * 1024, 0));
            }
            if (!WREG) {
#pragma unroll
                for (int q = 0; q < QN; ++q) bcur[q] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, q * 1024, 0));
            }

#pragma unroll
            for (int tap = 0; tap < 27; ++tap) {
                const int kd = tap / 9, kh = (tap / 3) % 3, kw = tap % 3;
                const int sb = kd == 0 ? sb0 : kd == 1 ? sb1 : sb2;
                // next tap's weights (the packed buffer carries one padding tap)
#pragma unroll
                for (int q = 0; q < QN; ++q) {
                    if (WREG) bnext[q] = wreg[tap + 1 < 27 ? tap + 1 : 26];
                    else if (ESTD_ABL & 2) { bnext[q] = bcur[q]; asm volatile("" : "+v"(bnext[q].x)); }
                    else bnext[q] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, ((tap + 1) * QN + q) * 1024, 0));
                }
                if (XOUT) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        xo_next[q] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_wxo, wlane, ((tap + 1) * 2 + q) * 1024, 0));
                }
                // one chunk of the NEXT tile's new slice per tap
                if (has_next && !(ESTD_ABL & 8)) {
                    if (tap < SIT)
                        pf[tap] = next_valid ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[tap], next_soff, 0))
                                             : make_float4(0.f, 0.f, 0.f, 0.f);
                    if (EXTRA && tap == SIT)
                        pfx = next_valid ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ex, voffx, (d + 2) * HW * 4, 0)) : 0.f;
                }
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int vs = (row0 + m + kh) * IN_W + kw + ii;
                    const int off0 = sb + lds_chunk_off<CM>(vs, gi);
                    float4 a0 = bcur[0];
                    if (!(ESTD_ABL & 4)) a0 = *reinterpret_cast<const float4*>(lds_main + off0);
                    float4 a1 = a0;
                    if (CM == 32 && !(ESTD_ABL & 4)) a1 = *reinterpret_cast<const float4*>(lds_main + (off0 ^ 64));
                    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    if (XOUT) {
                        const float wo[8] = {xo_cur[0].x, xo_cur[0].y, xo_cur[0].z, xo_cur[0].w, xo_cur[1].x, xo_cur[1].y, xo_cur[1].z, xo_cur[1].w};
#pragma unroll
                        for (int ks = 0; ks < KS; ++ks) xacc[m] = fmaf(av[ks], wo[ks], xacc[m]);
                    }
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
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
                if (XOUT) { xo_cur[0] = xo_next[0]; xo_cur[1] = xo_next[1]; }
                __builtin_amdgcn_sched_barrier(0);   // keep each tap's loads inside the tap (bounds live registers)
            }

            // ---- extra scalar input channel: its 27 taps form one more K chunk (28 = 7 x 4) ----
            if (EXTRA) {
                const __amdgpu_buffer_rsrc_t rs_wx = make_rsrc(p.w_extra, (size_t)XQ * 256);
                float4 wxx[2];
                if (XOUT) {
#pragma unroll
                    for (int q = 0; q < 2; ++q) wxx[q] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_wxo, wlane, (28 * 2 + q) * 1024, 0));
                }
                float4 bx[XQ];
#pragma unroll
                for (int q = 0; q < XQ; ++q) bx[q] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_wx, wlane, q * 1024, 0));
#pragma unroll
                for (int s = 0; s < XS; ++s) {
                    int tp = 4 * s + g;
                    tp = tp > 26 ? 26 : tp;           // tap 27 is padding (zero weight)
                    const int kd = tp / 9, kh = (tp / 3) % 3, kw = tp % 3;
                    const int xb = kd == 0 ? xb0 : kd == 1 ? xb1 : xb2;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const float a = lds_extra[xb + (row0 + m + kh) * IN_W + kw + pi];
                        if (XOUT) {
                            const float4 wq4 = wxx[s >> 2];
                            const float wv = (s & 3) == 0 ? wq4.x : (s & 3) == 1 ? wq4.y : (s & 3) == 2 ? wq4.z : wq4.w;
                            xacc[m] = fmaf(a, wv, xacc[m]);
                        }
#pragma unroll
                        for (int nn = 0; nn < NT; ++nn) {
                            const int idx = s * NT + nn;
                            const float4 bq = bx[idx >> 2];
                            const float b = (idx & 3) == 0 ? bq.x : (idx & 3) == 1 ? bq.y : (idx & 3) == 2 ? bq.z : bq.w;
                            acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[m][nn], 0, 0, 0);
                        }
                    }
                }
            }

            // the epilogue of this tile (global stores) is issued after the NEXT tile's ring update, so the
            // vmcnt(0) in front of that update never has to drain fresh stores
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < NT; ++nn) pend[m][nn] = acc[m][nn];
            if (XOUT) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    float v = xacc[m];                    // sum the four lane groups (4 x 8 channels of the same voxel)
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    pend_x[m] = v;
                }
            }
            pend_d = d;
            have_pend = true;
        }
        if (have_pend) epilogue(pend, pend_x, pend_d);     // last tile of the segment
    }
}

constexpr int PERSISTENT_WGS = 512;     // 256 CUs x 2 resident workgroups (LDS-limited)

template <int CM, int NT, bool EXTRA, bool XOUT>
int launch(const estd_conv3d_desc& d, hipStream_t stream)
{
    const int tiles_w = (d.W + TW - 1) / TW, tiles_h = (d.H + TH - 1) / TH;
    const int total = d.N * d.D * tiles_h * tiles_w;
    const int slots = estd_persistent_wgs(PERSISTENT_WGS / 256);
    int grid = total < slots ? total : slots;
    if (grid >= 8) grid &= ~7;
    const size_t lds = (size_t)3 * SL_VOX * CM * 4 + (EXTRA ? 3 * SL_VOX * 4 : 0) + 128;
    estd_allow_dynamic_lds<conv3d_k3_kernel<CM, NT, EXTRA, XOUT>>((int)lds);
    hipLaunchKernelGGL((conv3d_k3_kernel<CM, NT, EXTRA, XOUT>), dim3(grid), dim3(256), lds, stream, d, tiles_w, tiles_h, total);
    return hipGetLastError() == hipSuccess ? ESTD_OK : ESTD_ERR_LAUNCH;
}

}  // namespace

extern "C" int estd_conv3d_k3_grid(int N, int D, int H, int W)
{
    if (N <= 0 || D <= 0 || H <= 0 || W <= 0) return ESTD_ERR_ARG;
    return N * D * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
}

extern "C" int estd_conv3d_k3(const estd_conv3d_desc* dp, estd_stream_t s)
{
    if (!dp) return ESTD_ERR_ARG;
    const estd_conv3d_desc& d = *dp;
    hipStream_t stream = static_cast<hipStream_t>(s);
    if (d.N <= 0 || d.D <= 0 || d.H <= 0 || d.W <= 0) return ESTD_ERR_ARG;
    if (!d.in_main || !d.w_main || !d.scale || !d.shift) return ESTD_ERR_ARG;
    if (d.in_stride < d.cin_main || (d.in_stride & 3)) return ESTD_ERR_ARG;
    if (!d.out_main && !d.out_head) return ESTD_ERR_ARG;
    if (d.out_main && (d.out_stride < 16 * (d.n_tiles > 2 ? 2 : d.n_tiles) || (d.out_stride & 1))) return ESTD_ERR_ARG;
    if ((d.act_split & 1)) return ESTD_ERR_ARG;
    if (d.head_w && (d.n_tiles != 1 || !d.head_b || !d.out_head)) return ESTD_ERR_ARG;
    const bool extra = d.in_extra != nullptr;
    if (extra && !d.w_extra) return ESTD_ERR_ARG;
    if (d.n_tiles == 3 && (!d.out_extra || !d.w_xout)) return ESTD_ERR_ARG;
    if ((long long)d.N * d.D * ((d.H + TH - 1) / TH) * ((d.W + TW - 1) / TW) > 0x7fffffffLL) return ESTD_ERR_ARG;
    {   // buffer descriptors address one volume of the batch with 32-bit byte offsets
        const long long vox = (long long)d.D * d.H * d.W;
        const int widest = d.in_stride > d.out_stride ? d.in_stride : d.out_stride;
        if (vox * widest * 4 >= 0x7fffff00LL) return ESTD_ERR_UNSUPPORTED;
    }

    if (d.cin_main == 32 && d.n_tiles == 2 && !extra) return launch<32, 2, false, false>(d, stream);
    if (d.cin_main == 32 && d.n_tiles == 2 && extra)  return launch<32, 2, true, false>(d, stream);
    if (d.cin_main == 32 && d.n_tiles == 3 && extra)  return launch<32, 2, true, true>(d, stream);
    if (d.cin_main == 32 && d.n_tiles == 1 && !extra) return launch<32, 1, false, false>(d, stream);
    if (d.cin_main == 16 && d.n_tiles == 1 && !extra) return launch<16, 1, false, false>(d, stream);
    return ESTD_ERR_UNSUPPORTED;
}
