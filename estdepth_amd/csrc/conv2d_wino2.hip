// conv2d_wino2.hip -- 3x3 (stride 1, dilation 1) 2D convolution on NHWC maps with BOTH image axes in Winograd F(2,3) form, F(2x2, 3x3), on
// gfx950 fp32 MFMA, with the folded BatchNorm / ReLU / residual epilogue of csrc/conv2d_mfma.hip (same operator, same descriptor).
//
// the PSMNet matching-feature extractor (networks/psm_submodule.py:14-60,112-114), the ResNet-50 stride-1
// 3x3 convolutions (hybrid_models/resnet_encoder.py:43-49) and the 2D decoder blocks (hybrid_depth_decoder.py:17-30).  The row-only
// kernel (csrc/conv2d_wino.hip) issues 12 tap products per two output rows = 2/3 of the direct kernel's MFMAs; a timing ablation of it
// with every third tap's MFMAs removed runs 21-27 % faster (profiles/r4_conv2d_wino2.txt) -- the matrix pipe IS what its time follows.
//
//   2 x 2 outputs (rows y, y+1; columns x, x+1) from the 4 x 4 input patch:   T = B^T d B,  U = G g G^T (float64 on the host),
//   m[sh][sw] = T[sh][sw] . U[sh][sw] summed over the input channels,  Y = A^T m A:   16 products per 4 outputs instead of 36 -- 0.444
//   of the direct kernel's MFMA work (row-only form: 0.667); every product a v_mfma_f32_16x16x4_f32 with fp32 accumulation.
//
// Structure = the row-only kernel's: 256-thread persistent workgroups (2 per CU), 8 x 16-pixel output tiles x 32 output channels per
// work item, the input in 32-channel chunks through a 2-slot LDS ring with ONE LDS-only barrier per chunk, the next chunk (or the next
// item's first) prefetched into registers during the MFMAs; the ROW transform is applied by the 144 loader threads when a chunk enters
// LDS (brick columns in registers).  New:
//   * the COLUMN transform runs in registers between LDS and the MFMA, as the row transform of csrc/conv3d_wino2.hip does: an MFMA
//     column is a 2 x 2 output BLOCK; a lane reads the four brick columns 2cb .. 2cb+3 of its block's transformed row (4 ds_read_b128 per
//     16 channels) and forms s0 = c0 - c2, s1 = c1 + c2, s2 = c2 - c1, s3 = c1 - c3: one packed add per two operands, 16 MFMAs per step;
//   * wave (rpp, nt) owns the 16 blocks of two row pairs (2 x 8 column blocks = 4 x 16 pixels) x ONE 16-channel output tile with all 16
//     products m[sh][sw] (64 accumulator registers): no cross-wave reduction, the whole 2 x 2 output transform happens in the wave;
//   * transposed MFMAs (weights as A, blocks as B): a lane ends with four consecutive output channels of its block's four pixels --
//     four 16-byte stores (and residual loads) per work item instead of eight 8-byte ones;
//   * LDS slot = 16 transformed rows (sh-major: row = 4 sh + row pair) of 19 voxels (18 + 1 pad: an ODD pitch puts the two row pairs of
//     a wave in different halves of a 256-byte bank row) x 128 B; 16-byte chunk c of column x at chunk position c ^ ((x >> 1) & 7) -- the
//     key depends on the column only, so sh and the channel half are immediates.  MFMA column i <-> block: i in {0-3, 12-15} -> first
//     row pair, column blocks 0..7; i in {4-11} -> second row pair: the hardware's ds_read_b128 lane groups ({0-3,12-15,20-27}, ...)
//     then hold eight same-parity voxels with eight distinct chunk positions and eight of the other parity: conflict-free.
//   * weights: [group of 32 channels][chunk][8 steps = (sh, channel half)][4 sw][2 output tiles][64 lanes][4] (packing.py::pack_conv2d_wino2),
//     streamed from L2 two steps ahead, continuously across chunks and work items.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "estd_hip.h"
#include "estd_common.h"

#ifndef ESTD_C2W2ABL
#define ESTD_C2W2ABL 0   // timing ablations only (wrong results when != 0): 1 no output stores, 2 no transform writes, 8 no weight stream, 128 no column transform
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;

constexpr int TH = 8, TW = 16;
constexpr int TROWS = 16;                            // 4 transforms x 4 row pairs
constexpr int NSTEP = 8;                             // steps per 32-channel chunk: (sh, channel half)
#ifndef ESTD_C2W2_PFS
#define ESTD_C2W2_PFS 8
#endif
#ifndef ESTD_C2W2_WD
#define ESTD_C2W2_WD 2
#endif
// ESTD_C2W2_INLOOP = 1: the row transform + LDS write of the NEXT chunk inside the step loop of the current one (steps INLOOP_W0 .. NSTEP - 1, the
// brick's rows requested in the first INLOOP_PFS steps), as csrc/conv3d_wino2.hip rewrites its slices inside the tap loop: the other slot of
// the ring was last read in the previous chunk, which every wave has left once it passed this chunk's barrier -- so the write is legal there,
// and the top of a chunk is the barrier alone instead of "wait for the brick, transform, write, barrier" with three quarters of the threads idle.
#ifndef ESTD_C2W2_INLOOP
#define ESTD_C2W2_INLOOP 1
#endif
#ifndef ESTD_C2W2_INLOOP_PFS
#define ESTD_C2W2_INLOOP_PFS 4
#endif
#ifndef ESTD_C2W2_INLOOP_W0
#define ESTD_C2W2_INLOOP_W0 4
#endif
// ESTD_C2W2_SCHED = 1 (round 5): the step's memory requests (weights two steps ahead, the next brick's rows, the in-loop LDS writes of the next chunk) are issued
// BETWEEN its MFMAs -- one vector-memory read, one LDS write and up to four VALU instructions per pair of MFMAs -- instead of in a cluster in front of them (a 16-byte
// store costs the LDS store path 13 cycles, a vector-memory request ~16 of issue: a cluster of 5 + 4 of them in front of 16 x 32 matrix cycles left the pipe idle
// while it was issued; same finding as csrc/conv3d_wino2x.hip, profiles/r5_wino2x_table.txt)
#ifndef ESTD_C2W2_SCHED
#define ESTD_C2W2_SCHED 2
#endif
constexpr int WD = ESTD_C2W2_WD, WR = 4;             // weight stream: WD steps ahead, ring slot = step % WR (8 % WR == 0, WD < WR)

__device__ __forceinline__ float4 as_float4(u32x4 v) { float4 f; __builtin_memcpy(&f, &v, 16); return f; }
__device__ __forceinline__ u32x4 as_u32x4(float4 f) { u32x4 v; __builtin_memcpy(&v, &f, 16); return v; }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, size_t elems)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(elems * 4), 0x00020000);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// DIL = 2 (PSM layer4, networks/psm_submodule.py:58): pixels of equal parity form the F(2,3) sequences on both axes -- a block is the
// outputs (a, a+2) x (b, b+2) with a in {0,1,4,5}, b in {0,1,4,5,8,9,12,13} of the tile, fed by brick rows a, a+2, a+4, a+6 and columns b,
// b+2, b+4, b+6 of a 12 x 20 brick.  The pitch stays 20 (two slots of 16 x 20 x 128 B, two workgroups per CU = the whole LDS): the eight
// column blocks of a row pair already alternate between the halves of a bank row (b is even / odd), and the two row pairs of a
// ds_read_b128 lane group differ in bit 0 of the lane group index, i.e. of the chunk position.
template <int DIL>
__global__ __launch_bounds__(256, 2) void conv2d_wino2_kernel(const estd_conv2d_desc p, int tiles_w, int tiles_h, int total_items)
{
    constexpr int IN_H = TH + 2 * DIL, IN_W = TW + 2 * DIL;      // haloed brick: 10 x 18 | 12 x 20
    constexpr bool INLOOP = ESTD_C2W2_INLOOP != 0 && (DIL == 1 || ESTD_C2W2_INLOOP == 2);     // (the dilation-2 instance spills with it: 256 VGPRs, 9 spilled, +4 %)
    constexpr int PITCH = DIL == 1 ? IN_W + 1 : IN_W;            // voxels per transformed row: 19 (odd, see the header) | 20
    constexpr int SLOT_BYTES = TROWS * PITCH * 128;              // 38 912 | 40 960
    constexpr int SH_BYTES = 4 * PITCH * 128;                    // one row transform index further
    constexpr int LOADERS = IN_W * 8;                            // 144 | 160 threads hold the brick: (column, 16-byte chunk)
    auto lds_off = [](int row, int col, int chunk) { return (row * PITCH + col) * 128 + ((chunk ^ ((col >> 1) & 7)) << 4); };
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rpp = wave >> 1, nt = wave & 1;          // row pairs 2 rpp, 2 rpp + 1; output channels 16 nt .. 16 nt + 15 of the item's 32
    const int g = lane >> 4, i = lane & 15;
    const int blk_rp = (i >= 4 && i < 12) ? 1 : 0;     // MFMA column <-> block (see the header)
    const int cb = i < 4 ? i : i < 12 ? i - 4 : i - 8;
    const int rp = 2 * rpp + blk_rp;
    const int row_a = DIL == 1 ? 2 * rp : (rp & 1) + 4 * (rp >> 1);      // first tile row of the block's row pair (rows row_a, row_a + DIL)
    const int col_b = DIL == 1 ? 2 * cb : (cb & 1) + 4 * (cb >> 1);      // first tile column of the block (columns col_b, col_b + DIL)
    const int H = p.H, W = p.W, Cin = p.cin, Cout = p.cout;
    const int nchunks = Cin >> 5;
    const int tiles_per_group = p.N * tiles_h * tiles_w;
    const int wlane = lane * 16 + nt * 1024;

    int u, u_end;
    {
        const int G = gridDim.x, bid = blockIdx.x;
        const int r = ((G & 7) == 0) ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;
        u = (int)((long long)total_items * r / G);
        u_end = (int)((long long)total_items * (r + 1) / G);
    }
    if (u >= u_end) return;

    const size_t img_in = (size_t)H * W * Cin, img_out = (size_t)H * W * Cout;
    const bool loader = tid < LOADERS;
    const int lzx = tid >> 3, lc = tid & 7;           // this loader thread's brick column and 16-byte chunk
    const int wbase = loader ? lds_off(0, lzx, lc) : 0;   // LDS offset of (row 0, its column, its chunk); row r: + r * PITCH * 128

    // per-lane read offsets of the block's four brick columns (row transform 0, slot 0); the second channel half is 64 bytes away in the
    // swizzled chunk order (chunk + 4 = chunk position ^ 4)
    int aoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) aoff[q] = lds_off(rp, col_b + DIL * q, g);

    auto decode = [&](int item, int& grp, int& n, int& th0, int& tw0) {
        grp = item / tiles_per_group;
        int t = item - grp * tiles_per_group;
        const int twi = t % tiles_w; t /= tiles_w;
        const int thi = t % tiles_h; n = t / tiles_h;
        th0 = thi * TH; tw0 = twi * TW;
    };
    auto brick_offsets = [&](int th0, int tw0, unsigned (&voff)[IN_H], bool enable) {
        const int gx = tw0 - DIL + lzx;
        const bool colok = enable && loader && (unsigned)gx < (unsigned)W;
        const unsigned base = (unsigned)(gx * Cin + lc * 4) * 4u;
        const unsigned rowbytes = (unsigned)(W * Cin) * 4u;
#pragma unroll
        for (int zy = 0; zy < IN_H; ++zy) {
            const int gy = th0 - DIL + zy;                                   // uniform
            const bool rowok = (unsigned)gy < (unsigned)H;                   // uniform
            voff[zy] = (colok && rowok) ? base + (unsigned)gy * rowbytes : OOB_OFFSET;
        }
    };

    int grp, n, th0, tw0;
    decode(u, grp, n, th0, tw0);
    unsigned voff[IN_H];
    brick_offsets(th0, tw0, voff, true);
    __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in + (size_t)n * img_in, img_in);
    float4 pf[IN_H];
#pragma unroll
    for (int zy = 0; zy < IN_H; ++zy) pf[zy] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[zy], 0, 0));

    // weight stream: one step = the four sw taps of (sh, channel half) = 4 KB per output tile; WD steps ahead, continuous across chunks / items
    const size_t wgrp_elems = (size_t)nchunks * NSTEP * 4 * 2 * 256;          // packed floats per group of 32 output channels
    float4 bq[WR][4];
    {
        const __amdgpu_buffer_rsrc_t rs_w0 = make_rsrc(p.w_wino + (size_t)grp * wgrp_elems, wgrp_elems);
#pragma unroll
        for (int s = 0; s < WD; ++s)
#pragma unroll
            for (int sw = 0; sw < 4; ++sw) bq[s][sw] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w0, wlane, (s * 4 + sw) * 2048, 0));
    }
    const float floor_b = p.relu_before_residual ? 0.f : ESTD_NO_FLOOR;
    const float floor_a = p.relu_after_residual ? 0.f : ESTD_NO_FLOOR;

    int k = 0;                                              // global chunk counter -> LDS slot
    if (INLOOP) {                                           // the first brick of this workgroup: transformed into slot 0 here, every later one in a step loop
        if (loader && !(ESTD_C2W2ABL & 2)) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int a = DIL == 1 ? 2 * w : (w & 1) + 4 * (w >> 1);
                const float4 d0 = pf[a], d1 = pf[a + DIL], d2 = pf[a + 2 * DIL], d3 = pf[a + 3 * DIL];
                *reinterpret_cast<float4*>(smem + wbase + (0 * 4 + w) * PITCH * 128) = f4_sub(d0, d2);
                *reinterpret_cast<float4*>(smem + wbase + (1 * 4 + w) * PITCH * 128) = f4_add(d1, d2);
                *reinterpret_cast<float4*>(smem + wbase + (2 * 4 + w) * PITCH * 128) = f4_sub(d2, d1);
                *reinterpret_cast<float4*>(smem + wbase + (3 * 4 + w) * PITCH * 128) = f4_sub(d1, d3);
            }
        }
    }
    while (true) {
        const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w_wino + (size_t)grp * wgrp_elems, wgrp_elems);
        // folded BatchNorm of this lane's four channels, requested an item ahead of the epilogue that uses it
        const int cbase = grp * 32 + nt * 16 + 4 * g;
        const float4 sc4 = *reinterpret_cast<const float4*>(p.scale + cbase), sh4 = *reinterpret_cast<const float4*>(p.shift + cbase);
        f32x4 acc[4][4];                                    // m[sh][sw]; the first chunk's first product per accumulator takes C = 0

        const bool has_next_item = (u + 1 < u_end);
        int ngrp = grp, nn_ = n, nth0 = th0, ntw0 = tw0;
        if (has_next_item) decode(u + 1, ngrp, nn_, nth0, ntw0);
        const __amdgpu_buffer_rsrc_t rs_wni = make_rsrc(p.w_wino + (size_t)ngrp * wgrp_elems, wgrp_elems);

        // row transform of row pairs w0 .. w1 - 1 of the brick in pf, written to ``slot`` (row = 4 sh + row pair)
        auto write_rows = [&](char* slot, int w0, int w1) {
            if (loader && !(ESTD_C2W2ABL & 2)) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    if (w < w0 || w >= w1) continue;
                    const int a = DIL == 1 ? 2 * w : (w & 1) + 4 * (w >> 1);         // compile-time after unrolling
                    const float4 d0 = pf[a], d1 = pf[a + DIL], d2 = pf[a + 2 * DIL], d3 = pf[a + 3 * DIL];
                    *reinterpret_cast<float4*>(slot + wbase + (0 * 4 + w) * PITCH * 128) = f4_sub(d0, d2);
                    *reinterpret_cast<float4*>(slot + wbase + (1 * 4 + w) * PITCH * 128) = f4_add(d1, d2);
                    *reinterpret_cast<float4*>(slot + wbase + (2 * 4 + w) * PITCH * 128) = f4_sub(d2, d1);
                    *reinterpret_cast<float4*>(slot + wbase + (3 * 4 + w) * PITCH * 128) = f4_sub(d1, d3);
                }
            }
        };
        // one of the four transformed rows (index sh) of row pair w: a quarter of write_rows(slot, w, w + 1)
        auto write_row_sh = [&](char* slot, int w, int sh_) {
            if (loader && !(ESTD_C2W2ABL & 2)) {
                const int a = DIL == 1 ? 2 * w : (w & 1) + 4 * (w >> 1);
                const float4 d0 = pf[a], d1 = pf[a + DIL], d2 = pf[a + 2 * DIL], d3 = pf[a + 3 * DIL];
                const float4 v = sh_ == 0 ? f4_sub(d0, d2) : sh_ == 1 ? f4_add(d1, d2) : sh_ == 2 ? f4_sub(d2, d1) : f4_sub(d1, d3);
                *reinterpret_cast<float4*>(slot + wbase + (sh_ * 4 + w) * PITCH * 128) = v;
            }
        };
        auto chunk_body = [&](auto first_c, const int c) {
            constexpr bool FIRST = decltype(first_c)::value;
            const int sb = (k & 1) * SLOT_BYTES;
            char* slot = smem + sb;
            // ---- row transform B^T d of the brick columns, straight into the slot: row = 4 sh + row pair ----
            if (!INLOOP) write_rows(slot, 0, 4);
            lds_barrier();

            // what to prefetch while this chunk computes: the next 32-channel chunk of the same pixels, or chunk 0 of the next item
            const bool last_chunk = (c + 1 == nchunks);
            int pf_soff = (c + 1) * 128;
            if (last_chunk) {
                brick_offsets(nth0, ntw0, voff, has_next_item);
                if (nn_ != n) rs_in = make_rsrc(p.in + (size_t)nn_ * img_in, img_in);
                pf_soff = 0;
            }
            const int wbase_q = c * NSTEP * 4;                              // tap quads (2 KB each) in front of this chunk
            const __amdgpu_buffer_rsrc_t rs_wx = last_chunk ? rs_wni : rs_w; // where the stream goes on after this chunk
            const int wnext_q = last_chunk ? 0 : wbase_q + NSTEP * 4;

            int ao[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) ao[q] = aoff[q] + sb;
            auto load_rows = [&](int st, float4 (&Rr)[4]) {                 // the four brick columns of the block at (sh = st >> 1, half st & 1)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    Rr[q] = *reinterpret_cast<const float4*>(smem + ((st & 1) ? (ao[q] ^ 64) : ao[q]) + (st >> 1) * SH_BYTES);
            };
            auto xform2 = [&](const float4 (&Rr)[4], int h, f32x2 (&o)[4]) {
                f32x2 r[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) r[q] = h == 0 ? (f32x2){Rr[q].x, Rr[q].y} : (f32x2){Rr[q].z, Rr[q].w};
                if (ESTD_C2W2ABL & 128) { o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; }
                else { o[0] = r[0] - r[2]; o[1] = r[1] + r[2]; o[2] = r[2] - r[1]; o[3] = r[1] - r[3]; }
            };
            f32x2 T[2][4];
            float4 R[4];
            load_rows(0, R);
            xform2(R, 0, T[0]);
            xform2(R, 1, T[1]);
            load_rows(1, R);
            __builtin_amdgcn_sched_barrier(0);

#pragma clang loop unroll(full)
            for (int st = 0; st < NSTEP; ++st) {
                const int sh = st >> 1;
                f32x2 Tn[2][4];
                if (ESTD_C2W2_SCHED == 2) {
                    // the step in four quarters of 4 MFMAs (128 matrix cycles each): in front of quarter k one weight request (tap sw = k of step st + WD), a share of the
                    // next brick's rows and ONE 16-byte LDS write (+ its four adds) of the next chunk's transform; scheduling barriers pin the quarters
                    constexpr int PFS = INLOOP ? ESTD_C2W2_INLOOP_PFS : ESTD_C2W2_PFS;
                    const int r0 = PFS != 8 ? (st < PFS ? st * IN_H / PFS : IN_H) : DIL == 1 ? (st == 0 ? 0 : st + 1) : (3 * st + 1) / 2;
                    const int r1 = PFS != 8 ? (st < PFS ? (st + 1) * IN_H / PFS : IN_H)
                                            : DIL == 1 ? ((st == 0 || st == NSTEP - 1) ? r0 + 2 : r0 + 1) : (3 * (st + 1) + 1) / 2;
                    const bool wr = INLOOP && st >= ESTD_C2W2_INLOOP_W0 && (!last_chunk || has_next_item);
                    constexpr int WN = NSTEP - ESTD_C2W2_INLOOP_W0;
                    static_assert(!INLOOP || WN == 4, "quarter schedule: one row pair of the next chunk per step");
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            const int qk = 2 * h + e;
                            if (!(ESTD_C2W2ABL & 8)) {
                                const int tgt = st + WD;
                                bq[tgt % WR][qk] = tgt < NSTEP
                                    ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, (wbase_q + tgt * 4 + qk) * 2048, 0))
                                    : as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_wx, wlane, (wnext_q + (tgt - NSTEP) * 4 + qk) * 2048, 0));
                            }
#pragma unroll
                            for (int r = r0 + (r1 - r0) * qk / 4; r < r0 + (r1 - r0) * (qk + 1) / 4; ++r)
                                pf[r] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[r], pf_soff, 0));
                            if (wr) write_row_sh(smem + (((k + 1) & 1) * SLOT_BYTES), st - ESTD_C2W2_INLOOP_W0, qk);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int sw = 0; sw < 4; ++sw) {
                                const float4 b4 = bq[(ESTD_C2W2ABL & 8) ? 0 : st % WR][sw];
                                const float b = h == 0 ? (e == 0 ? b4.x : b4.y) : (e == 0 ? b4.z : b4.w);
                                const bool first_product = FIRST && (st & 1) == 0 && h == 0 && e == 0;
                                const f32x4 c_in = first_product ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[sh][sw];
                                acc[sh][sw] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, T[h][sw][e], c_in, 0, 0, 0);
                            }
                            if (e == 1 && st + 1 < NSTEP) xform2(R, h, Tn[h]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else {
                // weights of step st + WD (this chunk's, or the first steps of the next chunk / next item's first chunk)
                if (!(ESTD_C2W2ABL & 8)) {
                    const int tgt = st + WD;
#pragma unroll
                    for (int sw = 0; sw < 4; ++sw)
                        bq[tgt % WR][sw] = tgt < NSTEP
                            ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, (wbase_q + tgt * 4 + sw) * 2048, 0))
                            : as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_wx, wlane, (wnext_q + (tgt - NSTEP) * 4 + sw) * 2048, 0));
                }
                // the next brick over the eight steps: rows 0,1 | 2 | 3 | 4 | 5 | 6 | 7 | 8,9 (dilation 2: 0,1 | 2 | 3,4 | 5 | 6,7 | 8 | 9,10 | 11)
                {
                    constexpr int PFS = INLOOP ? ESTD_C2W2_INLOOP_PFS : ESTD_C2W2_PFS;      // the brick's rows go out over the first PFS steps (8: the spread above)
                    const int r0 = PFS != 8 ? (st < PFS ? st * IN_H / PFS : IN_H) : DIL == 1 ? (st == 0 ? 0 : st + 1) : (3 * st + 1) / 2;
                    const int r1 = PFS != 8 ? (st < PFS ? (st + 1) * IN_H / PFS : IN_H)
                                            : DIL == 1 ? ((st == 0 || st == NSTEP - 1) ? r0 + 2 : r0 + 1) : (3 * (st + 1) + 1) / 2;
#pragma unroll
                    for (int r = r0; r < r1; ++r) pf[r] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[r], pf_soff, 0));
                }
                if (INLOOP && st >= ESTD_C2W2_INLOOP_W0 && (!last_chunk || has_next_item)) {
                    // the next chunk's (or the next item's first) brick -> the other slot; its row pairs spread over the last steps
                    constexpr int WN = NSTEP - ESTD_C2W2_INLOOP_W0;
                    const int i0 = (st - ESTD_C2W2_INLOOP_W0) * 4 / WN, i1 = (st - ESTD_C2W2_INLOOP_W0 + 1) * 4 / WN;
                    write_rows(smem + (((k + 1) & 1) * SLOT_BYTES), i0, i1);
                }
                if (!ESTD_C2W2_SCHED) __builtin_amdgcn_sched_barrier(0);      // memory requests in front of the step's MFMAs
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int sw = 0; sw < 4; ++sw) {
                            const float4 b4 = bq[(ESTD_C2W2ABL & 8) ? 0 : st % WR][sw];
                            const float b = h == 0 ? (e == 0 ? b4.x : b4.y) : (e == 0 ? b4.z : b4.w);
                            const bool first_product = FIRST && (st & 1) == 0 && h == 0 && e == 0;
                            const f32x4 c_in = first_product ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[sh][sw];
                            acc[sh][sw] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, T[h][sw][e], c_in, 0, 0, 0);
                        }
                    if (st + 1 < NSTEP) xform2(R, h, Tn[h]);
                }
                }
                if (st + 2 < NSTEP) load_rows(st + 2, R);
                if (ESTD_C2W2_SCHED == 2) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                } else if (ESTD_C2W2_SCHED) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // two MFMAs
                        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one vector-memory read (weights / brick rows)
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // one LDS write (the next chunk's transformed rows)
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // transform arithmetic
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {            // order of the region: the MFMAs of two k-steps, then the next step's 4 transforms
                        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                }
                if (st + 1 < NSTEP) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int sw = 0; sw < 4; ++sw) T[h][sw] = Tn[h][sw];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        chunk_body(std::true_type{}, 0);
        ++k;
        for (int c = 1; c < nchunks; ++c, ++k) chunk_body(std::false_type{}, c);

        // ---- output transform A^T m A, then the epilogue: lane (g, i) = channels cbase .. cbase + 3 of the four pixels of block i ----
        auto epilogue = [&](auto has_res_c) {
            constexpr bool HAS_RES = decltype(has_res_c)::value;
            const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(p.out + (size_t)n * img_out, img_out);
            const __amdgpu_buffer_rsrc_t rs_res = make_rsrc((HAS_RES ? p.residual : p.out) + (size_t)n * img_out, img_out);
            const float floor_1 = HAS_RES ? floor_b : fmaxf(floor_b, floor_a);
            unsigned eo[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int y = th0 + row_a + DIL * r, x = tw0 + col_b + DIL * j;
                    eo[r][j] = (y < H && x < W) ? (unsigned)((y * W + x) * Cout + cbase) * 4u : OOB_OFFSET;
                }
            float4 res[2][2];
            if (HAS_RES) {
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int j = 0; j < 2; ++j) res[r][j] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_res, eo[r][j], 0, 0));
            }
            f32x4 z[4][2];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                z[s][0] = acc[s][0] + acc[s][1] + acc[s][2];
                z[s][1] = acc[s][1] - acc[s][2] - acc[s][3];
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f32x4 yv = r == 0 ? z[0][j] + z[1][j] + z[2][j] : z[1][j] - z[2][j] - z[3][j];
                    float4 v;
                    v.x = fmaxf(fmaf(yv[0], sc4.x, sh4.x), floor_1); v.y = fmaxf(fmaf(yv[1], sc4.y, sh4.y), floor_1);
                    v.z = fmaxf(fmaf(yv[2], sc4.z, sh4.z), floor_1); v.w = fmaxf(fmaf(yv[3], sc4.w, sh4.w), floor_1);
                    if (HAS_RES) {
                        v.x = fmaxf(v.x + res[r][j].x, floor_a); v.y = fmaxf(v.y + res[r][j].y, floor_a);
                        v.z = fmaxf(v.z + res[r][j].z, floor_a); v.w = fmaxf(v.w + res[r][j].w, floor_a);
                    }
                    if (ESTD_C2W2ABL & 1) asm volatile("" :: "v"(v.x), "v"(v.w));
                    else __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(v), rs_out, eo[r][j], 0, 0);
                }
        };
        if (p.residual) epilogue(std::true_type{}); else epilogue(std::false_type{});

        if (!has_next_item) break;
        ++u;
        grp = ngrp; n = nn_; th0 = nth0; tw0 = ntw0;
    }
}

}  // namespace

extern "C" int estd_conv2d_k3_wino2(const estd_conv2d_desc* dp, estd_stream_t s)
{
    if (!dp) return ESTD_ERR_ARG;
    const estd_conv2d_desc& d = *dp;
    if (d.N <= 0 || d.H <= 0 || d.W <= 0 || !d.in || !d.w_wino || !d.scale || !d.shift || !d.out) return ESTD_ERR_ARG;
    if (d.cin < 32 || (d.cin & 31) || d.cout < 32 || (d.cout & 31)) return ESTD_ERR_ARG;
    if (d.dilation != 1 && d.dilation != 2) return ESTD_ERR_UNSUPPORTED;
    const long long widest = (long long)d.H * d.W * (d.cin > d.cout ? d.cin : d.cout) * 4;
    if (widest >= 0x7fffff00LL) return ESTD_ERR_UNSUPPORTED;
    const int tiles_w = (d.W + TW - 1) / TW, tiles_h = (d.H + TH - 1) / TH;
    const int groups = d.cout / 32;
    const long long total = (long long)groups * d.N * tiles_h * tiles_w;
    if (total > 0x7fffffffLL) return ESTD_ERR_ARG;
    const int slots = estd_persistent_wgs(2);
    int grid = total < slots ? (int)total : slots;
    if (grid >= 8) grid &= ~7;
    if (d.dilation == 1) {
        const size_t lds = (size_t)2 * TROWS * (TW + 3) * 128;          // 76 KB: two workgroups per CU
        estd_allow_dynamic_lds<conv2d_wino2_kernel<1>>((int)lds);
        hipLaunchKernelGGL(conv2d_wino2_kernel<1>, dim3(grid), dim3(256), lds, estd_stream(s), d, tiles_w, tiles_h, (int)total);
    } else {
        const size_t lds = (size_t)2 * TROWS * (TW + 4) * 128;          // 80 KB: two workgroups per CU = all of the LDS
        estd_allow_dynamic_lds<conv2d_wino2_kernel<2>>((int)lds);
        hipLaunchKernelGGL(conv2d_wino2_kernel<2>, dim3(grid), dim3(256), lds, estd_stream(s), d, tiles_w, tiles_h, (int)total);
    }
    return ESTD_LAUNCH_CHECK();
}
