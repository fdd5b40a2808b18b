// conv3d_wino3.hip -- the 32 -> 32 (and 33 -> 32) 3x3x3 convolution with ALL THREE axes in Winograd F(2,3) form on gfx950 fp32 MFMA:
// F(2x2x2, 3x3x3), 64 products per 8 outputs = 8/27 of the direct MFMA work (two axes, csrc/conv3d_wino2.hip: 12/27).
//
// Same operator and descriptor as estd_conv3d_k3_wino2 (networks/layers_op.py:16-39 as used at hybrid_models/model_hybrid.py:59-60,:95 and
// hybrid_models/hybrid_depth_decoder.py:84-95).
//
//   2 x 2 x 2 outputs (planes d, d+1; rows y, y+1; columns x, x+1) from the 4 x 4 x 4 input patch:
//       T = B^T x B on the depth axis, the row axis and the column axis   (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1])
//       U = G g G^T on kd, kh, kw                                         (float64 on the host, rounded once: packing.pack_conv3d_wino3)
//       m[sd][sh][sw] = sum over input channels of U[sd][sh][sw] * T[sd][sh][sw]      64 products, each a chain of v_mfma_f32_16x16x4_f32
//       y = A^T m A on the three axes                                     (A^T = [1 1 1 0; 0 1 -1 -1])
//
// Work decomposition (what it shares with the two-axis kernel: 512 threads = 8 waves, ONE workgroup per CU, output tile 2 x 8 x 16 voxels, persistent
// XCD-contiguous ranges of the column-major tile list, the four DEPTH-transformed slices of 10 x 18 voxels x 32 channels in LDS, raw planes carried in
// registers along a depth column, the next tile's slices written inside the tap loop):
//   * an MFMA column is a 2 x 2 output BLOCK (row pair, column pair) of the tile instead of a voxel; wave (rq, nh, shh) owns the 16 blocks of tile rows
//     4rq .. 4rq+3 (two row pairs x eight column pairs) x the 16 output channels of half nh x the row-transform indices sh = 2shh, 2shh+1:
//     8 products m[sh][sw] of the CURRENT depth transform (32 accumulator registers), 256 MFMAs per tile and wave (two-axis kernel: 384);
//   * the ROW and the COLUMN transform run in registers between LDS and the MFMA: a half-sub-step (sd, channel chunk, channel pair, sh) reads two halo rows x four
//     columns of its block, 8 bytes = one channel pair each (the row the two sh of a wave share stays in registers: 12 ds_read_b64 per two half-sub-steps), forms the
//     row combination (4 packed adds) and its four column combinations (4 more), and multiplies them with the four taps sw of (sd, sh): 8 MFMAs, 2 VALU per MFMA.
//     (A 16-MFMA unit on 16-byte reads needs 56 registers more: it spilled 128.)  Lane groups g, g + 1 read opposite 8-byte halves of their 16-byte slot -- the weights
//     are packed for that k order -- so the 32 lanes of a ds_read_b64 pass touch 32 different 8-byte units of a bank row;
//   * when a depth transform is complete its 8 products go through the column and row halves of the output transform and are added into the partial sums
//     of the two output planes (32 registers); the two waves of a SIMD (shh = 0, 1: the two halves of the row transform) exchange one plane's partial sums
//     through LDS at the end of the tile and finish one plane each (the structure of the two-axis kernel's 32 -> 16 instance);
//   * LDS lines: line (halo row r, slice sd) = 18 records + 16 bytes at (4r + sd) * 2320 -- the four slices of a row are neighbours (every patch address is an
//     immediate on one per-lane base) and 8 lines = 128 (mod 256): halo rows two apart (the two row pairs of a wave) lie in opposite halves of a 256-byte bank row; with
//     the column-keyed chunk swizzle and MFMA column i <-> block (row pair = 4 <= i < 12, column pair = i < 4 ? i : i < 12 ? i - 4 : i - 8) every fragment read is conflict-free;
//   * NO BRANCH inside the unrolled tap loop (scheduling regions end at branches: with them the allocator spilled 150 registers): absent planes read through a null
//     buffer descriptor, threads without a third chunk of a slice write into a dummy LDS slot, a segment's first tile "stores" its deferred epilogue through the null descriptor.
// profiles/r5_wino3_table.txt: 0.65 ms against 0.81 ms (two-axis kernel) for 3 volumes of 64x120x160, error against fp64 below the two-axis kernel's.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "estd_hip.h"
#include "estd_common.h"

#ifndef ESTD_W3BD
#define ESTD_W3BD 4         // weight buffers in flight (half-sub-steps of 8 MFMAs): 4 = requested three half-sub-steps (~800 cycles) ahead
#endif
#ifndef ESTD_W3_SCHED
#define ESTD_W3_SCHED 1     // issue order of a half-sub-step: 0 = requests, then {4 MFMAs, 4 transforms} x 2, then the fragment reads; 1 = {2 MFMAs, transforms, 2 fragment
#endif                      // reads, every other time one weight request} x 4 (what paid in csrc/conv2d_wino2.hip: a cluster of requests drains the matrix pipe)
#ifndef ESTD_W3_WRING
#define ESTD_W3_WRING 0     // (A/B; slower: 0.69 vs 0.653 ms, the ring's registers spill at the tile boundary) 1: the weight ring runs on across the tiles of a column segment (the last half-sub-steps of a tile request the next tile's first blocks)
#endif
#ifndef ESTD_W3_RBQ
#define ESTD_W3_RBQ 5       // half-sub-step at which the deferred epilogue of a read-back instance consumes its loads (requested at 0, 1)
#endif
#ifndef ESTD_W3_TANH_HALVES
#define ESTD_W3_TANH_HALVES 1   // (A/B: 0 = per-element activation select in the tanh launches)
#endif
#ifndef ESTD_W3_PFQ
#define ESTD_W3_PFQ 4
#endif
#ifndef ESTD_W3_PFQ_RB
#define ESTD_W3_PFQ_RB 8
#endif
#ifndef ESTD_W3PRIO
#define ESTD_W3PRIO 0        // A/B: 1 = static priority 1 for the waves of the second row-transform half (the two waves of a SIMD run the same instruction stream)
#endif
#ifndef ESTD_W3PK
#define ESTD_W3PK 0          // A/B: the transforms as v_pk_add_f32 by inline assembly
#endif
#ifndef ESTD_W3_SCHED_VALU
#define ESTD_W3_SCHED_VALU 8
#endif
#ifndef ESTD_W3ABL
#define ESTD_W3ABL 0        // timing ablations only (results are wrong): 1 no output stores, 2 no slice writes, 8 no weight stream, 16 no next-plane prefetch,
#endif                      // 32 no fold, 128 no transforms

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));

constexpr int TH = 8, TW = 16;
constexpr int IN_H = TH + 2, IN_W = TW + 2;
constexpr int SL_VOX = IN_H * IN_W;                 // 180 records per input slice (with halo)
constexpr int SL_CHUNKS = SL_VOX * 8;               // 16-byte chunks per slice: 1440
// LDS lines: line (halo row r, depth slice sd) = 18 records + 16 bytes at (4r + sd) * LINE_BYTES -- the four slices of a row are neighbours, so
// every (slice, row) of a block's patch is a 16-bit immediate on ONE per-lane base register; 8 lines = 128 (mod 256): halo rows two apart (the two
// row pairs of a wave) lie in opposite halves of a 256-byte bank row
constexpr int LINE_BYTES = IN_W * 128 + 16;         // 2320
constexpr int ROW_BYTES = 4 * LINE_BYTES;           // 9280: next halo row, same slice
constexpr int SLICE_BYTES = LINE_BYTES;             // next slice, same halo row
constexpr int SLICES_BYTES = IN_H * ROW_BYTES;      // 92 800
constexpr int NTHREADS = 512, SIT = 3;              // chunks per thread and slice
constexpr int RED_BYTES = 512;
constexpr int SS_BYTES = 3 * 32 * 4;                // folded BN scale | shift | activation floor
constexpr int VTAB_BYTES = SIT * NTHREADS * 4;      // per-thread global offsets of the slice chunks
constexpr int XCH_BYTES = 8 * 4 * 64 * 16;          // [wave][4 quads][64 lanes] float4: one plane's partial sums per wave
constexpr int NTAPS = 64, TAP_BYTES = 4096;         // [block = ((4 sd + sh) * 2 + cc) * 2 + hh][2 halves nh][2 tap pairs][64 lanes][4]: packing.pack_conv3d_wino3
constexpr int DUMMY_BYTES = 96 * 16 + 2 * LINE_BYTES; // where the threads without a third chunk of a slice (1440 = 2 x 512 + 416) put their in-loop writes: the tap loop has no branch
constexpr int XSL_BYTES = 4 * SL_VOX * 4;            // EXTRA: the four depth-transformed slices of the scalar 33rd input channel ([sd][180] floats)
constexpr int XW_BYTES = 2 * 4 * 2 * 64 * 16;        // EXTRA: its weights [2 planes][4 sh][2 halves][64 lanes][4 sw] (packing.pack_conv3d_wino3_extra)
constexpr int LDS_BASE_BYTES = SLICES_BYTES + RED_BYTES + SS_BYTES + VTAB_BYTES + XCH_BYTES + DUMMY_BYTES;
// instances without a scalar channel: the weight blocks of half-sub-steps WL_Q0 .. WL_Q0 + 2 (both row-transform halves: 6 x 4 KB) live in LDS -- the requests a
// read-back instance would issue right behind its read-back loads (vector-memory loads return in order: they would wait for HBM with them)
#ifndef ESTD_W3_WLDS
#define ESTD_W3_WLDS 1
#endif
#ifndef ESTD_W3_WLQ0_PLAIN
#define ESTD_W3_WLQ0_PLAIN 4     // (A/B) launches without read-back streams: 0 = the FIRST three blocks of a tile instead -- measured the same (0.692 / 0.694 vs 0.690 / 0.689 ms): the other wave of the SIMD covers a tile's first L2 round trip
#endif
constexpr int WL_N = 3;
constexpr int WL_BYTES = 2 * WL_N * 4096;
constexpr int LDS_BYTES_EXTRA = LDS_BASE_BYTES + XSL_BYTES + XW_BYTES, LDS_BYTES_PLAIN = LDS_BASE_BYTES + (ESTD_W3_WLDS ? WL_BYTES : 0);
static_assert(LDS_BYTES_EXTRA <= 160 * 1024 && LDS_BYTES_PLAIN <= 160 * 1024, "LDS budget");
constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;        // beyond num_records of any descriptor: loads return 0, stores are dropped

__device__ __forceinline__ float4 as_float4(u32x4 v)
{
    float4 f;
    __builtin_memcpy(&f, &v, 16);
    return f;
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, size_t elems)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(elems * 4), 0x00020000);
}
// sum of a double over the 16 lanes of a DPP row, every lane gets the total (as in csrc/conv3d_wino2.hip)
template <int CTRL>
__device__ __forceinline__ double dpp_add_f64(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, 0xf, 0xf, false);
    return v + __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double row16_sum_f64(double v)
{
    v = dpp_add_f64<0xB1>(v);      // quad_perm [1,0,3,2]
    v = dpp_add_f64<0x4E>(v);      // quad_perm [2,3,0,1]
    v = dpp_add_f64<0x124>(v);     // row_ror:4
    v = dpp_add_f64<0x128>(v);     // row_ror:8
    return v;
}
// workgroup barrier that only orders LDS traffic (no vmcnt drain: prefetches and output stores stay in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int lds_colkey_off(int col, int c) { return col * 128 + ((c ^ ((col >> 1) & 7)) << 4); }
__device__ __forceinline__ float tanh_fast(float x)
{
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}
__device__ __forceinline__ float act_apply(float v, int act)
{
    if (act == ESTD_ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == ESTD_ACT_TANH) return tanh_fast(v);
    return v;
}
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// RBK: read-back streams of the epilogue.  0 none; 1 = running sum only (out += result); 2 = residual (+ residual2) + scale; 3 = both.
// Absent streams of a kind (no residual2) read through a null descriptor (zeros, no memory access): the tap loop has no branch.
#ifndef ESTD_W3DEFER
#define ESTD_W3DEFER 1      // 1: the epilogue of tile k runs inside the first half-sub-steps of tile k + 1 (A/B: 0 = between the tiles)
#endif
// STATS: GroupNorm(1 group per channel half) partial sums of the raw outputs (the ConvGRU's gate convolution, transformer/epipolar_transformer.py:21): one
// more barrier per tile.
// EXTRA: a scalar 33rd INPUT channel (the key || value convolution, hybrid_depth_decoder.py:190-191 on cat[dres2 output]): its four depth-transformed slices
// in 2.9 KB of LDS, its weights in 16 KB.  At the TOP of a tile (registers are free there, and the first weight blocks of the main stream are still on their way)
// lane group g row- and column-transforms its block's patch of slice sd = g; per output plane ONE MFMA per (sh, sw) sums the four depth transforms with the plane's
// output-transform coefficients folded into the weights (plane 0: 1 1 1 0, plane 1: 0 1 -1 -1) -- 16 MFMAs per tile and wave -- and the products go through the column
// and row halves of the output transform straight into the partial sums P, which the depth transforms of the main channels then add to.
template <int RBK, bool STATS, bool EXTRA>
__global__ __launch_bounds__(NTHREADS, 1) void conv3d_wino3_kernel(const estd_conv3d_desc p, int tiles_w, int tiles_h, int dpairs, int total_tiles)
{
    constexpr bool RB_ACC = RBK == 1 || RBK == 3, RB_RES = RBK == 2 || RBK == 3;
    constexpr bool DEFER = ESTD_W3DEFER != 0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rq = wave & 1;            // tile rows 4rq .. 4rq+3
    const int nh = (wave >> 1) & 1;     // output channel half
    const int shh = wave >> 2;          // row-transform indices 2shh, 2shh+1; finishes output plane d0 + shh.  (waves w, w + 4 share a SIMD)
    const int g = lane >> 4;            // k index inside an MFMA
    const int i = lane & 15;            // MFMA column = block
    const int rpl = (i >= 4 && i < 12) ? 1 : 0;          // row pair of the block inside the wave's four rows
    const int cb = i < 4 ? i : i < 12 ? i - 4 : i - 8;   // column pair
    const int D = p.D, H = p.H, W = p.W;
    const int HW = H * W;
    const size_t vol = (size_t)D * HW;

    int u, u_end;
    {
        const int G = gridDim.x, bid = blockIdx.x;
        const int r = ((G & 7) == 0) ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;       // XCD x owns a contiguous block of ranges
        u = (int)((long long)total_tiles * r / G);
        u_end = (int)((long long)total_tiles * (r + 1) / G);
    }
    if (u >= u_end) return;

    float* lds_ss = reinterpret_cast<float*>(smem + SLICES_BYTES + RED_BYTES);
    if (tid < 64) lds_ss[tid] = tid < 32 ? p.scale[tid & 31] : p.shift[tid & 31];
    if (tid >= 64 && tid < 96) lds_ss[tid] = ((tid - 64) < p.act_split ? p.act_a : p.act_b) == ESTD_ACT_RELU ? 0.0f : ESTD_NO_FLOOR;
    const bool any_tanh = p.act_a == ESTD_ACT_TANH || p.act_b == ESTD_ACT_TANH;                     // uniform
    const bool tanh_halves = ESTD_W3_TANH_HALVES && any_tanh && (p.act_split & 15) == 0;            // uniform
    const int wave_act = __builtin_amdgcn_readfirstlane(16 * nh < p.act_split ? p.act_a : p.act_b);   // the activation of this wave's 16 output channels (tanh_halves)
    unsigned* lds_vt = reinterpret_cast<unsigned*>(smem + SLICES_BYTES + RED_BYTES + SS_BYTES);     // [it][thread]
    float4* lds_xch = reinterpret_cast<float4*>(smem + SLICES_BYTES + RED_BYTES + SS_BYTES + VTAB_BYTES);
    const __amdgpu_buffer_rsrc_t rs_null = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w_wino2), 0, 0, 0x00020000);     // num_records 0: loads return 0
    float* lds_x = reinterpret_cast<float*>(smem + SLICES_BYTES + RED_BYTES + SS_BYTES + VTAB_BYTES + XCH_BYTES + DUMMY_BYTES);      // [4][SL_VOX] (EXTRA)
    // EXTRA: the scalar channel's weights in LDS (copied once per workgroup; visible after the first tile's barriers)
    char* lds_wx = smem + SLICES_BYTES + RED_BYTES + SS_BYTES + VTAB_BYTES + XCH_BYTES + DUMMY_BYTES + XSL_BYTES;
    if (EXTRA) {
        for (int e = tid; e < XW_BYTES / 16; e += NTHREADS) reinterpret_cast<float4*>(lds_wx)[e] = reinterpret_cast<const float4*>(p.w_extra)[e];
    }
    constexpr bool WL = ESTD_W3_WLDS != 0 && !EXTRA;
    constexpr int WL_Q0 = RBK == 0 ? ESTD_W3_WLQ0_PLAIN : 4;
    char* lds_wl = smem + LDS_BASE_BYTES;                // [row-transform half][WL_N half-sub-steps][4096] (shares the scalar channel's place)
    if (WL) {
        for (int e = tid; e < WL_BYTES / 16; e += NTHREADS) {
            const int blk = e >> 8, q_ = WL_Q0 + blk % WL_N, sh_ = 2 * (blk / WL_N) + (q_ & 1);
            const int src = ((((q_ >> 3) * 4 + sh_) * 2 + ((q_ >> 2) & 1)) * 2 + ((q_ >> 1) & 1));
            reinterpret_cast<float4*>(lds_wl)[e] = reinterpret_cast<const float4*>(p.w_wino2)[src * 256 + (e & 255)];
        }
    }
    const int xbase = ((4 * rq + 2 * rpl) * IN_W + 2 * cb) * 4 + g * (SL_VOX * 4);       // byte offset of the block's patch origin in the scalar slice sd = g (the lane group's k index)

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w_wino2, (size_t)NTAPS * 1024);
    const int wlane = lane * 16 + nh * 2048;
    // byte offset of (halo row 4rq + 2rpl, halo column 2cb + j, chunk g + 4cc) of slice 0
    // + the channel pair of half-sub-step half hh: 8 bytes further for hh ^ (g & 1) -- lane groups g, g + 1 read opposite 8-byte halves of their
    // 16-byte slots, so the 32 lanes of a ds_read_b64 pass touch 32 different 8-byte units of a bank row
    int cbase[4][2][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) cbase[j][cc][hh] = (4 * rq + 2 * rpl) * ROW_BYTES + lds_colkey_off(2 * cb + j, g + 4 * cc) + 8 * (hh ^ (g & 1));

    constexpr int BD = ESTD_W3BD;
    static_assert(32 % BD == 0, "the weight ring runs on across tiles");
    if (ESTD_W3PRIO == 1 && shh != 0) __builtin_amdgcn_s_setprio(1);
    while (u < u_end) {
        // ---- column segment [u, seg_end): same (n, h-tile, w-tile), consecutive depth pairs ----
        const int col = u / dpairs;
        int dp = u - col * dpairs;
        const int twi = col % tiles_w, c2 = col / tiles_w;
        const int thi = c2 % tiles_h, n = c2 / tiles_h;
        const int tw0 = twi * TW, th0 = thi * TH;
        const int seg_end = min(u_end, (col + 1) * dpairs);

        const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in_main + (size_t)n * vol * p.in_stride, vol * p.in_stride);
        __amdgpu_buffer_rsrc_t rs_res = rs_in, rs_res2 = rs_in;
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(p.out_main + (size_t)n * vol * p.out_stride, vol * p.out_stride);
        rs_res = (RB_RES && p.residual) ? make_rsrc(p.residual + (size_t)n * vol * p.out_stride, vol * p.out_stride) : rs_null;
        rs_res2 = (RB_RES && p.residual2) ? make_rsrc(p.residual2 + (size_t)n * vol * p.out_stride, vol * p.out_stride) : rs_null;
        const __amdgpu_buffer_rsrc_t rs_ex = EXTRA ? make_rsrc(p.in_extra + (size_t)n * vol, vol) : rs_null;
        // EXTRA: thread t < 180 owns voxel t of the scalar channel's haloed slices
        unsigned xoff = OOB_OFFSET;
        if (EXTRA && tid < SL_VOX) {
            const int zy = tid / IN_W, zx = tid % IN_W;
            const int gy = th0 - 1 + zy, gx = tw0 - 1 + zx;
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) xoff = (unsigned)(gy * W + gx) * 4u;
        }
        auto load_x = [&](int pd) {
            return (unsigned)pd < (unsigned)D ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ex, xoff, pd * HW * 4, 0)) : 0.0f;
        };
        const int in_slice_bytes = HW * p.in_stride * 4;
        const int out_plane_bytes = HW * p.out_stride * 4;

        // per-thread slice chunks: chunk it of a slice = chunk tid + it * NTHREADS; global offsets in a per-thread LDS table (own entries only)
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int e = tid + it * NTHREADS;
            const int vs = e >> 3, c = e & 7;
            const int zy = vs / IN_W, zx = vs % IN_W;
            const int gy = th0 - 1 + zy, gx = tw0 - 1 + zx;
            const bool ok = e < SL_CHUNKS && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            lds_vt[it * NTHREADS + tid] = ok ? (unsigned)((gy * W + gx) * p.in_stride + c * 4) * 4u : OOB_OFFSET;
        }
        auto chunk_voff = [&](int it) {
            const int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            return lds_vt[it * NTHREADS + wave * 64 + l];
        };
        int loffk[SIT];
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int vs = (tid >> 3) + it * (NTHREADS / 8);
            loffk[it] = (vs / IN_W) * ROW_BYTES + lds_colkey_off(vs % IN_W, tid & 7);
        }
        const bool last_ok = tid + (SIT - 1) * NTHREADS < SL_CHUNKS;
        // in-loop writes: the same for every slice (slices 0..2 only: a thread without a third chunk writes into the dummy area, which also holds its offsets of sl = 1, 2)
        int loffw[SIT];
#pragma unroll
        for (int it = 0; it < SIT; ++it) loffw[it] = (it < SIT - 1 || last_ok) ? loffk[it] : (SLICES_BYTES + RED_BYTES + SS_BYTES + VTAB_BYTES + XCH_BYTES + (tid - 416) * 16);
        auto load_plane = [&](int pd, float4 (&dst)[SIT]) {
            const bool pv = (unsigned)pd < (unsigned)D;        // wave-uniform; planes outside the volume are zero padding
#pragma unroll
            for (int it = 0; it < SIT; ++it)
                dst[it] = pv ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, chunk_voff(it), pd * in_slice_bytes, 0)) : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        // lane (g, i) holds the four consecutive channels 16nh + 4g .. +3 of the 2 x 2 voxels of its block, in the plane its wave finishes
        unsigned eoff[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int y = th0 + 4 * rq + 2 * rpl + m, x = tw0 + 2 * cb + c;
                eoff[m][c] = (y < H && x < W) ? (unsigned)((y * W + x) * p.out_stride + 16 * nh + 4 * g) * 4u : OOB_OFFSET;
            }
        auto bn_act = [&](const f32x4& a, int chb, float4& v) {
            const float4 sc4 = *reinterpret_cast<const float4*>(lds_ss + chb), sh4 = *reinterpret_cast<const float4*>(lds_ss + 32 + chb);
            if (!any_tanh) {
                const float4 lo = *reinterpret_cast<const float4*>(lds_ss + 64 + chb);
                v.x = fmaxf(a[0] * sc4.x + sh4.x, lo.x);
                v.y = fmaxf(a[1] * sc4.y + sh4.y, lo.y);
                v.z = fmaxf(a[2] * sc4.z + sh4.z, lo.z);
                v.w = fmaxf(a[3] * sc4.w + sh4.w, lo.w);
                return;
            }
            if (tanh_halves) {     // the 16 channels of a wave share the activation (split is a multiple of 16): a wave-uniform choice, tanh from the exp / rcp units
                const float u0 = a[0] * sc4.x + sh4.x, u1 = a[1] * sc4.y + sh4.y, u2 = a[2] * sc4.z + sh4.z, u3 = a[3] * sc4.w + sh4.w;
                if (wave_act == ESTD_ACT_TANH) { v.x = tanh_fast(u0); v.y = tanh_fast(u1); v.z = tanh_fast(u2); v.w = tanh_fast(u3); }
                else if (wave_act == ESTD_ACT_RELU) { v.x = fmaxf(u0, 0.f); v.y = fmaxf(u1, 0.f); v.z = fmaxf(u2, 0.f); v.w = fmaxf(u3, 0.f); }
                else { v.x = u0; v.y = u1; v.z = u2; v.w = u3; }
                return;
            }
            v.x = act_apply(a[0] * sc4.x + sh4.x, chb + 0 < p.act_split ? p.act_a : p.act_b);
            v.y = act_apply(a[1] * sc4.y + sh4.y, chb + 1 < p.act_split ? p.act_a : p.act_b);
            v.z = act_apply(a[2] * sc4.z + sh4.z, chb + 2 < p.act_split ? p.act_a : p.act_b);
            v.w = act_apply(a[3] * sc4.w + sh4.w, chb + 3 < p.act_split ? p.act_a : p.act_b);
        };
        // epilogue of this wave's plane (2 x 2 voxels x 4 channels per lane) in two halves m = block row: the read-back loads of a half back to back,
        // waited for once.  The descriptors are arguments: the deferred form passes null descriptors when there is no previous tile.
        struct EpiLoads { float4 r1[2], r2[2], ro[2]; };
        auto epi_issue = [&](int m, int so, const __amdgpu_buffer_rsrc_t& q1, const __amdgpu_buffer_rsrc_t& q2, const __amdgpu_buffer_rsrc_t& qo, EpiLoads& L) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                if (RB_RES) L.r1[c] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(q1, eoff[m][c], so, 0));
                if (RB_RES) L.r2[c] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(q2, eoff[m][c], so, 0));
                if (RB_ACC) L.ro[c] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(qo, eoff[m][c], so, 0));
            }
        };
        auto epi_finish = [&](const f32x4 (&a)[2], int m, int so, const __amdgpu_buffer_rsrc_t& qo, const EpiLoads& L) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float4 v;
                bn_act(a[c], 16 * nh + 4 * g, v);
                if (RB_RES) {
                    v = f4_add(f4_add(v, L.r1[c]), L.r2[c]);
                    v = make_float4(v.x * p.out_scale, v.y * p.out_scale, v.z * p.out_scale, v.w * p.out_scale);
                }
                if (RB_ACC) v = f4_add(v, L.ro[c]);
                u32x4 bits;
                __builtin_memcpy(&bits, &v, 16);
                if (!(ESTD_W3ABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(bits, qo, eoff[m][c], so, 0);
            }
        };
        auto epi_plane = [&](const f32x4 (&a)[2][2], int dd) {
            const int so = dd * out_plane_bytes;
            EpiLoads L0, L1;
            epi_issue(0, so, rs_res, rs_res2, rs_out, L0);
            epi_issue(1, so, rs_res, rs_res2, rs_out, L1);
            epi_finish(a[0], 0, so, rs_out, L0);
            epi_finish(a[1], 1, so, rs_out, L1);
        };
        // DEFER: the finished plane of the previous tile of this column segment and where it goes
        f32x4 py[2][2];
        int pso = 0;
        bool have_prev = false;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int c = 0; c < 2; ++c) py[m][c] = (f32x4){0.f, 0.f, 0.f, 0.f};

        // raw planes in registers: xa = x[d0-1], xb = x[d0], xc = x[d0+1], xd = x[d0+2]
        float4 xa[SIT], xb[SIT], xc[SIT], xd[SIT];
        {
            const int d0 = 2 * dp;
            load_plane(d0 - 1, xa);
            load_plane(d0, xb);
            load_plane(d0 + 1, xc);
            load_plane(d0 + 2, xd);
        }
        float ea = 0.f, eb = 0.f, ec = 0.f, ed = 0.f;    // EXTRA: the scalar channel's four planes at this thread's voxel
        if (EXTRA) {
            const int d0 = 2 * dp;
            ea = load_x(d0 - 1); eb = load_x(d0); ec = load_x(d0 + 1); ed = load_x(d0 + 2);
        }
        // (threads without a voxel write into the dummy area: no branch in the tap loop)
        char* xw_ptr = tid < SL_VOX ? reinterpret_cast<char*>(lds_x) + tid * 4 : smem + SLICES_BYTES + RED_BYTES + SS_BYTES + VTAB_BYTES + XCH_BYTES + (tid & 63) * 4;
        auto write_x_slice = [&](int sl) {
            if (EXTRA) *reinterpret_cast<float*>(xw_ptr + sl * (SL_VOX * 4)) = sl == 0 ? ea - ec : sl == 1 ? eb + ec : sl == 2 ? ec - eb : eb - ed;
        };
        // depth transform B^T x of the planes in (xa, xb, xc, xd), straight into LDS slice sl
        auto write_slice = [&](int sl) {
#pragma unroll
            for (int it = 0; it < SIT; ++it) {
                if ((it < SIT - 1 || last_ok) && !(ESTD_W3ABL & 2)) {
                    const float4 v = sl == 0 ? f4_sub(xa[it], xc[it]) : sl == 1 ? f4_add(xb[it], xc[it])
                                   : sl == 2 ? f4_sub(xc[it], xb[it]) : f4_sub(xb[it], xd[it]);
                    *reinterpret_cast<float4*>(smem + loffk[it] + sl * SLICE_BYTES) = v;
                }
            }
        };
        auto shift_planes = [&]() {                      // planes d0+1, d0+2 are planes d0'-1, d0' of the next tile
#pragma unroll
            for (int it = 0; it < SIT; ++it) { xa[it] = xc[it]; xb[it] = xd[it]; }
            if (EXTRA) { ea = ec; eb = ed; }
        };
        bool first = true;
        int stats_parity = 0;
        float4 bq[BD][2];                                // weight ring: [buffer][tap pair]: (sw 2p, e 0), (2p, 1), (2p + 1, 0), (2p + 1, 1) of a half-sub-step
        bool w_primed = false;                           // (per column segment: held across its setup code the ring spills)

        for (; u < seg_end; ++u, ++dp) {
            const int d0 = 2 * dp;
            if (first) {
                lds_barrier();                          // every wave is done reading the previous segment's slices
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) write_slice(sl);
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) write_x_slice(sl);
                shift_planes();
                lds_barrier();
                first = false;
            }
            const bool has_next = (u + 1 < seg_end);     // wave-uniform
            const int nd = d0 + 3;                       // new planes of the next tile: nd, nd + 1
            const __amdgpu_buffer_rsrc_t rs_pf0 = (has_next && nd < D) ? rs_in : rs_null, rs_pf1 = (has_next && nd + 1 < D) ? rs_in : rs_null;
            const __amdgpu_buffer_rsrc_t rs_px0 = (EXTRA && has_next && nd < D) ? rs_ex : rs_null, rs_px1 = (EXTRA && has_next && nd + 1 < D) ? rs_ex : rs_null;

            f32x4 P[2][2][2];                            // partial sums [plane][row][column] of this wave's two row-transform indices
            // the previous tile's plane leaves inside this tile's first half-sub-steps (no previous tile: null descriptors, the stores are dropped)
            const __amdgpu_buffer_rsrc_t rp_out = have_prev ? rs_out : rs_null, rp_res = have_prev ? rs_res : rs_null, rp_res2 = have_prev ? rs_res2 : rs_null;

            // one tile's tap loop for the waves of row-transform half SHH (compile-time: rows, signs and the output-transform rows differ)
            auto tap_loop = [&](auto shh_c) {
                constexpr int SHH = decltype(shh_c)::value;
                // half-sub-step q = (sd = q >> 3, channel chunk cc = (q >> 2) & 1, channel-pair half hh = (q >> 1) & 1, sl = q & 1): 8 MFMAs =
                // 4 taps sw x 2 k-steps e.  (Registers: a 16-MFMA unit on 16-byte fragment reads needs 56 more -- it spilled 128.)
                constexpr int NQ = 32;
                // halo rows (relative to the block's first) of the two raw rows of a half-sub-step: even q: A = RA0, B = RSH; odd q: A = RA1, B stays
                //   SHH 0: sh 0 = r0 - r2, sh 1 = r1 + r2       SHH 1: sh 2 = r2 - r1, sh 3 = r1 - r3
                constexpr int RA0 = SHH == 0 ? 0 : 2, RSH = SHH == 0 ? 2 : 1, RA1 = SHH == 0 ? 1 : 3;
                f32x4 m[2][4];                           // products [sl][sw] of the current depth transform
                f32x2 RA[4], RBs[4];                     // raw rows (four block columns, one channel pair) of the next half-sub-step
                f32x2 T[4];                              // [sw], components = k-steps e

                auto load_w = [&](int q, float4 (&b)[2]) {
                    const int sd = q >> 3, cc = (q >> 2) & 1, hh = (q >> 1) & 1, sh = 2 * SHH + (q & 1);
#pragma unroll
                    for (int sp = 0; sp < 2; ++sp) {
                        if (WL && q >= WL_Q0 && q < WL_Q0 + WL_N) b[sp] = *reinterpret_cast<const float4*>(lds_wl + (SHH * WL_N + q - WL_Q0) * 4096 + sp * 1024 + wlane);
                        else b[sp] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, ((((sd * 4 + sh) * 2 + cc) * 2 + hh) * TAP_BYTES) + sp * 1024, 0));
                    }
                };
                auto load_rowA = [&](int q) {
                    const int sd = q >> 3, cc = (q >> 2) & 1, hh = (q >> 1) & 1, r = (q & 1) ? RA1 : RA0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) RA[j] = *reinterpret_cast<const f32x2*>(smem + cbase[j][cc][hh] + sd * SLICE_BYTES + r * ROW_BYTES);
                };
                auto load_rowB = [&](int q) {
                    const int sd = q >> 3, cc = (q >> 2) & 1, hh = (q >> 1) & 1;
#pragma unroll
                    for (int j = 0; j < 4; ++j) RBs[j] = *reinterpret_cast<const f32x2*>(smem + cbase[j][cc][hh] + sd * SLICE_BYTES + RSH * ROW_BYTES);
                };
                // the four transformed operands (two k-steps each) of half-sub-step q from its raw rows: row combination, then the column transform
                auto xform = [&](int q, f32x2 (&o)[4]) {
                    f32x2 X[4];
                    if (ESTD_W3PK && !(ESTD_W3ABL & 128)) {
                        // inline assembly: left to itself the compiler emits ~60 % of these as two plain adds each (a heuristic for the bf16 matrix pipe)
                        auto pk_sub = [](f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; };
                        auto pk_add = [](f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
#pragma unroll
                        for (int j = 0; j < 4; ++j) X[j] = (q & 1) == 0 ? pk_sub(RA[j], RBs[j]) : SHH == 0 ? pk_add(RA[j], RBs[j]) : pk_sub(RBs[j], RA[j]);
                        o[0] = pk_sub(X[0], X[2]); o[1] = pk_add(X[1], X[2]); o[2] = pk_sub(X[2], X[1]); o[3] = pk_sub(X[1], X[3]);
                        return;
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (ESTD_W3ABL & 128) X[j] = RA[j];
                        else if ((q & 1) == 0) X[j] = RA[j] - RBs[j];
                        else X[j] = SHH == 0 ? RA[j] + RBs[j] : RBs[j] - RA[j];
                    }
                    if (ESTD_W3ABL & 128) { o[0] = X[0]; o[1] = X[1]; o[2] = X[2]; o[3] = X[3]; }
                    else { o[0] = X[0] - X[2]; o[1] = X[1] + X[2]; o[2] = X[2] - X[1]; o[3] = X[1] - X[3]; }
                };

                if (!ESTD_W3_WRING || !w_primed) {                         // the weight stream runs on across tiles (every tile reads the same 64 blocks): filled once per column segment
#pragma unroll
                    for (int b = 0; b < BD - 1; ++b) load_w(b, bq[b]);
                    w_primed = true;
                }
                load_rowA(0);
                load_rowB(0);
                xform(0, T);
                load_rowA(1);
                if (EXTRA && !(ESTD_W3ABL & 256)) {
                    f32x2 xpt[3][2];                     // the block's patch of scalar slice sd = g: rows RA0, shared, RA1 x column pairs
#pragma unroll
                    for (int r3 = 0; r3 < 3; ++r3)
#pragma unroll
                        for (int jp = 0; jp < 2; ++jp) {
                            const int row = r3 == 0 ? RA0 : r3 == 1 ? RSH : RA1;
                            xpt[r3][jp] = *reinterpret_cast<const f32x2*>(reinterpret_cast<const char*>(lds_x) + xbase + row * (IN_W * 4) + jp * 8);
                        }
                    float tx[2][4];
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const f32x2 a0 = s2 == 0 ? xpt[0][0] : xpt[2][0], a1 = s2 == 0 ? xpt[0][1] : xpt[2][1];
                        const f32x2 X01 = s2 == 0 ? a0 - xpt[1][0] : (SHH == 0 ? a0 + xpt[1][0] : xpt[1][0] - a0);
                        const f32x2 X23 = s2 == 0 ? a1 - xpt[1][1] : (SHH == 0 ? a1 + xpt[1][1] : xpt[1][1] - a1);
                        tx[s2][0] = X01.x - X23.x; tx[s2][1] = X01.y + X23.x; tx[s2][2] = X23.x - X01.y; tx[s2][3] = X01.y - X23.y;
                    }
#pragma unroll
                    for (int pl_ = 0; pl_ < 2; ++pl_) {
                        f32x4 mx[2][4];
#pragma unroll
                        for (int s2 = 0; s2 < 2; ++s2) {
                            const float4 w4 = *reinterpret_cast<const float4*>(lds_wx + ((pl_ * 4 + 2 * SHH + s2) * 2 + nh) * 1024 + lane * 16);
                            mx[s2][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.x, tx[s2][0], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                            mx[s2][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.y, tx[s2][1], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                            mx[s2][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.z, tx[s2][2], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                            mx[s2][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(w4.w, tx[s2][3], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                        }
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const f32x4 v0_ = c == 0 ? mx[0][0] + mx[0][1] + mx[0][2] : mx[0][1] - mx[0][2] - mx[0][3];
                            const f32x4 v1_ = c == 0 ? mx[1][0] + mx[1][1] + mx[1][2] : mx[1][1] - mx[1][2] - mx[1][3];
                            if (SHH == 0) { P[pl_][0][c] = v0_ + v1_; P[pl_][1][c] = v1_; }
                            else          { P[pl_][0][c] = v0_;       P[pl_][1][c] = -(v0_ + v1_); }
                        }
                    }
                } else if (EXTRA) {
#pragma unroll
                    for (int pl_ = 0; pl_ < 2; ++pl_)
#pragma unroll
                        for (int r = 0; r < 2; ++r)
#pragma unroll
                            for (int c = 0; c < 2; ++c) P[pl_][r][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
                unsigned vo_next = 0;
                EpiLoads pl, pl1;
                constexpr int PF_Q = RBK != 0 ? ESTD_W3_PFQ_RB : ESTD_W3_PFQ;    // (read-back launches: later, clear of the read-back loads of half-sub-steps 0, 1: running sum 0.735 -> 0.710 ms)
                // next-plane prefetch: one chunk per two half-sub-steps, q = PF_Q, PF_Q + 2, .. PF_Q + 10
                constexpr int RW_Q = 22;                 // slices 0..2 of the next tile: every read of them has been issued (rows are fetched two half-sub-steps ahead)
                __builtin_amdgcn_sched_barrier(0);

#pragma clang loop unroll(full)
                for (int q = 0; q < NQ; ++q) {
                    const int sd = q >> 3, sl = q & 1;
                    if (q == RW_Q) lds_barrier();        // (also on a segment's last tile: it separates the partner's read of the exchange buffer from this tile's write)
                    if (q >= RW_Q && q < RW_Q + 9 && !(ESTD_W3ABL & 2)) {      // one 16-byte chunk of the next tile's slices 0..2 per half-sub-step (no next tile: dead slices, harmless)
                        const int sl_w = (q - RW_Q) / SIT, it = (q - RW_Q) % SIT;
                        const float4 v = sl_w == 0 ? f4_sub(xa[it], xc[it]) : sl_w == 1 ? f4_add(xb[it], xc[it]) : f4_sub(xc[it], xb[it]);
                        *reinterpret_cast<float4*>(smem + loffw[it] + sl_w * SLICE_BYTES) = v;
                    }
                    if ((ESTD_W3_WRING || q + BD - 1 < NQ) && !(ESTD_W3ABL & 8)) load_w((q + BD - 1) % NQ, bq[(q + BD - 1) % BD]);      // (the last ones: the next tile's first blocks)
                    if (q >= PF_Q && q < PF_Q + 12 && ((q - PF_Q) & 1) == 0 && !(ESTD_W3ABL & 16)) {      // (no next tile / plane outside the volume: null descriptor, zeros)
                        const int idx = (q - PF_Q) >> 1, it = idx % SIT;
                        if (idx < SIT) xc[it] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_pf0, vo_next, nd * in_slice_bytes, 0));
                        else           xd[it] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_pf1, vo_next, (nd + 1) * in_slice_bytes, 0));
                    }
                    if (EXTRA) {
                        // the scalar channel: its next planes requested at q = 16, 17, ALL FOUR of its slices rewritten behind the barrier (they are read at the top of a tile only)
                        if (q == 16) ec = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_px0, xoff, nd * HW * 4, 0));
                        if (q == 17) ed = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_px1, xoff, (nd + 1) * HW * 4, 0));
                        if (q >= RW_Q && q < RW_Q + 4) write_x_slice(q - RW_Q);
                    }
                    if (DEFER) {
                        // read-back instances: the two halves requested at q = 0, 1 and finished at q = RBQ, RBQ + 1 -- vector-memory loads return in
                        // order: a weight request behind a read-back load waits for HBM with it, and the loads' first use waits for them; else finished at q = 1, 2
                        constexpr bool RB = RBK != 0;
                        constexpr int RBQ = ESTD_W3_RBQ;
                        if (RB && q == 0) epi_issue(0, pso, rp_res, rp_res2, rp_out, pl);
                        if (RB && q == 1) epi_issue(1, pso, rp_res, rp_res2, rp_out, pl1);
                        if (q == (RB ? RBQ : 1)) epi_finish(py[0], 0, pso, rp_out, pl);
                        if (q == (RB ? RBQ + 1 : 2)) epi_finish(py[1], 1, pso, rp_out, pl1);
                    }
                    if (ESTD_W3_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
                    const int cur = (ESTD_W3ABL & 8) ? 0 : q % BD;
                    f32x2 Tn[4];
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int sw = 0; sw < 4; ++sw) {
                            const float4 b4 = bq[cur][sw >> 1];
                            const float b = (sw & 1) == 0 ? (e == 0 ? b4.x : b4.y) : (e == 0 ? b4.z : b4.w);
                            const bool first_product = (q & 7) < 2 && e == 0;       // channel chunk 0, pair half 0, k-step 0 of the depth transform
                            const f32x4 c_in = first_product ? (f32x4){0.f, 0.f, 0.f, 0.f} : m[sl][sw];
                            m[sl][sw] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, T[sw][e], c_in, 0, 0, 0);
                        }
                    if (q + 1 < NQ) xform(q + 1, Tn);
                    if (q + 2 < NQ) {
                        load_rowA(q + 2);
                        if (((q + 2) & 1) == 0) load_rowB(q + 2);
                    }
                    if (q + 1 >= PF_Q && q + 1 < PF_Q + 12 && ((q + 1 - PF_Q) & 1) == 0 && !(ESTD_W3ABL & 16)) vo_next = chunk_voff(((q + 1 - PF_Q) >> 1) % SIT);
                    if (ESTD_W3_SCHED == 0) {
                        // order of the region: four MFMAs, half the next half-sub-step's transform, four MFMAs, the other half, the fragment reads
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
                    } else {
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            if (EXTRA && (q & 7) == 4) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);      // (+ the scalar channel's 8 MFMAs)
                            else __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);       // one vector-memory read (weights, next plane, read-back)
                            __builtin_amdgcn_sched_group_barrier(0x002, ESTD_W3_SCHED_VALU, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);       // (q >= 22: the slice chunk)
                        }
                    }
                    if (q + 1 < NQ) {
#pragma unroll
                        for (int sw = 0; sw < 4; ++sw) T[sw] = Tn[sw];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if ((q & 7) == 7 && !(ESTD_W3ABL & 32)) {
                        // depth transform sd complete: column half, then this wave's part of the row half of the output transform, then the planes
                        f32x4 z[2][2];                   // [row][column]
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const f32x4 v0_ = c == 0 ? m[0][0] + m[0][1] + m[0][2] : m[0][1] - m[0][2] - m[0][3];
                            const f32x4 v1_ = c == 0 ? m[1][0] + m[1][1] + m[1][2] : m[1][1] - m[1][2] - m[1][3];
                            if (SHH == 0) { z[0][c] = v0_ + v1_; z[1][c] = v1_; }                  // rows: y0 = t0 + t1 (+ t2), y1 = t1 (- t2 - t3)
                            else          { z[0][c] = v0_;       z[1][c] = -(v0_ + v1_); }         //       y0 = (..) + t2,      y1 = (..) - t2 - t3
                        }
#pragma unroll
                        for (int r = 0; r < 2; ++r)
#pragma unroll
                            for (int c = 0; c < 2; ++c) {
                                if (sd == 0) { if (EXTRA) P[0][r][c] += z[r][c]; else P[0][r][c] = z[r][c]; }
                                else if (sd == 1) { P[0][r][c] += z[r][c]; if (EXTRA) P[1][r][c] += z[r][c]; else P[1][r][c] = z[r][c]; }
                                else if (sd == 2) { P[0][r][c] += z[r][c]; P[1][r][c] -= z[r][c]; }
                                else P[1][r][c] -= z[r][c];
                            }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (ESTD_W3ABL & 32) {
#pragma unroll
                    for (int r = 0; r < 2; ++r)
#pragma unroll
                        for (int c = 0; c < 2; ++c) { P[0][r][c] = m[0][2 * r + c]; P[1][r][c] = m[1][2 * r + c]; }
                }
                // the other plane's partial sums go to the partner wave (wave ^ 4: same rows and channels, the other half of the row transform)
                float4* xw = lds_xch + (wave * 4) * 64 + lane;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const f32x4 snd = P[1 - SHH][r][c];
                        xw[(2 * r + c) * 64] = make_float4(snd[0], snd[1], snd[2], snd[3]);
                    }
                lds_barrier();                            // every wave has read slice 3 for the last time; the exchange is visible
                const float4* xr = lds_xch + ((wave ^ 4) * 4) * 64 + lane;
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float4 o = xr[(2 * r + c) * 64];
                        P[0][r][c] = P[SHH][r][c] + (f32x4){o.x, o.y, o.z, o.w};           // P[0] now = the finished outputs of plane d0 + SHH
                    }
            };
            if (shh == 0) tap_loop(std::integral_constant<int, 0>{});
            else tap_loop(std::integral_constant<int, 1>{});

            if (STATS) {
                // fixed-order reduction -> deterministic: the lanes of a wave by DPP moves + two shuffles, the two row quads of a (plane, channel half)
                // through LDS.  Partial index = canonical tile id (n, d, thi, twi) of each plane, as the direct kernel writes it.
                double* red = reinterpret_cast<double*>(smem + SLICES_BYTES) + (stats_parity ? 32 : 0);      // [plane][channel half][row quad][sum, sumsq]
                stats_parity ^= 1;
                double v0_ = 0.0, v1_ = 0.0;
                const int chb = 16 * nh + 4 * g;
                const float4 sc4 = *reinterpret_cast<const float4*>(lds_ss + chb), sh4 = *reinterpret_cast<const float4*>(lds_ss + 32 + chb);
                const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        if (eoff[m][c] != OOB_OFFSET) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) { const double uu = (double)(P[0][m][c][r] * scv[r] + shv[r]); v0_ += uu; v1_ += uu * uu; }
                        }
                v0_ = row16_sum_f64(v0_); v1_ = row16_sum_f64(v1_);
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) { v0_ += __shfl_xor(v0_, o); v1_ += __shfl_xor(v1_, o); }
                if (lane == 0) { red[((shh * 2 + nh) * 2 + rq) * 2] = v0_; red[((shh * 2 + nh) * 2 + rq) * 2 + 1] = v1_; }
                __syncthreads();
                if (tid < 8) {                                       // (plane, channel half, {sum, sumsq})
                    const int pl_ = tid >> 2, grp = (tid >> 1) & 1, qq = tid & 1;
                    if (d0 + pl_ < D) {                              // (odd D: the last pair has one plane)
                        const double tot = red[((pl_ * 2 + grp) * 2 + 0) * 2 + qq] + red[((pl_ * 2 + grp) * 2 + 1) * 2 + qq];
                        const size_t tile_id = (((size_t)n * D + d0 + pl_) * tiles_h + thi) * tiles_w + twi;
                        p.stats_partials[tile_id * 4 + grp * 2 + qq] = tot;
                    }
                }
            }
            if (has_next) {                               // slice 3 of the next tile (published by the next tile's in-loop barrier: first read at the end of its sub-step 10)
                write_slice(3);
                shift_planes();
            }
            if (DEFER && has_next) {                      // (a tile with a successor is never the odd last plane pair: d0 + shh < D)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int c = 0; c < 2; ++c) py[m][c] = P[0][m][c];
                pso = (d0 + shh) * out_plane_bytes;
                have_prev = true;
            } else {
                if (d0 + shh < D) epi_plane(P[0], d0 + shh);
                have_prev = false;
            }
        }
    }
}

constexpr int PERSISTENT_WGS = 256;     // one 512-thread workgroup per CU (LDS-limited)

}  // namespace

extern "C" int estd_conv3d_k3_wino3(const estd_conv3d_desc* dp, estd_stream_t s)
{
    if (!dp) return ESTD_ERR_ARG;
    const estd_conv3d_desc& d = *dp;
    if (d.N <= 0 || d.D <= 0 || d.H <= 0 || d.W <= 0) return ESTD_ERR_ARG;
    if (!d.in_main || !d.w_wino2 || !d.scale || !d.shift || !d.out_main) return ESTD_ERR_ARG;
    // 32 output channels: 32 -> 32 with every read-back epilogue, 32 -> 32 + GroupNorm partial sums and 33 -> 32 (scalar input channel) without
    // read-back streams; no 33rd output channel, no fused head, no gate (include/estd_hip.h)
    if (d.cin_main != 32 || d.n_tiles != 2 || d.out_head || d.out_extra || d.gate_r) return ESTD_ERR_UNSUPPORTED;
    const bool extra = d.in_extra != nullptr;
    if (extra != (d.w_extra != nullptr)) return ESTD_ERR_ARG;
    if (d.in_stride < 32 || (d.in_stride & 3) || d.out_stride < 32 || (d.out_stride & 3) || (d.act_split & 1)) return ESTD_ERR_ARG;
    const int tiles_w = (d.W + TW - 1) / TW, tiles_h = (d.H + TH - 1) / TH, dpairs = (d.D + 1) / 2;
    const long long total = (long long)d.N * dpairs * tiles_h * tiles_w;
    if (total > 0x7fffffffLL) return ESTD_ERR_ARG;
    {   // buffer descriptors address one volume of the batch with 32-bit byte offsets
        const long long vox = (long long)d.D * d.H * d.W;
        const int widest = d.in_stride > d.out_stride ? d.in_stride : d.out_stride;
        if (vox * widest * 4 >= 0x7fffff00LL) return ESTD_ERR_UNSUPPORTED;
    }
    const int slots = estd_persistent_wgs(PERSISTENT_WGS / 256);
    int grid = total < slots ? (int)total : slots;
    if (grid >= 8) grid &= ~7;
    // read-back kind of the launch (the scale multiply lives in the residual instances)
    const bool res_any = d.residual || d.residual2 || d.out_scale != 1.0f;
    const int rbk = (!res_any && !d.accumulate) ? 0 : (!res_any ? 1 : (!d.accumulate ? 2 : 3));
    if (d.stats_partials && rbk != 0) return ESTD_ERR_UNSUPPORTED;       // GroupNorm partial sums: without read-back streams (the gate convolution has none)
#define ESTD_W3_LAUNCH(RBV, STV, EXV)                                                                                                           \
    do {                                                                                                                             \
        estd_allow_dynamic_lds<conv3d_wino3_kernel<RBV, STV, EXV>>(EXV ? LDS_BYTES_EXTRA : LDS_BYTES_PLAIN);                                                                 \
        hipLaunchKernelGGL((conv3d_wino3_kernel<RBV, STV, EXV>), dim3(grid), dim3(NTHREADS), EXV ? LDS_BYTES_EXTRA : LDS_BYTES_PLAIN, estd_stream(s), d, tiles_w, tiles_h,   \
                           dpairs, (int)total);                                                                                      \
    } while (0)
    if (extra && (rbk != 0 || d.stats_partials)) return ESTD_ERR_UNSUPPORTED;       // the key || value convolution has neither
    switch (rbk) {
    case 0: if (extra) ESTD_W3_LAUNCH(0, false, true); else if (d.stats_partials) ESTD_W3_LAUNCH(0, true, false); else ESTD_W3_LAUNCH(0, false, false); break;
    case 1: ESTD_W3_LAUNCH(1, false, false); break;
    case 2: ESTD_W3_LAUNCH(2, false, false); break;
    default: ESTD_W3_LAUNCH(3, false, false); break;
    }
#undef ESTD_W3_LAUNCH
    return ESTD_LAUNCH_CHECK();
}
