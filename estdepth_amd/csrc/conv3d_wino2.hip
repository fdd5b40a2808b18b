// conv3d_wino2.hip -- the plain 32 -> 32 3x3x3 convolution with TWO axes (depth and image rows) in Winograd F(2,3) form, on
// gfx950 fp32 MFMA: F(2x2, 3x3) over (d, h), a 3-tap direct convolution along w.
//
// Same operator and descriptor as estd_conv3d_k3 / estd_conv3d_k3_wino (networks/layers_op.py:16-39 as used at
// hybrid_models/model_hybrid.py:59-60,:95, hybrid_models/hybrid_depth_decoder.py:84-95,:190-191 and the gate convolution of
// transformer/epipolar_transformer.py:21).  The depth-only Winograd kernel (csrc/conv3d_wino.hip) keeps the matrix pipe 80 % busy
// with 2/3 of the direct kernel's products; the lever left in fp32 is fewer products again:
//
//   2 x 2 outputs (planes d, d+1; rows y, y+1) of one w position from the 4 x 4 input patch (planes d-1..d+2, rows y-1..y+2):
//       T = B^T x B   (B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1] on the depth axis, then on the row axis)
//       U = G g G^T   (G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1] on kd, then on kh; float64 on the host, rounded once)
//       m[sd][sh] = conv1d_w(T[sd][sh], U[sd][sh][kw])        16 x 3 tap products
//       y = A^T m A   (A^T = [1 1 1 0; 0 1 -1 -1])
//   48 tap products per 4 outputs instead of 4 x 27: 0.444 of the direct MFMA work (depth-only form: 0.667); every product is
//   still a v_mfma_f32_16x16x4_f32 with fp32 accumulation.
//
// Work decomposition: the depth-only kernel's (512 threads = 8 waves, ONE workgroup per CU, output tile 2 x 8 x 16 voxels,
// persistent XCD-contiguous ranges of the column-major tile list, the four DEPTH-transformed slices of 10 x 18 voxels x 32
// channels in LDS with the same swizzle and MFMA row <-> voxel permutation, raw planes carried in registers, the next tile's
// slices written inside the tap loop, the same epilogue).  What differs:
//   * the ROW transform is applied in registers between LDS and the MFMA: wave (rp, nh) owns the row pair 2rp, 2rp+1 of the tile,
//     i.e. halo rows 2rp .. 2rp+3; per group (sd, kw) it reads those four rows ONCE (8 ds_read_b128, as many as the depth-only
//     kernel reads per four taps) and forms the four transformed fragments with one VALU add/sub per operand register
//     (1 VALU per MFMA, issued in the MFMA's shadow): t0 = r0 - r2, t1 = r1 + r2, t2 = r2 - r1, t3 = r1 - r3;
//   * 16 accumulators m[sd][sh] (64 registers) for one 16-voxel M tile x one 16-channel N tile; two of them alternate inside a
//     step (the 16x16x4 MFMA has a 40-cycle dependent latency at a 32-cycle issue interval);
//   * 48 weight taps of 4 KB (192 KB): taps 0..15 from an LDS copy made once per workgroup, the rest streamed from L2 one step
//     (two taps) ahead; tap = (3 sd + kw) * 4 + sh.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "estd_hip.h"
#include "estd_common.h"

#ifndef ESTD_W2PRIO
#define ESTD_W2PRIO 2   // 0: none; 1: per-step alternating s_setprio between the two waves of a SIMD (slower); 2: static priority 1 for waves 4..7 (default: -1.2 % at sustained clocks)
#endif
#ifndef ESTD_W2PK
#define ESTD_W2PK 0     // row transforms: 0 vector arithmetic (the compiler packs some, unpacks others next to MFMAs), 1 inline-assembly
#endif                  // v_pk_add_f32 fenced between the MFMA halves, 2 inline assembly without inner fences
#ifndef ESTD_W2_LIBM_TANH
#define ESTD_W2_LIBM_TANH 0     // A/B: tanhf of the device library in the epilogue instead of tanh_fast
#endif
#ifndef ESTD_W2_RB_BATCH
#define ESTD_W2_RB_BATCH 0      // A/B: read-back loads of both planes of a tile in one batch (82 spilled registers: 0.93 -> 1.38 ms)
#endif
#ifndef ESTD_W2_STATS_DEFER
#define ESTD_W2_STATS_DEFER 0   // A/B: deferred epilogue also for the launches that write GroupNorm partial sums
#endif
#ifndef ESTD_W2DEFER
#define ESTD_W2DEFER 1  // 1: epilogue of tile k inside the first steps of tile k+1, two barriers per tile; 0: epilogue between the tiles
#endif
#ifndef ESTD_W2_AUX_IN
#define ESTD_W2_AUX_IN 0     // cache policy of the plane loads (A/B: 2 = non-temporal)
#endif
#ifndef ESTD_W2_AUX_OUT
#define ESTD_W2_AUX_OUT 0    // cache policy of the output stores (A/B: 2 = non-temporal)
#endif
#ifndef ESTD_W2_QSCHED
#define ESTD_W2_QSCHED 0   // A/B (round 5): the four weight requests of a step one per quarter of its 16 MFMAs (the schedule that pays in csrc/conv2d_wino2.hip): +-0.1 % here, 33 -> 33 +0.7 % (profiles/r5_conv3d_wino2_qsched.txt)
#endif
#ifndef ESTD_W2FOLD
#define ESTD_W2FOLD 0   // 1: the products of a depth transform are folded into the output planes as soon as it is complete (32 accumulator registers less)
#endif
#ifndef ESTD_W2ABL
#define ESTD_W2ABL 0    // timing ablations only (results are wrong): 1 no output stores, 2 no slice writes, 8 no weight stream,
#endif                  // 16 no next-plane prefetch, 128 no row transform (raw rows as operands)

#ifdef ESTD_W2TIME
// debug build only: s_memtime stamps of waves 0 and 4 of workgroup 0 at fixed points of its first tiles, written over the
// GroupNorm partial-sum buffer (tools/wino2_timeline.py); perturbs the timing a little (every stamp drains lgkmcnt)
#define W2STAMP(pt)                                                                                                         \
    do {                                                                                                                    \
        if (blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && p.stats_partials && tl_tile < 8)                             \
            p.stats_partials[((wave >> 2) * 8 + tl_tile) * 16 + (pt)] = (double)__builtin_amdgcn_s_memtime();               \
    } while (0)
#else
#define W2STAMP(pt) do { } while (0)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));

constexpr int TH = 8, TW = 16;
constexpr int IN_H = TH + 2, IN_W = TW + 2;
constexpr int SL_VOX = IN_H * IN_W;                 // 180 voxels per input slice (with halo)
constexpr int SLICE_BYTES = SL_VOX * 128;           // 32 channels
constexpr int SL_CHUNKS = SL_VOX * 8;               // 16-byte chunks per slice: 1440
constexpr int RED_BYTES = 2 * 8 * 4 * 8;            // GroupNorm scratch: 2 copies x 8 (row pair, channel half) x {sum, sumsq} x 2 planes, doubles
constexpr int NTAPS = 48;
#ifndef ESTD_W2LDS_TAPS
#define ESTD_W2LDS_TAPS 14
#endif
constexpr int WLDS_TAPS = ESTD_W2LDS_TAPS;           // weights of WLDS_TAPS consecutive taps of every tile come from LDS ...
#ifndef ESTD_W2LDS_T0
#define ESTD_W2LDS_T0 0     // with the weights two steps ahead (ESTD_W2BD = 3) the window's place no longer matters for the plain instance and tap 0 is best for the read-back instances (0: 0.806-0.808 / 0.821 / 0.826 / 0.882, 4: 0.808 / 0.832 / 0.840 / 0.890, 2: 0.805-0.808 / 0.841-0.845 / 0.840-0.843 / 0.891 ms plain / + sum / + residual / + 2 residuals).  WITH ONE STEP OF COVER (BD = 2) it was 4: taps 4..17 = steps 2..8: the plane prefetch (steps 4..9) then sits next to LDS-fed steps -- a weight request behind a plane request waits for HBM (vector-memory loads return in order); T0 = 0: 0.839, 4: 0.823, 6: 0.827, 8: 0.826-0.833, 10: 0.850, 12: 0.845, 16: 0.871 ms (N = 3)
#endif
constexpr int WLDS_T0 = ESTD_W2LDS_T0;               // ... starting with this tap (32-channel instances; the 16-output-channel instance: tap 0)
constexpr int WLDS_BYTES = WLDS_TAPS * 4096;         // [tap][2 halves][2 quads][64 lanes][4]
constexpr int SS_BYTES = 3 * 32 * 4;                 // folded BN scale | shift | activation floor of the 32 output channels
constexpr int VTAB_BYTES = 6 * 256 * 4 + 512 * 4;     // per-thread global offsets of the slice chunks (3 x 512 or 6 x 256 threads) + of the scalar channel's voxel
constexpr int XSL_BYTES = 4 * SL_VOX * 4;             // EXTRA: the four depth-transformed slices of the scalar 33rd input channel
constexpr int LDS_BYTES = 4 * SLICE_BYTES + RED_BYTES + SS_BYTES + VTAB_BYTES + XSL_BYTES + WLDS_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
// O16 (16 output channels): a tap's weights are 2 KB; 20 taps in LDS + the 16 KB exchange buffer of the cross-wave reduction
constexpr int O16_WLDS_TAPS = 20;
constexpr int O16_XCH_BYTES = 8 * 2 * 64 * 16;
constexpr int O16_LDS_BYTES = 4 * SLICE_BYTES + RED_BYTES + SS_BYTES + VTAB_BYTES + O16_XCH_BYTES + O16_WLDS_TAPS * 2048;
static_assert(O16_LDS_BYTES <= 160 * 1024, "LDS budget (O16)");
// XOUT (a 33rd OUTPUT channel, dres2): two LDS-resident taps less; their place takes the 33rd channel's own weights -- [24 steps][4 sh][4 lane
// groups][4] main input channels + [4 sd][4 lane groups][4 sh] scalar input channel (packing.pack_conv3d_wino2_xout) -- and the exchange
// buffer of its cross-wave sum, [2 tile parities][4 row pairs][64 lanes]
constexpr int XOUT_WLDS_TAPS = WLDS_TAPS - 2;
constexpr int XOUT_W_BYTES = (24 * 4 * 4 + 4 * 4) * 16;
constexpr int XOUT_XCH_BYTES = 2 * 4 * 64 * 4;
constexpr int XOUT_LDS_BYTES = 4 * SLICE_BYTES + RED_BYTES + SS_BYTES + VTAB_BYTES + XSL_BYTES + XOUT_WLDS_TAPS * 4096 + XOUT_W_BYTES + XOUT_XCH_BYTES;
static_assert(XOUT_LDS_BYTES <= 160 * 1024, "LDS budget (XOUT)");
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

// workgroup barrier that only orders LDS traffic (no vmcnt drain: prefetches and output stores stay in flight)
// sum of a double over the 16 lanes of a DPP row, every lane gets the total: two quad permutations + two row rotations of the two
// dwords (v_mov_b32 with a DPP modifier: no LDS round trip) and one v_add_f64 per level
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

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int lds_chunk_off(int v, int c) { return v * 128 + ((c ^ ((v >> 1) & 7)) << 4); }
// ESTD_W2COLKEY (round 4): the swizzle key is taken from the COLUMN of the haloed slice only -- (col >> 1) & 7; IN_W is even, so the
// bank half (voxel & 1) is the column's parity and the 8 even / 8 odd columns a fragment read touches still have 8 distinct keys --
// which makes the byte offset of (row, col, chunk) = row * IN_W * 128 + f(col, chunk): the fragment reads of a tap loop then need one
// address register per (column tap, channel chunk) and immediates for the halo row and the depth slice, instead of one register per
// (row, column tap) plus an XOR per second-chunk read.
#ifndef ESTD_W2COLKEY
#define ESTD_W2COLKEY 1
#endif
__device__ __forceinline__ int lds_colkey_off(int col, int c) { return col * 128 + ((c ^ ((col >> 1) & 7)) << 4); }

// tanh x = 1 - 2 / (exp(2x) + 1) on the transcendental units (v_exp_f32, v_rcp_f32: 1 ulp each; five instructions instead of the device
// library's ~30).  Absolute error <= 2e-7 over the whole range (cancellation near 0 costs relative, not absolute accuracy); +-inf -> +-1.
__device__ __forceinline__ float tanh_fast(float x)
{
    const float t = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

__device__ __forceinline__ float act_apply(float v, int act)
{
    if (act == ESTD_ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == ESTD_ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// NW = 8: 512 threads, two waves per SIMD (256 registers each), wave = (row pair rp, channel half nh).
// NW = 4: 256 threads, ONE wave per SIMD (512 registers: the accumulators of both channel halves), wave = row pair rp: every
//         fragment read and every row transform serves 2 x 16 output channels -- half the VALU instructions per MFMA -- and
//         no second wave competes for the SIMD's VALU issue.  Measured cost of one VALU instruction in the 8-wave form:
//         ~6.5 SIMD cycles, NOT hidden behind the MFMAs (time is linear in the VALU count, profiles/r3_wino2_*).
// RB: the launch has read-back streams in its epilogue (residual, residual2 or a running sum); the instance without them
// (conv + BN + activation only) carries no registers for them.
// EXTRA: a scalar 33rd INPUT channel (the key || value convolution, hybrid_depth_decoder.py:190-191 on cat[dres2 output]): its own
// four depth-transformed slices in LDS (2.9 KB); after the tap loop ONE more k-step per product m[sd][sh] -- lane group g multiplies
// column tap kw = g of the row-transformed scalar rows (g = 3: zero weight) -- i.e. 16 more MFMAs per tile and wave.
// O16: 32 -> 16 output channels (the GRU output convolution, transformer/epipolar_transformer.py:26): one 16-channel tile per product, so
// the two waves of a SIMD split the INPUT channels instead -- wave (rp, cw) multiplies chunk cw (16 of the 32 channels) in 12 steps of
// 16 MFMAs -- and sum their outputs through a 16 KB LDS exchange after the output transform: wave cw keeps plane d0 + cw, sends the
// other one to its partner, and runs the epilogue of its plane only.
// XOUT: a 33rd OUTPUT channel on top of EXTRA (dres2, hybrid_depth_decoder.py:106 / :195): a GEMV -- 1/16 efficient as an MFMA tile -- so it
// runs on the VALU from the row-transformed fragments T the MFMAs of a step consume anyway.  The two waves of a SIMD hold the SAME
// fragments (they differ in the output channel half), so they split the k range: wave (rp, nh) takes the steps of channel chunk c = nh --
// 4 weight reads and 8 packed FMAs in every other step -- folds its partial products m33[sd][sh] through both output transforms as the
// depth transforms complete, and after the loop the lane groups (shuffles) and the two waves (1 KB through LDS, published by the tile's
// post-loop barrier) add up; wave nh = 0 applies BN + activation and stores the 2 planes x 2 rows x 16 voxels to out_extra.
// GATE (O16 only): the ConvGRU's reset gate folded into the plane loads of the output convolution (transformer/epipolar_transformer.py:46,:51:
// the convolution's input is cat[x, sigmoid(GN(r)) * h]): threads that own a 16-byte chunk of the h half (chunk index >= 4) also load the same
// chunk of r (the gate convolution's raw output, channels 0..15 of gate_r) and scale their values when a plane arrives -- sigmoid on the
// exp / rcp units, five VALU operations per value, ~170 per wave and tile against a 393 MB pass of its own (estd_gru_reset_apply: 70 us).
template <int NW, int RBK, bool EXTRA, bool O16, bool XOUT = false, bool GATE = false>
__global__ __launch_bounds__(64 * NW, 1) void conv3d_wino2_kernel(const estd_conv3d_desc p, int tiles_w, int tiles_h, int dpairs, int total_tiles)
{
    static_assert(!GATE || (O16 && NW == 8), "GATE: the 32 -> 16 instance");
    // RBK: read-back streams of the epilogue.  0 none; 1 = running sum only (out += result: the second source view of pre1), deferred
    // epilogue like the plain instance; 2 = two residuals + scale (pre2 over both source views), deferred; 3 = any combination, epilogue
    // between the tiles.  (profiles/r4_wino2_rb.txt)
    constexpr bool RB = RBK != 0;
    constexpr bool RB_ACC = RBK == 1 || RBK == 3, RB_RES = RBK == 2 || RBK == 3;
    constexpr int NTHREADS = 64 * NW;
    constexpr int NHW = 8 / NW;                          // channel halves per wave
    constexpr int SIT = (SL_CHUNKS + NTHREADS - 1) / NTHREADS;     // chunks per thread per slice: 3 | 6
    constexpr bool HOLD = NW == 4;                       // per-thread offsets held in registers (there is room) or re-formed at each use
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rp = wave & 3;            // tile rows 2rp, 2rp+1 (halo rows 2rp .. 2rp+3)
    static_assert(!O16 || (NW == 8 && !EXTRA), "O16: 8-wave form without the scalar channel");
    static_assert(!XOUT || (NW == 8 && EXTRA && RBK == 0), "XOUT: 8-wave form with the scalar channel, no read-back streams");
    // the products of a depth transform folded into the output planes as soon as it is complete: 32 accumulator registers less on paper;
    // measured neutral to ~2 % slower, and the allocator spills MORE in the 33 -> 33 instance with it (62 vs 32 registers)
    constexpr bool FOLD = ESTD_W2FOLD != 0;
    const int cw = wave >> 2;                           // O16: this wave's input-channel chunk and the output plane (d0 + cw) it finishes
    const int nh0 = (NW == 8 && !O16) ? wave >> 2 : 0;  // first channel half of this wave
    constexpr int NSTEPS = O16 ? 12 : 24;
    constexpr int TAP_BYTES = O16 ? 2048 : 4096;
    constexpr int WTAPS = O16 ? O16_WLDS_TAPS : XOUT ? XOUT_WLDS_TAPS : WLDS_TAPS;
    const int g = lane >> 4;            // k index inside an MFMA
    const int i = lane & 15;            // voxel column of the (transposed) MFMA
    // MFMA column <-> voxel of a tile row (conflict-free ds_read_b128 for every tap; see csrc/conv3d_wino.hip)
    const int pi = i < 4 ? 2 * i : i < 12 ? 2 * i - 7 : 2 * i - 16;
    const int D = p.D, H = p.H, W = p.W;
    const int HW = H * W;
    const size_t vol = (size_t)D * HW;

    // ---- range of the flattened tile list (column-major: the d pairs of one (n, h-tile, w-tile) column are consecutive) ----
    int u, u_end;
    {
        const int G = gridDim.x, bid = blockIdx.x;
        const int r = ((G & 7) == 0) ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;       // XCD x owns a contiguous block of ranges
        u = (int)((long long)total_tiles * r / G);
        u_end = (int)((long long)total_tiles * (r + 1) / G);
    }
    if (u >= u_end) return;

    float* lds_ss = reinterpret_cast<float*>(smem + 4 * SLICE_BYTES + RED_BYTES);  // scale[32] | shift[32]: read in the epilogue
    if (tid < 64) {                                                               // (a global load there is an exposed L2 round trip)
        const int c = tid & 31;
        if (!(GATE && c >= 16)) lds_ss[tid] = (O16 && c >= 16) ? 0.0f : (tid < 32 ? p.scale[c] : p.shift[c]);      // (GATE: the upper halves hold the gate constants)
    }
    // activation as a per-channel floor: ReLU = max(v, 0), none = max(v, -inf) -- two VALU operations per value in the epilogue
    // (fp32 MFMAs hide no VALU work: every epilogue instruction is paid in matrix-pipe time); tanh takes the generic path
    if (tid >= 64 && tid < 96) lds_ss[tid] = ((tid - 64) < p.act_split ? p.act_a : p.act_b) == ESTD_ACT_RELU ? 0.0f : ESTD_NO_FLOOR;
    // (O16: scale / shift hold 16 entries; lanes read channels 0..15 only)
    const bool any_tanh = p.act_a == ESTD_ACT_TANH || p.act_b == ESTD_ACT_TANH;                     // uniform
    const bool tanh_quads = any_tanh && (p.act_split & 3) == 0 && !ESTD_W2_LIBM_TANH;                // uniform
    unsigned* lds_vt = reinterpret_cast<unsigned*>(smem + 4 * SLICE_BYTES + RED_BYTES + SS_BYTES);     // [it][thread]
    float* lds_x = reinterpret_cast<float*>(smem + 4 * SLICE_BYTES + RED_BYTES + SS_BYTES + VTAB_BYTES);          // [4][SL_VOX] (EXTRA)
    char* lds_xch = smem + 4 * SLICE_BYTES + RED_BYTES + SS_BYTES + VTAB_BYTES;               // O16: [8 waves][2 rows][64 lanes] float4
    char* lds_w = smem + 4 * SLICE_BYTES + RED_BYTES + SS_BYTES + VTAB_BYTES + (O16 ? O16_XCH_BYTES : XSL_BYTES);     // weights of the first taps
    constexpr int WT0 = O16 ? 0 : WLDS_T0;                                        // first LDS-resident tap
    char* lds_wxo = lds_w + WTAPS * TAP_BYTES;                                    // XOUT: the 33rd output channel's weights ...
    float* lds_xo = reinterpret_cast<float*>(lds_wxo + XOUT_W_BYTES);             // ... and the exchange buffer of its cross-wave sum
    if (XOUT) {
        for (int e = tid; e < XOUT_W_BYTES / 16; e += NTHREADS)
            reinterpret_cast<float4*>(lds_wxo)[e] = reinterpret_cast<const float4*>(p.w_xout)[e];
    }
    for (int e = tid; e < WTAPS * TAP_BYTES / 16; e += NTHREADS)                  // (visible after the first tile's barriers)
        reinterpret_cast<float4*>(lds_w)[e] = reinterpret_cast<const float4*>(p.w_wino2)[WT0 * (TAP_BYTES / 16) + e];

    // packed weights: [48 taps][2 halves][2 quads][64 lanes][4]
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w_wino2, (size_t)NTAPS * (O16 ? 1 : 2) * 2 * 256);
    const int wlane = lane * 16 + nh0 * 2048;
    const int row0 = 2 * rp;
    int rbase[3][2];                     // column-keyed swizzle: byte offset of (halo row row0, column kw + pi, chunk g + 4c) of slice 0
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int c = 0; c < 2; ++c) rbase[kw][c] = row0 * (IN_W * 128) + lds_colkey_off(kw + pi, g + 4 * c);

    // GATE: this thread's chunk of a voxel record is tid & 7 for every slice chunk it owns (NTHREADS is a multiple of 8); chunks 4..7 = h
    const bool gate_lane = GATE && (tid & 4) != 0;
    // sigmoid(GN(r)) = 1 / (1 + exp2(r * a + b)) with a = -log2(e) * rstd * gamma, b = -log2(e) * (beta - mean * rstd * gamma): the 16 (a, b) pairs
    // lie in the upper halves of the scale / shift arrays of lds_ss, which the 16-output-channel instance does not use
    if (GATE && tid < 16) {
        const float mean = p.gate_stats[0], rstd = p.gate_stats[1];
        const float ga = rstd * p.gate_gamma[tid], gb = p.gate_beta[tid] - mean * ga;
        lds_ss[16 + tid] = -1.4426950408889634f * ga;
        lds_ss[48 + tid] = -1.4426950408889634f * gb;
    }
    auto gate_chunk = [&](float4 x, float4 r) {
        const float4 a4 = *reinterpret_cast<const float4*>(lds_ss + 16 + (tid & 3) * 4), b4 = *reinterpret_cast<const float4*>(lds_ss + 48 + (tid & 3) * 4);
        return make_float4(x.x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(r.x, a4.x, b4.x))),
                           x.y * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(r.y, a4.y, b4.y))),
                           x.z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(r.z, a4.z, b4.z))),
                           x.w * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(fmaf(r.w, a4.w, b4.w))));
    };
    int tl_tile = 0;        // (timeline builds) tiles done by this workgroup
#if ESTD_W2PRIO == 2
    if (NW == 8 && nh0 != 0) __builtin_amdgcn_s_setprio(1);
#endif
    while (u < u_end) {
        // ---- column segment [u, seg_end): same (n, h-tile, w-tile), consecutive depth pairs ----
        const int col = u / dpairs;
        int dp = u - col * dpairs;
        const int twi = col % tiles_w, c2 = col / tiles_w;
        const int thi = c2 % tiles_h, n = c2 / tiles_h;
        const int tw0 = twi * TW, th0 = thi * TH;
        const int seg_end = min(u_end, (col + 1) * dpairs);

        const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in_main + (size_t)n * vol * p.in_stride, vol * p.in_stride);
        __amdgpu_buffer_rsrc_t rs_out = rs_in, rs_res = rs_in, rs_res2 = rs_in;
        rs_out = make_rsrc(p.out_main + (size_t)n * vol * p.out_stride, vol * p.out_stride);
        if (p.residual) rs_res = make_rsrc(p.residual + (size_t)n * vol * p.out_stride, vol * p.out_stride);
        if (p.residual2) rs_res2 = make_rsrc(p.residual2 + (size_t)n * vol * p.out_stride, vol * p.out_stride);
        const int in_slice_bytes = HW * p.in_stride * 4;
        const int out_plane_bytes = HW * p.out_stride * 4;
        __amdgpu_buffer_rsrc_t rs_ex = rs_in;
        if (EXTRA) rs_ex = make_rsrc(p.in_extra + (size_t)n * vol, vol);

        // per-thread slice elements (validity in y / x does not depend on d): chunk it of a slice = chunk tid + it * NTHREADS.
        // Global offsets are held per column segment; the LDS offset of chunk it is loff0 + it * NTHREADS * 16 exactly (the
        // swizzle key (voxel >> 1) & 7 does not change when the voxel advances by NTHREADS / 8), i.e. one register + immediates.
        // Global offsets: a per-thread table in LDS, rewritten per column segment (own entries only: no barrier needed) -- held in
        // registers across the tap loop they spill, and a scratch reload in front of a prefetch waits for every weight load before it.
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int e = tid + it * NTHREADS;
            const int vs = e >> 3, c = e & 7;
            const int zy = vs / IN_W, zx = vs % IN_W;
            const int gy = th0 - 1 + zy, gx = tw0 - 1 + zx;
            const bool ok = e < SL_CHUNKS && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            lds_vt[it * NTHREADS + tid] = ok ? (unsigned)((gy * W + gx) * p.in_stride + c * 4) * 4u : OOB_OFFSET;
        }
        if (EXTRA) {                                     // the scalar channel: thread t < 180 owns voxel t of its haloed slices
            const int zy = tid / IN_W, zx = tid % IN_W;
            const int gy = th0 - 1 + zy, gx = tw0 - 1 + zx;
            if (tid < 512) lds_vt[6 * 256 + tid] = (tid < SL_VOX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) ? (unsigned)(gy * W + gx) * 4u : OOB_OFFSET;
        }
        auto x_voff = [&]() {
            const int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            return lds_vt[6 * 256 + wave * 64 + l];
        };
        auto load_x = [&](int pd, unsigned vo) {
            return (unsigned)pd < (unsigned)D ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ex, vo, pd * HW * 4, 0)) : 0.0f;
        };
        // (the thread's table slot is re-formed from the lane id where it is read: nothing is held across the tap loop)
        auto chunk_voff = [&](int it) {
            const int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            return lds_vt[it * NTHREADS + wave * 64 + l];
        };
        const int loff0 = lds_chunk_off(tid >> 3, tid & 7);
        int loffk[SIT];                                  // column-keyed swizzle: the key changes with the chunk's column
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int vs = (tid >> 3) + it * (NTHREADS / 8);
            loffk[it] = (vs / IN_W) * (IN_W * 128) + lds_colkey_off(vs % IN_W, tid & 7);
        }
        const bool last_ok = tid + (SIT - 1) * NTHREADS < SL_CHUNKS;      // the last chunk of a slice exists for this thread
        auto load_plane = [&](int pd, float4 (&dst)[SIT]) {
            const bool pv = (unsigned)pd < (unsigned)D;        // wave-uniform; planes outside the volume are zero padding
#pragma unroll
            for (int it = 0; it < SIT; ++it)
                dst[it] = pv ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, chunk_voff(it), pd * in_slice_bytes, ESTD_W2_AUX_IN))
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        };

        __amdgpu_buffer_rsrc_t rs_gate = rs_in;
        if (GATE) rs_gate = make_rsrc(p.gate_r + (size_t)n * vol * 32, vol * 32);
        // GATE: the r chunk that belongs to this thread's h chunk lies 64 bytes in front of it in a record of the same stride (in_stride = 32)
        auto gate_voff = [&](unsigned vo) { return gate_lane ? vo - 64u : OOB_OFFSET; };
        auto load_plane_r = [&](int pd, float4 (&dst)[SIT]) {
            const bool pv = (unsigned)pd < (unsigned)D;
#pragma unroll
            for (int it = 0; it < SIT; ++it)
                dst[it] = pv ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_gate, gate_voff(chunk_voff(it)), pd * (HW * 32 * 4), 0))
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        auto gate_plane = [&](float4 (&x)[SIT], const float4 (&r)[SIT]) {
            if (gate_lane) {
#pragma unroll
                for (int it = 0; it < SIT; ++it) x[it] = gate_chunk(x[it], r[it]);
            }
        };
        // The MFMAs run TRANSPOSED (weights as the A operand, voxels as B): D row = output channel, D column = voxel, so lane
        // (g, i) holds the four consecutive channels 16nh + 4g .. +3 of voxel pi(i) of a tile row -- one 16-byte store (and one
        // 16-byte read per residual stream) per plane, row and channel half instead of four 4-byte ones.
        auto eoff_f = [&](int m) {                     // channel half nh0; the second half of a 4-wave lane is 64 bytes further
            const int l = lane;
            const int ii = l & 15;
            const int y = th0 + row0 + m, x = tw0 + (ii < 4 ? 2 * ii : ii < 12 ? 2 * ii - 7 : 2 * ii - 16);
            return (y < H && x < W) ? (unsigned)((y * W + x) * p.out_stride + 16 * nh0 + 4 * (l >> 4)) * 4u : OOB_OFFSET;
        };
        unsigned eoff_h[2];                            // segment constants
        eoff_h[0] = eoff_f(0);
        eoff_h[1] = eoff_f(1);
        auto eoff_of = [&](int m) { return eoff_h[m]; };
        auto bn_act = [&](const f32x4& a, int cb, float4& v) {
            const float4 sc4 = *reinterpret_cast<const float4*>(lds_ss + cb), sh4 = *reinterpret_cast<const float4*>(lds_ss + 32 + cb);
            if (!any_tanh) {
                const float4 lo = *reinterpret_cast<const float4*>(lds_ss + 64 + cb);
                v.x = fmaxf(a[0] * sc4.x + sh4.x, lo.x);
                v.y = fmaxf(a[1] * sc4.y + sh4.y, lo.y);
                v.z = fmaxf(a[2] * sc4.z + sh4.z, lo.z);
                v.w = fmaxf(a[3] * sc4.w + sh4.w, lo.w);
                return;
            }
            if (tanh_quads) {      // the four channels of a lane share the activation (split is a multiple of 4): tanh from the exp / rcp units
                const float4 lo = *reinterpret_cast<const float4*>(lds_ss + 64 + cb);
                const float u0 = a[0] * sc4.x + sh4.x, u1 = a[1] * sc4.y + sh4.y, u2 = a[2] * sc4.z + sh4.z, u3 = a[3] * sc4.w + sh4.w;
                if ((cb < p.act_split ? p.act_a : p.act_b) == ESTD_ACT_TANH) {
                    v.x = tanh_fast(u0); v.y = tanh_fast(u1); v.z = tanh_fast(u2); v.w = tanh_fast(u3);
                } else {
                    v.x = fmaxf(u0, lo.x); v.y = fmaxf(u1, lo.y); v.z = fmaxf(u2, lo.z); v.w = fmaxf(u3, lo.w);
                }
                return;
            }
            v.x = act_apply(a[0] * sc4.x + sh4.x, cb + 0 < p.act_split ? p.act_a : p.act_b);
            v.y = act_apply(a[1] * sc4.y + sh4.y, cb + 1 < p.act_split ? p.act_a : p.act_b);
            v.z = act_apply(a[2] * sc4.z + sh4.z, cb + 2 < p.act_split ? p.act_a : p.act_b);
            v.w = act_apply(a[3] * sc4.w + sh4.w, cb + 3 < p.act_split ? p.act_a : p.act_b);
        };
        // epilogue of one plane: tile rows row0, row0 + 1 (m), channel halves (x).  Every read-back stream (residuals, the running
        // sum) issues its loads back to back and is waited for ONCE.
        struct EpiLoads { float4 r1[2][NHW], r2[2][NHW], ro[2][NHW]; };
        auto epi_issue = [&](int dd, EpiLoads& L) {       // the read-back streams of one plane, all loads back to back
            const int so = dd * out_plane_bytes;
            if (RB_RES && p.residual) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int x = 0; x < NHW; ++x) L.r1[m][x] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_res, eoff_of(m), so + 64 * x, 0));
            }
            if (RB_RES && p.residual2) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int x = 0; x < NHW; ++x) L.r2[m][x] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_res2, eoff_of(m), so + 64 * x, 0));
            }
            if (RB_ACC && p.accumulate) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int x = 0; x < NHW; ++x) L.ro[m][x] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_out, eoff_of(m), so + 64 * x, 0));
            }
        };
        auto epi_finish = [&](const f32x4 (&a)[2][NHW], int dd, const EpiLoads& L) {
            const int so = dd * out_plane_bytes;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int x = 0; x < NHW; ++x) {
                    float4 v;
                    bn_act(a[m][x], 16 * (nh0 + x) + 4 * g, v);
                    if (RB_RES && p.residual) v = f4_add(v, L.r1[m][x]);
                    if (RB_RES && p.residual2) v = f4_add(v, L.r2[m][x]);
                    if (RB_RES) v = make_float4(v.x * p.out_scale, v.y * p.out_scale, v.z * p.out_scale, v.w * p.out_scale);
                    if (RB_ACC && p.accumulate) v = f4_add(v, L.ro[m][x]);
                    u32x4 bits;
                    __builtin_memcpy(&bits, &v, 16);
                    if (!(ESTD_W2ABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(bits, rs_out, eoff_of(m), so + 64 * x, ESTD_W2_AUX_OUT);
                }
        };
        // epilogue of one plane: tile rows row0, row0 + 1 (m), channel halves (x).  Every read-back stream (residuals, the running
        // sum) issues its loads back to back and is waited for ONCE.
        auto epi_plane = [&](const f32x4 (&a)[2][NHW], int dd) {
            EpiLoads L;
            epi_issue(dd, L);
            epi_finish(a, dd, L);
        };
        // GroupNorm(1 group) partial sums of the raw outputs of BOTH planes of a tile: group = channel half.  Fixed-order reduction ->
        // deterministic: the 16 lanes of a DPP row by four DPP-modified moves + adds (no LDS round trip), the four rows by two
        // ds_bpermute levels, the four row pairs of a half through LDS.  One barrier pair per tile.  Workgroup-uniform call.
        // (Round 3: one call per plane with a six-level ds_bpermute butterfly on doubles and its own barrier pair cost the gate
        // convolution 15 %: 0.33 ms against 0.287 ms for the same launch without statistics.)
        int stats_parity = 0;
        auto wave_sum4 = [&](double (&v)[4]) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = row16_sum_f64(v[k]);
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1)
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] += __shfl_xor(v[k], o);
        };
        auto tile_stats = [&](const f32x4 (&a0)[2][NHW], const f32x4 (&a1)[2][NHW], int d0_) {
            // scratch [tile parity][channel half][row pair][sum0, sq0, sum1, sq1]: two copies, so that the only barrier is the one between
            // the writes and the final adds (the next write to a copy is two tiles = several barriers later)
            double* red = reinterpret_cast<double*>(smem + 4 * SLICE_BYTES) + (stats_parity ? 32 : 0);
            stats_parity ^= 1;
            double v[NHW][4];
#pragma unroll
            for (int x = 0; x < NHW; ++x) {
                v[x][0] = v[x][1] = v[x][2] = v[x][3] = 0.0;
                const int cb = 16 * (nh0 + x) + 4 * g;
                const float4 sc4 = *reinterpret_cast<const float4*>(lds_ss + cb), sh4 = *reinterpret_cast<const float4*>(lds_ss + 32 + cb);
                const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    if (eoff_of(m) != OOB_OFFSET) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const double u0 = (double)(a0[m][x][r] * scv[r] + shv[r]), u1 = (double)(a1[m][x][r] * scv[r] + shv[r]);
                            v[x][0] += u0; v[x][1] += u0 * u0; v[x][2] += u1; v[x][3] += u1 * u1;
                        }
                    }
                wave_sum4(v[x]);
            }
            if (lane == 0) {
#pragma unroll
                for (int x = 0; x < NHW; ++x)
#pragma unroll
                    for (int k = 0; k < 4; ++k) red[((nh0 + x) * 4 + rp) * 4 + k] = v[x][k];
            }
            __syncthreads();
            if (tid < 8) {                                       // (plane, channel half, {sum, sumsq})
                const int pl_ = tid >> 2, grp = (tid >> 1) & 1, q = tid & 1;
                if (d0_ + pl_ < D) {                             // (odd D: the last pair has one plane)
                    const int k = pl_ * 2 + q;
                    const double tot = red[(grp * 4 + 0) * 4 + k] + red[(grp * 4 + 1) * 4 + k] + red[(grp * 4 + 2) * 4 + k] + red[(grp * 4 + 3) * 4 + k];
                    // partial index = canonical tile id (n, d, thi, twi), as the direct kernel writes it
                    const size_t tile_id = (((size_t)n * D + d0_ + pl_) * tiles_h + thi) * tiles_w + twi;
                    p.stats_partials[tile_id * 4 + grp * 2 + q] = tot;
                }
            }
        };

        // O16: one 16-channel group; wave (rp, cw) holds plane d0 + cw -> both planes of the tile in ONE reduction: group slot 0 of
        // the plane's canonical tile id receives {sum, sumsq}, slot 1 zeros (estd_groupnorm_finalize reads both groups).
        auto plane_stats_o16 = [&](const f32x4 (&a)[2][NHW], int d0_) {
            double v[4] = {0.0, 0.0, 0.0, 0.0};
            const int cb = 4 * g;
            const float4 sc4 = *reinterpret_cast<const float4*>(lds_ss + cb), sh4 = *reinterpret_cast<const float4*>(lds_ss + 32 + cb);
            const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
            for (int m = 0; m < 2; ++m)
                if (eoff_of(m) != OOB_OFFSET) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const double u = (double)(a[m][0][r] * scv[r] + shv[r]); v[0] += u; v[1] += u * u; }
                }
            v[0] = row16_sum_f64(v[0]); v[1] = row16_sum_f64(v[1]);
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) { v[0] += __shfl_xor(v[0], o); v[1] += __shfl_xor(v[1], o); }
            double* red = reinterpret_cast<double*>(smem + 4 * SLICE_BYTES) + (stats_parity ? 32 : 0);       // [tile parity][plane cw][row pair][sum, sumsq]
            stats_parity ^= 1;
            if (lane == 0) { red[(cw * 4 + rp) * 2] = v[0]; red[(cw * 4 + rp) * 2 + 1] = v[1]; }
            __syncthreads();
            if (tid < 4) {
                const int pl_ = tid >> 1, q = tid & 1;
                if (d0_ + pl_ < D) {
                    const double tot = red[(pl_ * 4 + 0) * 2 + q] + red[(pl_ * 4 + 1) * 2 + q] + red[(pl_ * 4 + 2) * 2 + q] + red[(pl_ * 4 + 3) * 2 + q];
                    const size_t tile_id = (((size_t)n * D + d0_ + pl_) * tiles_h + thi) * tiles_w + twi;
                    p.stats_partials[tile_id * 4 + q] = tot;
                    p.stats_partials[tile_id * 4 + 2 + q] = 0.0;
                }
            }
        };

        // raw planes in registers: xa = x[d0-1], xb = x[d0], xc = x[d0+1], xd = x[d0+2]
        float4 xa[SIT], xb[SIT], xc[SIT], xd[SIT];
        {
            const int d0 = 2 * dp;
            load_plane(d0 - 1, xa);
            load_plane(d0, xb);
            load_plane(d0 + 1, xc);
            load_plane(d0 + 2, xd);
            if (GATE) {
                float4 ra[SIT], rb[SIT];
                __syncthreads();                          // (the gate constants in lds_ss: written by threads 0..15 at kernel start)
                load_plane_r(d0 - 1, ra); load_plane_r(d0, rb);
                gate_plane(xa, ra); gate_plane(xb, rb);
                load_plane_r(d0 + 1, ra); load_plane_r(d0 + 2, rb);
                gate_plane(xc, ra); gate_plane(xd, rb);
            }
        }
        float4 rc[SIT], rd[SIT];                         // GATE: the r chunks of the two planes in flight
        float ea = 0.f, eb = 0.f, ec = 0.f, ed = 0.f;    // EXTRA: the scalar channel's four planes at this thread's voxel
        if (EXTRA) {
            const int d0 = 2 * dp;
            const unsigned vo = x_voff();
            ea = load_x(d0 - 1, vo); eb = load_x(d0, vo); ec = load_x(d0 + 1, vo); ed = load_x(d0 + 2, vo);
        }
        auto write_x_slices = [&]() {
            if (EXTRA && tid < SL_VOX) {
                lds_x[0 * SL_VOX + tid] = ea - ec;
                lds_x[1 * SL_VOX + tid] = eb + ec;
                lds_x[2 * SL_VOX + tid] = ec - eb;
                lds_x[3 * SL_VOX + tid] = eb - ed;
            }
        };

        // depth transform B^T x of the planes in (xa, xb, xc, xd), straight into LDS slice sl
        auto write_slice = [&](int sl) {
#pragma unroll
            for (int it = 0; it < SIT; ++it) {
                if ((it < SIT - 1 || last_ok) && !(ESTD_W2ABL & 2)) {
                    const float4 v = sl == 0 ? f4_sub(xa[it], xc[it]) : sl == 1 ? f4_add(xb[it], xc[it])
                                   : sl == 2 ? f4_sub(xc[it], xb[it]) : f4_sub(xb[it], xd[it]);
                    if (ESTD_W2COLKEY) *reinterpret_cast<float4*>(smem + loffk[it] + sl * SLICE_BYTES) = v;
                    else *reinterpret_cast<float4*>(smem + loff0 + sl * SLICE_BYTES + it * NTHREADS * 16) = v;
                }
            }
        };
        auto shift_planes = [&]() {                      // planes d0+1, d0+2 are planes d0'-1, d0' of the next tile
#pragma unroll
            for (int it = 0; it < SIT; ++it) { xa[it] = xc[it]; xb[it] = xd[it]; }
            if (EXTRA) { ea = ec; eb = ed; }
        };
        bool first = true;
        // DEFER: the outputs of the previous tile of this column segment, stored inside the first steps of the current one -- between
        // the last MFMA of a tile and the first of the next there is then ONE barrier and the prologue (measured before: ~6 000 of
        // 35 500 cycles per tile without a single MFMA: two barriers, the slice-3 rewrite and the epilogue of all eight waves at once,
        // profiles/r3_wino2_tile_timeline.txt).  Their registers are the ones the next-plane prefetch occupies later in the loop.
#ifndef ESTD_W2_RB_DEFER
#define ESTD_W2_RB_DEFER 3      // bit 0: deferred epilogue for RBK = 1 (0.969 -> 0.946 ms, 34 spilled VGPRs), bit 1: for RBK = 2 (1.04 -> 1.09 ms: 43 spilled, off)
#endif
        constexpr bool DEFER = ESTD_W2DEFER != 0 && (RBK == 0 || (RBK == 1 && (ESTD_W2_RB_DEFER & 1)) || (RBK == 2 && (ESTD_W2_RB_DEFER & 2)) || (RBK == 3 && (ESTD_W2_RB_DEFER & 4)));   // (the generic read-back instance spills in the deferred form)
        // step in front of which slices 0..2 are rewritten (every read of them has been issued: rows are fetched two steps ahead)
        constexpr int RB_STEP = O16 ? (DEFER ? 7 : 9) : (DEFER ? 16 : 18);
#ifndef ESTD_W2PF_STEP
#define ESTD_W2PF_STEP 4    // (A/B) >= 4: the deferred outputs live in the prefetch registers through steps 0..3; <= 10: six steps of requests in front of the step-16 rewrite
#endif
        constexpr int PF_STEP = (DEFER && !O16) ? ESTD_W2PF_STEP : 0; // first step of the next-plane prefetch
        f32x4 py0[2][NHW], py1[2][NHW];
        int pd0 = 0;
        bool have_prev = false;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int x = 0; x < NHW; ++x) { py0[m][x] = (f32x4){0.f, 0.f, 0.f, 0.f}; py1[m][x] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

        for (; u < seg_end; ++u, ++dp, ++tl_tile) {
            const int d0 = 2 * dp;
            W2STAMP(0);
            if (first) {
                lds_barrier();                          // every wave is done reading the previous tile's slices
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) write_slice(sl);
                write_x_slices();
                shift_planes();
                lds_barrier();
                first = false;
            }

            const bool has_next = (u + 1 < seg_end);     // wave-uniform
            const int nd = d0 + 3;                       // new planes of the next tile: nd, nd + 1
            const bool v0 = nd < D, v1 = nd + 1 < D;

            // m[sd][sh] per channel half.  No zero fill (64 register moves per tile and wave, paid in matrix time, §3.0 of DESIGN.md): the
            // first product of every accumulator -- column tap 0, channel chunk 0, k-step 0 of its depth transform -- takes C = 0.
            // FOLD (round 4): only the four products m[sd][0..3] of the CURRENT depth transform are held (the steps run sd-major);
            // behind the last step of a depth transform they go through the row half of the output transform and are added into the two
            // output planes (y0 = z0 + z1 + z2, y1 = z1 - z2 - z3) -- 16 + 16 accumulator registers per channel half instead of 64.
            f32x4 acc[FOLD ? 1 : 4][4][NHW];
            f32x4 y0[2][NHW], y1[2][NHW];
            auto fold_sd = [&](int sd_, const f32x4 (&a)[4][NHW]) {
#pragma unroll
                for (int x = 0; x < NHW; ++x) {
                    const f32x4 z0 = a[0][x] + a[1][x] + a[2][x], z1 = a[1][x] - a[2][x] - a[3][x];
                    if (sd_ == 0) { y0[0][x] = z0; y0[1][x] = z1; }
                    else if (sd_ == 1) { y0[0][x] += z0; y0[1][x] += z1; y1[0][x] = z0; y1[1][x] = z1; }
                    else if (sd_ == 2) { y0[0][x] += z0; y0[1][x] += z1; y1[0][x] -= z0; y1[1][x] -= z1; }
                    else { y1[0][x] -= z0; y1[1][x] -= z1; }
                }
            };

            auto load_w = [&](int t, int q, int x) {        // t, x are compile-time constants after unrolling (q too, except O16: q = cw)
                if (t >= WT0 && t < WT0 + WTAPS) return *reinterpret_cast<const float4*>(lds_w + (t - WT0) * TAP_BYTES + x * 2048 + q * 1024 + wlane);
                return as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane + (O16 ? q * 1024 : 0), t * TAP_BYTES + x * 2048 + (O16 ? 0 : q * 1024), 0));
            };
            // 16-byte chunk c (channels 4g.. for c = 0, 16+4g.. for c = 1) of halo row 2rp + r at column shift kw of depth slice sd
            auto load_row = [&](int sd, int kw, int c, int r) {
                if (ESTD_W2COLKEY)
                    return *reinterpret_cast<const float4*>(smem + rbase[kw][c] + sd * SLICE_BYTES + r * (IN_W * 128));
                const int vs = (row0 + r) * IN_W + kw + pi;
                int off = sd * SLICE_BYTES + lds_chunk_off(vs, g);
                if (c) off ^= 64;
                return *reinterpret_cast<const float4*>(smem + off);
            };
            // transform components 2h, 2h+1 of the four row fragments from the raw rows: PACKED adds (v_pk_add_f32).  The fp32 MFMA
            // runs at the f32 vector rate on the same lanes, so VALU work is not hidden behind it -- measured: kernel time is linear
            // in the VALU instruction count, ~7 SIMD cycles per instruction -- and a packed add transforms two operands per slot.
            auto xform2 = [&](const float4 (&Rr)[4], int h, f32x2 (&o)[4]) {
                f32x2 r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) r[k] = h == 0 ? (f32x2){Rr[k].x, Rr[k].y} : (f32x2){Rr[k].z, Rr[k].w};
                if (ESTD_W2ABL & 128) { o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; }
                else if (ESTD_W2PK == 0) { o[0] = r[0] - r[2]; o[1] = r[1] + r[2]; o[2] = r[2] - r[1]; o[3] = r[1] - r[3]; }
                else {
                    // inline assembly: left to itself the compiler UNPACKS packed adds next to MFMAs into two plain ones (a
                    // heuristic for the bf16 matrix pipe, which overlaps plain VALU work).  Measured: forcing them packed is
                    // SLOWER (0.938 vs 0.913 ms), with or without fences -- kept as an A/B switch.
                    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o[0]) : "v"(r[0]), "v"(r[2]));
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(o[1]) : "v"(r[1]), "v"(r[2]));
                    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o[2]) : "v"(r[2]), "v"(r[1]));
                    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(o[3]) : "v"(r[1]), "v"(r[3]));
                }
            };
            auto load_rows = [&](int st, float4 (&Rr)[4]) {
                const int ng = O16 ? st : st >> 1;
#pragma unroll
                for (int r = 0; r < 4; ++r) Rr[r] = load_row(ng / 3, ng % 3, O16 ? cw : (st & 1), r);
            };
            auto load_b = [&](int st, float4 (&bq)[4][NHW]) {
                const int ng = O16 ? st : st >> 1, nc = O16 ? cw : (st & 1);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int x = 0; x < NHW; ++x) bq[t][x] = load_w(4 * ng + t, nc, x);
            };

            // software pipeline inside the wave: step s multiplies the fragments T (transformed during step s-1) while the VALU
            // transforms the raw rows of step s+1 (read during step s-1) two components at a time -- T's registers are recycled as
            // the MFMAs consume them -- and then reads the raw rows of step s+2.  Weights: one step ahead (8 waves), two steps ahead
            // with one wave per SIMD (nobody covers an L2 round trip there, and the registers are free).
#ifndef ESTD_W2BD
#define ESTD_W2BD 3     // since the column-keyed swizzle the registers are there (242 VGPRs, no spill): N = 3 plain 0.819 -> 0.810 ms, + running sum 0.853 -> 0.838,
#endif                  // + residual 0.862 -> 0.847, two residuals 0.911 -> 0.893, 33 -> 32 0.861 -> 0.850, 32 -> 16 0.158 -> 0.155 (profiles/r4_wino2_ablation.txt)
            // the 33 -> 33 instance keeps one step of cover: with two it spills (0.982 -> 1.124 ms)
            // ... and so does the reset-gated 32 -> 16 instance (round 6): with two steps of cover it carries 16 spilled registers reloaded once per tile
            // (profiles/r6_kernel_resources.txt); with one, none -- one ConvGRU 0.5297 -> 0.5265 ms, Joint step -0.04 ms (three alternating pairs)
            constexpr int BD = NW == 4 ? 3 : (XOUT || (O16 && GATE)) ? 2 : ESTD_W2BD;  // weight buffers in flight
            constexpr bool QSCHED = ESTD_W2_QSCHED != 0 && NW == 8;
            float4 bq[BD][4][NHW];                       // [buffer][sh][channel half]: one step's quad of the four taps of a group
            f32x2 T[2][4];                               // [component pair][sh]
            float4 R[4];
            load_b(0, bq[0]);
            if (BD == 3) load_b(1, bq[1]);
            load_rows(0, R);
            xform2(R, 0, T[0]);
            xform2(R, 1, T[1]);
            load_rows(1, R);
            constexpr int PER = SIT / 3;                 // plane chunks per step: the 2 x SIT chunks go out in six steps
            unsigned vo_next[PER];
            if (has_next && PF_STEP == 0 && !(ESTD_W2ABL & 16)) {
#pragma unroll
                for (int k = 0; k < PER; ++k) vo_next[k] = chunk_voff(k % SIT);
            }
            EpiLoads pl;                                 // read-back loads of the deferred epilogue (issued one step before use)
#ifndef ESTD_W2_RB_EARLY
#define ESTD_W2_RB_EARLY 0      // A/B: non-deferred read-back instances request the streams of BOTH planes in the last two steps of the tap loop
#endif                          // (the raw rows / next-step transforms are dead there), the epilogue behind the loop finds them on their way
            constexpr bool RB_EARLY = ESTD_W2_RB_EARLY != 0 && RB && !DEFER && !O16;
            EpiLoads el0, el1;
            // XOUT: partial products of the current depth transform per row-transform index (two running sums each: the products pair up
            // into v_pk_fma_f32), the [plane][row] sums behind both output transforms, and this step's four weight quads
            f32x2 xm[4], xP[2][2];
            float4 xw[4];
            __builtin_amdgcn_sched_barrier(0);
            W2STAMP(1);

#pragma clang loop unroll(full)
            for (int step = 0; step < NSTEPS; ++step) {  // step = (group gi = 3 sd + kw, channel chunk c); O16: the chunk is the wave's
                const int gi = O16 ? step : step >> 1;
                const int sd = gi / 3;
                if (step == NSTEPS / 4) W2STAMP(2);
                if (step == NSTEPS / 2) W2STAMP(3);
                if (step == 3 * NSTEPS / 4) W2STAMP(4);
                if (step == 3 * NSTEPS / 4 + 1) W2STAMP(5);
#ifndef ESTD_W2SPREAD
#define ESTD_W2SPREAD 0     // A/B: 1 = the rewrite of slices 0..2 one slice per step (steps RB_STEP .. RB_STEP + 2) instead of all nine writes at once
#endif
                if (GATE && has_next && step >= 2 && step < 2 + 2 * SIT) {      // the chunk requested two steps ago (steps 0..5 -> gated in steps 2..7 = RB_STEP)
                    const int idx = step - 2, it = idx % SIT;
                    if (gate_lane) {
                        if (idx < SIT) xc[it] = gate_chunk(xc[it], rc[it]);
                        else           xd[it] = gate_chunk(xd[it], rd[it]);
                    }
                }
                if (has_next && step == RB_STEP) {
                    // slices 0..2 have been read for the last time by every wave (the rows of step 17 are fetched at the end of step
                    // 15); DEFER: this barrier also publishes slice 3, rewritten at the top of this tile and first read at the end of
                    // step 16
                    lds_barrier();
                    write_slice(0);
                    if (!ESTD_W2SPREAD) {
                        write_slice(1);
                        write_slice(2);
                    }
                }
                if (ESTD_W2SPREAD && has_next && step == RB_STEP + 1) write_slice(1);
                if (ESTD_W2SPREAD && has_next && step == RB_STEP + 2) write_slice(2);
                if (XOUT && nh0 == (step & 1)) {         // (uniform branch) this wave's channel chunk: the 33rd output channel's weights of the step
#pragma unroll
                    for (int t = 0; t < 4; ++t) xw[t] = *reinterpret_cast<const float4*>(lds_wxo + ((step * 4 + t) * 4) * 16 + g * 16);
                }
                // weights of step + BD - 1
                if (!QSCHED && step + BD - 1 < NSTEPS && !(ESTD_W2ABL & 8)) load_b(step + BD - 1, bq[(step + BD - 1) % BD]);
                if (RB_EARLY && step == NSTEPS - 2) epi_issue(d0, el0);
                if (RB_EARLY && step == NSTEPS - 1 && d0 + 1 < D) epi_issue(d0 + 1, el1);
                // chunks of the NEXT tile's two new planes: spread over the first steps (their offsets were read from the LDS table
                // at the end of the previous step)
                if (has_next && !(ESTD_W2ABL & 16)) {
                    if (step >= PF_STEP && step < PF_STEP + 6) {
#pragma unroll
                        for (int k = 0; k < PER; ++k) {
                            const int idx = (step - PF_STEP) * PER + k, it = idx % SIT;
                            const unsigned vo = vo_next[k];
                            if (idx < SIT) xc[it] = v0 ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, nd * in_slice_bytes, ESTD_W2_AUX_IN))
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
                            else           xd[it] = v1 ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, (nd + 1) * in_slice_bytes, ESTD_W2_AUX_IN))
                                                       : make_float4(0.f, 0.f, 0.f, 0.f);
                            if (GATE) {
                                if (idx < SIT) rc[it] = v0 ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_gate, gate_voff(vo), nd * (HW * 32 * 4), 0))
                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
                                else           rd[it] = v1 ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_gate, gate_voff(vo), (nd + 1) * (HW * 32 * 4), 0))
                                                           : make_float4(0.f, 0.f, 0.f, 0.f);
                            }
                        }
                    }
                }
                if (EXTRA && has_next && (step == PF_STEP + 6 || step == PF_STEP + 7)) {      // the scalar channel's two new planes
                    const unsigned vo = x_voff();
                    if (step == PF_STEP + 6) ec = v0 ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ex, vo, nd * HW * 4, 0)) : 0.f;
                    else                     ed = v1 ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ex, vo, (nd + 1) * HW * 4, 0)) : 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);       // the loads above are issued BEFORE this step's MFMAs (left alone, the
                                                         // scheduler sinks them to the end of the step: no prefetch at all)
#if ESTD_W2PRIO == 1
                // the two waves of a SIMD take turns at priority 1, a step each: left to age-based arbitration the older wave
                // runs ahead and then idles at the step-18 barrier while the younger one, alone, cannot keep the matrix pipe busy
                if (NW == 8) { if (((step & 1) ^ nh0) != 0) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
                const int cur = (ESTD_W2ABL & 8) ? 0 : step % BD;
                f32x2 Tn[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {            // the products rotate: no MFMA waits for its own predecessor
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        if (QSCHED) {
                            // ESTD_W2_QSCHED (round 5, from csrc/conv2d_wino2.hip): the step in four quarters of 4 MFMAs, ONE weight request (tap t = quarter) of step
                            // + BD - 1 in front of each instead of four in a cluster in front of the step; a scheduling barrier pins the quarter
                            const int qk = 2 * h + e, tstep = step + BD - 1;
                            if (tstep < NSTEPS && !(ESTD_W2ABL & 8)) {
                                const int ng = O16 ? tstep : tstep >> 1, nc = O16 ? cw : (tstep & 1);
#pragma unroll
                                for (int x = 0; x < NHW; ++x) bq[tstep % BD][qk][x] = load_w(4 * ng + qk, nc, x);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int x = 0; x < NHW; ++x) {
                                const float4 b4 = bq[cur][t][x];
                                const float b = h == 0 ? (e == 0 ? b4.x : b4.y) : (e == 0 ? b4.z : b4.w);
                                const bool first_product = gi % 3 == 0 && (O16 || (step & 1) == 0) && h == 0 && e == 0;
                                const f32x4 c_in = first_product ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[FOLD ? 0 : sd][t][x];
                                acc[FOLD ? 0 : sd][t][x] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, T[h][t][e], c_in, 0, 0, 0);
                            }
                        if (QSCHED && e == 0) __builtin_amdgcn_sched_barrier(0);
                    }
                    if (ESTD_W2PK == 1) __builtin_amdgcn_sched_barrier(0);   // the MFMAs of two components, then the next step's 4 packed transforms
                    if (step + 1 < NSTEPS) xform2(R, h, Tn[h]);
                    if (ESTD_W2PK == 1 && h == 0) __builtin_amdgcn_sched_barrier(0);
                }
                if (step + 2 < NSTEPS) load_rows(step + 2, R);
                if (has_next && step + 1 >= PF_STEP && step + 1 < PF_STEP + 6 && !(ESTD_W2ABL & 16)) {
#pragma unroll
                    for (int k = 0; k < PER; ++k) vo_next[k] = chunk_voff(((step + 1 - PF_STEP) * PER + k) % SIT);
                }
                if (DEFER && !O16 && have_prev && step < 4) {     // the previous tile's epilogue: plane d0 in steps 0-1, plane d0 + 1 in steps 2-3
                    const bool second = step >= 2;
                    if (!second || pd0 + 1 < D) {
                        if ((step & 1) == 0) epi_issue(pd0 + (second ? 1 : 0), pl);
                        else epi_finish(second ? py1 : py0, pd0 + (second ? 1 : 0), pl);
                    }
                }
                if (DEFER && O16 && have_prev && step < 2 && pd0 + cw < D) {   // O16: this wave's ONE plane of the previous tile
                    if (step == 0) epi_issue(pd0 + cw, pl);
                    else epi_finish(py0, pd0 + cw, pl);
                }
                if (ESTD_W2PK == 0 && !QSCHED) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {        // order of the region: the MFMAs of two components, then the next step's 4 transforms
                        __builtin_amdgcn_sched_group_barrier(0x008, 8 * NHW, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                }
                // (the FMAs at the START of the step -- T is live there anyway -- with the weights read one step earlier: 59 spilled registers
                // instead of 32, 1.06 instead of 0.98 ms)
                if (XOUT && nh0 == (step & 1)) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const f32x2 wlo = {xw[t].x, xw[t].y}, whi = {xw[t].z, xw[t].w};
                        xm[t] = gi % 3 == 0 ? T[0][t] * wlo : __builtin_elementwise_fma(T[0][t], wlo, xm[t]);
                        xm[t] = __builtin_elementwise_fma(T[1][t], whi, xm[t]);
                    }
                    if (gi % 3 == 2) {                   // depth transform sd complete (for this wave's chunk): row half, then depth half
                        const f32x2 r0 = xm[0] + xm[1] + xm[2], r1 = xm[1] - xm[2] - xm[3];
                        if (sd == 0) { xP[0][0] = r0; xP[0][1] = r1; }
                        else if (sd == 1) { xP[0][0] += r0; xP[0][1] += r1; xP[1][0] = r0; xP[1][1] = r1; }
                        else if (sd == 2) { xP[0][0] += r0; xP[0][1] += r1; xP[1][0] -= r0; xP[1][1] -= r1; }
                        else { xP[1][0] -= r0; xP[1][1] -= r1; }
                    }
                }
                if (step + 1 < NSTEPS) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int t = 0; t < 4; ++t) T[h][t] = Tn[h][t];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (FOLD && gi % 3 == 2 && (O16 || (step & 1) == 1)) {      // last step of depth transform sd
                    fold_sd(sd, acc[0]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }

            W2STAMP(6);
            // ---- the scalar input channel: one more k-step per product (lane group g = column tap kw = g) ----
            if (EXTRA) {
                // weights: [4 sd][2 halves][64 lanes][4 sh] (packing.pack_conv3d_wino2_extra), L2-resident
                const __amdgpu_buffer_rsrc_t rs_wx = make_rsrc(p.w_extra, (size_t)4 * 2 * 256);
                const int kwc = g < 3 ? g : 2;               // (g = 3 carries zero weights: any in-slice address)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    float4 wx[NHW];
#pragma unroll
                    for (int x = 0; x < NHW; ++x) wx[x] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_wx, lane * 16, ((s * 2 + nh0 + x) * 64) * 16, 0));
                    float r[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) r[k] = lds_x[s * SL_VOX + (row0 + k) * IN_W + kwc + pi];
                    const float t[4] = {r[0] - r[2], r[1] + r[2], r[2] - r[1], r[1] - r[3]};
                    if (FOLD) {
                        // the scalar channel's products of depth transform s on their own (C = 0) and folded like the others: the output
                        // transform is linear.  (They cannot join the tap loop: lds_x is published by the in-loop barrier only.)
                        f32x4 ax[4][NHW];
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int x = 0; x < NHW; ++x) {
                                const float w = q == 0 ? wx[x].x : q == 1 ? wx[x].y : q == 2 ? wx[x].z : wx[x].w;
                                ax[q][x] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, t[q], (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                            }
#pragma unroll
                        for (int x = 0; x < NHW; ++x) {
                            const f32x4 z0 = ax[0][x] + ax[1][x] + ax[2][x], z1 = ax[1][x] - ax[2][x] - ax[3][x];
                            if (s <= 2) { y0[0][x] += z0; y0[1][x] += z1; }
                            if (s == 1) { y1[0][x] += z0; y1[1][x] += z1; }
                            if (s >= 2) { y1[0][x] -= z0; y1[1][x] -= z1; }
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int x = 0; x < NHW; ++x) {
                                const float w = q == 0 ? wx[x].x : q == 1 ? wx[x].y : q == 2 ? wx[x].z : wx[x].w;
                                acc[s][q][x] = __builtin_amdgcn_mfma_f32_16x16x4f32(w, t[q], acc[s][q][x], 0, 0, 0);
                            }
                    }
                    if (XOUT && nh0 == 0) {              // (uniform) scalar input channel x 33rd output channel: column tap kw = g, once per row pair
                        const float4 ux = *reinterpret_cast<const float4*>(lds_wxo + 24 * 4 * 4 * 16 + (s * 4 + g) * 16);
                        const float m0 = t[0] * ux.x, m1 = t[1] * ux.y, m2 = t[2] * ux.z, m3 = t[3] * ux.w;
                        const float r0 = m0 + m1 + m2, r1 = m1 - m2 - m3;
                        if (s <= 2) { xP[0][0].x += r0; xP[0][1].x += r1; }
                        if (s == 1) { xP[1][0].x += r0; xP[1][1].x += r1; }
                        if (s >= 2) { xP[1][0].x -= r0; xP[1][1].x -= r1; }
                    }
                }
            }
            // ---- output transform A^T m A (FOLD: done, one depth transform at a time, inside the loop) ----
            if (!FOLD) {
#pragma unroll
                for (int x = 0; x < NHW; ++x) {
                    f32x4 z[4][2];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        z[s][0] = acc[s][0][x] + acc[s][1][x] + acc[s][2][x];
                        z[s][1] = acc[s][1][x] - acc[s][2][x] - acc[s][3][x];
                    }
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        y0[m][x] = z[0][m] + z[1][m] + z[2][m];
                        y1[m][x] = z[1][m] - z[2][m] - z[3][m];
                    }
                }
            }
            if (O16) {
                // cross-wave reduction over the two input-channel chunks: wave cw keeps plane d0 + cw and hands the other plane's
                // partial sums to its partner (wave ^ 4, same rows) through LDS; the barrier below is the tile's post-loop barrier
                float4* xw = reinterpret_cast<float4*>(lds_xch) + (wave * 2) * 64 + lane;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const f32x4 snd = cw == 0 ? y1[m][0] : y0[m][0];
                    xw[m * 64] = make_float4(snd[0], snd[1], snd[2], snd[3]);
                }
            }
            float xo = 0.f;
            if (XOUT) {
                // sum over the four lane groups (the k range of an MFMA), transposing on the way: lane group g ends up with value g of
                // (plane 0 row 0, plane 0 row 1, plane 1 row 0, plane 1 row 1) -- three shuffles instead of eight
                const float a = xP[0][0].x + xP[0][0].y, b = xP[0][1].x + xP[0][1].y, c = xP[1][0].x + xP[1][0].y, d = xP[1][1].x + xP[1][1].y;
                const bool g1 = (g & 1) != 0, g2 = (g & 2) != 0;
                const float k0 = (g1 ? b : a) + __shfl_xor(g1 ? a : b, 16);
                const float k1 = (g1 ? d : c) + __shfl_xor(g1 ? c : d, 16);
                xo = (g2 ? k1 : k0) + __shfl_xor(g2 ? k0 : k1, 32);
                if (nh0 == 1) lds_xo[((tl_tile & 1) * 4 + rp) * 64 + lane] = xo;       // (double-buffered by tile parity: the partner reads it behind the barrier)
            }
            if (has_next || O16 || XOUT) {
                lds_barrier();                            // every wave has read slice 3 for the last time; slices 0..2 (rewritten in the loop) are visible
            }
            if (XOUT && nh0 == 0) {
                xo += lds_xo[((tl_tile & 1) * 4 + rp) * 64 + lane];
                const int dd = d0 + (g >> 1), y = th0 + row0 + (g & 1), x = tw0 + pi;
                const int act2 = 32 < p.act_split ? p.act_a : p.act_b;
                if (dd < D && y < H && x < W)
                    p.out_extra[((size_t)n * D + dd) * HW + (size_t)y * W + x] = act_apply(xo * p.scale[32] + p.shift[32], act2);
            }
            if (O16) {
                const float4* xr = reinterpret_cast<const float4*>(lds_xch) + ((wave ^ 4) * 2) * 64 + lane;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const float4 o = xr[m * 64];
                    const f32x4 own = cw == 0 ? y0[m][0] : y1[m][0];
                    y0[m][0] = own + (f32x4){o.x, o.y, o.z, o.w};          // y0 now = the finished outputs of plane d0 + cw
                }
            }
            if (has_next) {                               // slice 3 of the next tile
                write_slice(3);
                write_x_slices();                         // (read only after the tap loop: published by the next tile's in-loop barrier too)
                shift_planes();
                if (!DEFER) lds_barrier();                // DEFER: slice 3 is published by the next tile's in-loop barrier
            }
            W2STAMP(7);
#ifdef ESTD_W2TIME
            const bool defer_this = DEFER && has_next;
#else
            const bool defer_this = DEFER && has_next && (ESTD_W2_STATS_DEFER || !p.stats_partials);      // uniform
#endif
#ifndef ESTD_W2TIME
            if (p.stats_partials) {                      // uniform; the GRU convolutions (one volume per launch)
                if (O16) plane_stats_o16(y0, d0);
                else tile_stats(y0, y1, d0);
            }
#endif
            if (!defer_this) {
                if (O16) { if (d0 + cw < D) epi_plane(y0, d0 + cw); }
                else if (RB_EARLY) {
                    epi_finish(y0, d0, el0);
                    if (d0 + 1 < D) epi_finish(y1, d0 + 1, el1);
                } else if (RB && ESTD_W2_RB_BATCH) {
                    // read-back instance: the loads of BOTH planes back to back, one exposed round trip per tile instead of two
                    EpiLoads l0, l1;
                    epi_issue(d0, l0);
                    if (d0 + 1 < D) epi_issue(d0 + 1, l1);
                    epi_finish(y0, d0, l0);
                    if (d0 + 1 < D) epi_finish(y1, d0 + 1, l1);
                } else {
                    epi_plane(y0, d0);
                    if (d0 + 1 < D) epi_plane(y1, d0 + 1);
                }
            }
            // (assigned on both paths: a value that survives only on the non-deferred path would stay live through the whole loop)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int x = 0; x < NHW; ++x) { py0[m][x] = y0[m][x]; if (!O16) py1[m][x] = y1[m][x]; }
            pd0 = d0;
            have_prev = defer_this;
            W2STAMP(8);
        }
    }
}

constexpr int PERSISTENT_WGS = 256;     // one 512-thread workgroup per CU (LDS-limited)

}  // namespace

extern "C" int estd_conv3d_k3_wino2(const estd_conv3d_desc* dp, estd_stream_t s)
{
    if (!dp) return ESTD_ERR_ARG;
    const estd_conv3d_desc& d = *dp;
    if (d.N <= 0 || d.D <= 0 || d.H <= 0 || d.W <= 0) return ESTD_ERR_ARG;
    if (!d.in_main || !d.w_wino2 || !d.scale || !d.shift) return ESTD_ERR_ARG;
    if (d.cin_main == 16) {
        // 16 -> 16 + fused 1x1x1 head, only the logit volume leaves the kernel (the stereo heads): csrc/conv3d_wino2_c16.hip
        if (d.n_tiles != 1 || !d.head_w || !d.head_b || !d.out_head) return ESTD_ERR_ARG;
        if (d.out_main || d.in_extra || d.w_extra || d.out_extra || d.residual || d.residual2 || d.accumulate || d.stats_partials ||
            d.out_scale != 1.0f || d.act_a == ESTD_ACT_TANH || d.act_b == ESTD_ACT_TANH) return ESTD_ERR_UNSUPPORTED;
        if (d.in_stride < 16 || (d.in_stride & 3)) return ESTD_ERR_ARG;
        return estd_wino2_c16_launch(d, estd_stream(s));
    }
    if (!d.out_main) return ESTD_ERR_ARG;
    // 32 input channels on the MFMA (+ an optional scalar 33rd input channel), 32 or 16 output channels; no 33rd output channel, no fused head
    if (d.cin_main != 32 || (d.n_tiles != 2 && d.n_tiles != 1 && d.n_tiles != 3) || d.out_head) return ESTD_ERR_UNSUPPORTED;
    const bool o16 = d.n_tiles == 1;                 // 32 -> 16 (the GRU output convolution)
    const bool extra = d.in_extra != nullptr;
    const bool xout = d.n_tiles == 3;                // 33 -> 33 (dres2): w_xout in pack_conv3d_wino2_xout form
    if (xout && (!extra || !d.out_extra || !d.w_xout)) return ESTD_ERR_ARG;
    if (!xout && d.out_extra) return ESTD_ERR_UNSUPPORTED;
    if (xout && (d.residual || d.residual2 || d.accumulate || d.out_scale != 1.0f || d.stats_partials)) return ESTD_ERR_UNSUPPORTED;
    if (o16 && (extra || d.out_stride < 16)) return ESTD_ERR_UNSUPPORTED;
    if (extra != (d.w_extra != nullptr)) return ESTD_ERR_ARG;
    if (extra && d.stats_partials) return ESTD_ERR_UNSUPPORTED;
    if (d.in_stride < 32 || (d.in_stride & 3) || d.out_stride < (o16 ? 16 : 32) || (d.out_stride & 3) || (d.act_split & 1)) return ESTD_ERR_ARG;
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
    static const int nw = [] { const char* e = getenv("ESTD_WINO2_WAVES"); return (e && atoi(e) == 4) ? 4 : 8; }();
    // read-back kind of the launch (the scale multiply lives in the residual instances)
    const bool res_any = d.residual || d.residual2 || d.out_scale != 1.0f;
    const int rbk = (!res_any && !d.accumulate) ? 0 : (!res_any ? 1 : (!d.accumulate ? 2 : 3));
    const bool rb = rbk != 0;
#define ESTD_W2_LAUNCH(NWV, RBV, EXV, OV)                                                                                            \
    do {                                                                                                                             \
        const int lds_bytes = OV ? O16_LDS_BYTES : LDS_BYTES;                                                                        \
        estd_allow_dynamic_lds<conv3d_wino2_kernel<NWV, RBV, EXV, OV>>(lds_bytes);                                                   \
        hipLaunchKernelGGL((conv3d_wino2_kernel<NWV, RBV, EXV, OV>), dim3(grid), dim3(64 * NWV), lds_bytes, estd_stream(s), d,       \
                           tiles_w, tiles_h, dpairs, (int)total);                                                                    \
    } while (0)
    const bool gate = d.gate_r != nullptr;
    if (gate && (!o16 || rb || !d.gate_stats || !d.gate_gamma || !d.gate_beta || d.in_stride != 32)) return ESTD_ERR_UNSUPPORTED;
    if (o16 && gate) {
        estd_allow_dynamic_lds<conv3d_wino2_kernel<8, 0, false, true, false, true>>(O16_LDS_BYTES);
        hipLaunchKernelGGL((conv3d_wino2_kernel<8, 0, false, true, false, true>), dim3(grid), dim3(512), O16_LDS_BYTES, estd_stream(s), d,
                           tiles_w, tiles_h, dpairs, (int)total);
    } else if (o16) {                                // 8-wave form only
        if (rb) ESTD_W2_LAUNCH(8, 3, false, true); else ESTD_W2_LAUNCH(8, 0, false, true);
    } else if (xout) {
        estd_allow_dynamic_lds<conv3d_wino2_kernel<8, 0, true, false, true>>(XOUT_LDS_BYTES);
        hipLaunchKernelGGL((conv3d_wino2_kernel<8, 0, true, false, true>), dim3(grid), dim3(512), XOUT_LDS_BYTES, estd_stream(s), d,
                           tiles_w, tiles_h, dpairs, (int)total);
    } else if (extra) {                              // 8-wave form only (the key || value convolution)
        if (rb) ESTD_W2_LAUNCH(8, 3, true, false); else ESTD_W2_LAUNCH(8, 0, true, false);
    } else if (nw == 8) {
        switch (rbk) {
        case 0: ESTD_W2_LAUNCH(8, 0, false, false); break;
        case 1: ESTD_W2_LAUNCH(8, 1, false, false); break;
        case 2: ESTD_W2_LAUNCH(8, 2, false, false); break;
        default: ESTD_W2_LAUNCH(8, 3, false, false); break;
        }
    } else { if (rb) ESTD_W2_LAUNCH(4, 3, false, false); else ESTD_W2_LAUNCH(4, 0, false, false); }
#undef ESTD_W2_LAUNCH
    return ESTD_LAUNCH_CHECK();
}
