// conv3d_wino.hip -- the plain 32 -> 32 3x3x3 convolution with the DEPTH axis in Winograd F(2,3) form, on gfx950 fp32 MFMA.
//
// Same operator and descriptor as estd_conv3d_k3 (csrc/conv3d_mfma.hip; networks/layers_op.py:16-39 as used at
// hybrid_models/model_hybrid.py:59-60,:95, hybrid_models/hybrid_depth_decoder.py:84-95,:190-191 and the gate convolution of
// transformer/epipolar_transformer.py:21): 24 of the ~33 volume convolutions of a Joint step are this instance, and the direct
// kernel already keeps the matrix pipe 90 % busy -- the way to go faster in fp32 is to issue fewer multiplies.
//
// F(2,3) along d (Lavin & Gray; exact in real arithmetic, every product still an fp32 MFMA with fp32 accumulation):
//   two output planes y_d, y_d+1 from four input planes x_d-1 .. x_d+2 and the three depth taps g0, g1, g2 of a (kh, kw) filter
//   column:
//       t0 = x_d-1 - x_d+1      U0 = g0                   y_d   = m0 + m1 + m2
//       t1 = x_d   + x_d+1      U1 = (g0 + g1 + g2) / 2   y_d+1 = m1 - m2 - m3        with m_i = conv2d_3x3(t_i, U_i)
//       t2 = x_d+1 - x_d        U2 = (g0 - g1 + g2) / 2
//       t3 = x_d   - x_d+2      U3 = g2
//   4 x 9 tap products instead of 2 x 27: 2/3 of the MFMA work.  The transformed filters U are packed on the host in fp64
//   (estdepth_amd/packing.py::pack_conv3d_wino); the input transform is two subtractions and an addition per element, done in
//   registers when a slice pair enters LDS.  Rounding: one extra fp32 rounding on the inputs (|t| <= 2 max|x|) and on U1, U2 --
//   the measured error against an fp64 convolution is 1.3x the direct kernel's (tests/test_gpu_wino.py).
//
// Work decomposition (CDNA4):
//   * a 512-thread workgroup (8 waves, ONE per CU: 92 KB of LDS) owns an output tile of 2 x 8 x 16 voxels (d, h, w) and walks a
//     contiguous range of the column-major tile list upwards in d (persistent, XCD-aware ranges as in the direct kernel);
//   * wave w = (row pair w & 3, channel half w >> 2): two 16-voxel M tiles (tile rows 2rp, 2rp+1) x one 16-channel N tile
//     (channels 16nh .. 16nh+15) x the four products m0..m3 = 32 accumulator registers; the two waves of a SIMD cover each
//     other's LDS latency and epilogue, which is what two co-resident workgroups do for the direct kernel;
//   * LDS holds the four transformed slices (10 x 18 voxels x 32 channels each, same 16-byte XOR swizzle as the direct kernel);
//     the two NEW raw planes of the next tile are fetched into registers during the MFMA loop (one 16-byte buffer load in each
//     of the first six taps), the two planes shared with the current tile stay in registers;
//   * weights stream from L2 in packed [tap][half][quad][lane] order, one tap ahead (2 KB per wave and tap, 295 KB in all);
//   * epilogue: folded BN / bias, ReLU, residual(s), scale, running accumulation, GroupNorm partial sums -- the descriptor
//     fields of the direct kernel's plain instance, applied to both output planes.
// Measured (MI355X, 3 volumes of 64x120x160): 1.16-1.18 ms vs 1.56-1.59 ms for the direct kernel (1.35x; 173-176 TFLOP/s of
// ALGORITHMIC work = 115-117 TFLOP/s executed on the matrix pipe).  Tried and dropped (profiles/README.md, round 2): a six-slot
// ring of RAW slices with the input transform applied between LDS and the MFMA (one barrier per tile, no fill phase:
// 1.31 ms -- twice the LDS reads), the tile's output stores issued from inside the next tile's tap loop (1.17 ms, no gain:
// both waves of a SIMD are in the same phase), the next planes requested after the tap loop (1.20 ms).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "estd_hip.h"
#include "estd_common.h"

#ifndef ESTD_WABL
#define ESTD_WABL 0     // timing ablations only (results are wrong for 1..16): 1 no output stores, 2 no transform writes,
#endif                  // 8 no weight stream, 16 no next-plane prefetch; correct A/B switches: 32 slice writes between the
                        // tiles only (no in-loop write of the next tile's slices), 64 weights one tap ahead instead of two

#ifndef ESTD_XOUT_DEPTH
#define ESTD_XOUT_DEPTH 2   // items the 33rd-output-channel pass reads ahead (registers: 16 per item)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));

constexpr int TH = 8, TW = 16;
constexpr int IN_H = TH + 2, IN_W = TW + 2;
constexpr int SL_VOX = IN_H * IN_W;                 // 180 voxels per input slice (with halo)
constexpr int SLICE_BYTES = SL_VOX * 128;           // 32 channels
constexpr int NTHREADS = 512;
constexpr int SL_CHUNKS = SL_VOX * 8;               // 16-byte chunks per slice: 1440
constexpr int SIT = (SL_CHUNKS + NTHREADS - 1) / NTHREADS;     // chunks per thread per slice: 3
constexpr int XSL_BYTES = 4 * SL_VOX * 4;           // the four transformed slices of the scalar 33rd input channel
constexpr int RED_BYTES = 8 * 2 * 8;                // GroupNorm scratch: 8 waves x {sum, sumsq} doubles
constexpr int WXO_BYTES = (36 * 2 + 3) * 64;        // the 33rd output channel's weights (XOUT)
// The first WLDS_TAPS taps of every tile take their weights from an LDS copy made once per workgroup: vector-memory loads return in
// order, so a weight load issued after the next tile's plane prefetch (HBM, taps 0..5) cannot be consumed before those planes have
// arrived -- 2-4 taps of stall per tile (SQ_WAIT_INST_ANY 57 % of the wave cycles, "no plane prefetch" ablation -8 %).  With LDS
// weights the first vector-memory weight load after the prefetch is the one for tap WLDS_TAPS, issued two taps earlier.
#ifndef ESTD_WLDS_TAPS
#define ESTD_WLDS_TAPS 14
#endif
constexpr int WLDS_TAPS = ESTD_WLDS_TAPS;
constexpr int WLDS_BYTES = WLDS_TAPS * 4096;         // [tap][2 halves][2 quads][64 lanes][4]
constexpr int LDS_BYTES = 4 * SLICE_BYTES + XSL_BYTES + RED_BYTES + WXO_BYTES + WLDS_BYTES;
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
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ int lds_chunk_off(int v, int c) { return v * 128 + ((c ^ ((v >> 1) & 7)) << 4); }

__device__ __forceinline__ float act_apply(float v, int act)
{
    if (act == ESTD_ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == ESTD_ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

// EXTRA: a scalar 33rd INPUT channel (the semantic plane scores of dres2, the 33rd channel of its output for key||value): its
//        3x3 taps per transform form three more k-steps (lane group g multiplies tap 4s+g; taps 9..11 carry zero weights).
// XOUT:  a 33rd OUTPUT channel (dres2): a GEMV, 1/16 efficient on the MFMA -> evaluated on the VALU in a second pass over the
//        transformed slices still in LDS, after the epilogue of the 32 MFMA channels (whose accumulators are dead by then: doing
//        it inside the tap loop from the A fragments, as the direct kernel does, needs 54 registers more than a wave may hold).
//        Wave (rp, nh) takes tile row 2 rp + nh; its lane groups split the channels like the A fragments; two shuffles reduce.
template <bool EXTRA, bool XOUT>
__global__ __launch_bounds__(NTHREADS, 1) void conv3d_wino_kernel(const estd_conv3d_desc p, int tiles_w, int tiles_h, int dpairs, int total_tiles)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lds_x = reinterpret_cast<float*>(smem + 4 * SLICE_BYTES);       // [4][SL_VOX] transformed scalar-channel slices

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rp = wave & 3;            // tile rows 2rp, 2rp+1
    const int nh = wave >> 2;           // output channels 16nh .. 16nh+15
    const int g = lane >> 4;            // k index inside an MFMA
    const int i = lane & 15;            // M row (A) / N column (B, D)
    // MFMA row <-> voxel of a tile row: ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... --
    // rows {0-3,12-15} of lane group g TOGETHER with rows {4-11} of lane group g^1.  With row i = voxel i the XOR swizzle below
    // leaves a 2-way conflict on every tap whose first voxel is odd (PMC: 36 % of the LDS cycles).  Rows {0-3,12-15} take the even
    // voxels and rows {4-11} the odd ones: the two halves of a hardware group then sit in different halves of every 256-byte
    // bank row, and inside a half the eight voxels have distinct swizzle keys -- conflict-free for every tap.
    const int pi = ESTD_WABL & 512 ? i : (i < 4 ? 2 * i : i < 12 ? 2 * i - 7 : 2 * i - 16);
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

    char* lds_wxo = smem + 4 * SLICE_BYTES + XSL_BYTES + RED_BYTES;               // XOUT: the 33rd output channel's weights
    char* lds_w = lds_wxo + WXO_BYTES;                                            // weights of taps 0 .. WLDS_TAPS-1
    for (int e = tid; e < WLDS_BYTES / 16; e += NTHREADS)                         // (visible after the first tile's barriers)
        reinterpret_cast<float4*>(lds_w)[e] = reinterpret_cast<const float4*>(p.w_wino)[e];
    if (XOUT && tid < (36 * 2 + 3) * 4)                                     // (visible after the first tile's barriers)
        reinterpret_cast<float4*>(lds_wxo)[tid] = reinterpret_cast<const float4*>(p.w_xout)[tid];

    const int ch = 16 * nh + i;                     // this lane's output channel
    const float sc = p.scale[ch], sh = p.shift[ch];
    const int act0 = ch < p.act_split ? p.act_a : p.act_b;
    // plain launches (no read-back stream, no tanh): the activation is a per-lane floor and the epilogue has no branch per element
    const bool plain_epi = !p.residual && !p.residual2 && !p.accumulate && p.out_scale == 1.0f &&
                           p.act_a != ESTD_ACT_TANH && p.act_b != ESTD_ACT_TANH && !(ESTD_WABL & 64);
    const float act_floor = act0 == ESTD_ACT_RELU ? 0.f : ESTD_NO_FLOOR;
    // packed weights: [37 taps (36 + 1 the prefetch may read)][2 halves][2 quads][64 lanes][4]
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w_wino, (size_t)37 * 2 * 2 * 256);
    const int wlane = lane * 16 + nh * 2048;
    const int row0 = 2 * rp;
    // scalar-channel weights: [2 halves][3 quads][64 lanes][4], element 3 s + k of lane (g, j) = U_s[16 nh + j][extra][tap 4 k + g]
    const __amdgpu_buffer_rsrc_t rs_wx = make_rsrc(EXTRA ? p.w_extra : p.w_wino, (size_t)2 * 3 * 256);
    // this lane's taps of the scalar channel: k-step k covers tap 4k + g (clamped: taps 9..11 have zero weights)
    int xtap_off[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int tp = min(4 * k + g, 8);
        xtap_off[k] = (tp / 3) * IN_W + (tp % 3) + pi;
    }

    while (u < u_end) {
        // ---- column segment [u, seg_end): same (n, h-tile, w-tile), consecutive depth pairs ----
        const int col = u / dpairs;
        int dp = u - col * dpairs;
        const int twi = col % tiles_w, c2 = col / tiles_w;
        const int thi = c2 % tiles_h, n = c2 / tiles_h;
        const int tw0 = twi * TW, th0 = thi * TH;
        const int seg_end = min(u_end, (col + 1) * dpairs);

        const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in_main + (size_t)n * vol * p.in_stride, vol * p.in_stride);
        __amdgpu_buffer_rsrc_t rs_ex = rs_in;
        if (EXTRA) rs_ex = make_rsrc(p.in_extra + (size_t)n * vol, vol);
        unsigned voffx = OOB_OFFSET;                 // this thread's voxel of a scalar slice (threads 0..179)
        if (EXTRA) {
            const int zy = tid / IN_W, zx = tid % IN_W;
            const int gy = th0 - 1 + zy, gx = tw0 - 1 + zx;
            if (tid < SL_VOX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) voffx = (unsigned)(gy * W + gx) * 4u;
        }
        auto load_x = [&](int pd) {
            return (unsigned)pd < (unsigned)D ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ex, voffx, pd * HW * 4, 0)) : 0.0f;
        };
        __amdgpu_buffer_rsrc_t rs_out = rs_in, rs_res = rs_in, rs_res2 = rs_in;
        rs_out = make_rsrc(p.out_main + (size_t)n * vol * p.out_stride, vol * p.out_stride);
        if (p.residual) rs_res = make_rsrc(p.residual + (size_t)n * vol * p.out_stride, vol * p.out_stride);
        if (p.residual2) rs_res2 = make_rsrc(p.residual2 + (size_t)n * vol * p.out_stride, vol * p.out_stride);
        const int in_slice_bytes = HW * p.in_stride * 4;
        const int out_plane_bytes = HW * p.out_stride * 4;

        // per-thread slice elements (validity in y / x does not depend on d)
        unsigned voff[SIT];
        int loff[SIT];
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int e = tid + it * NTHREADS;
            const int vs = e >> 3, c = e & 7;
            const int zy = vs / IN_W, zx = vs % IN_W;
            const int gy = th0 - 1 + zy, gx = tw0 - 1 + zx;
            const bool ok = e < SL_CHUNKS && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            voff[it] = ok ? (unsigned)((gy * W + gx) * p.in_stride + c * 4) * 4u : OOB_OFFSET;
            loff[it] = e < SL_CHUNKS ? lds_chunk_off(vs, c) : -1;
        }
        auto load_plane = [&](int pd, float4 (&dst)[SIT]) {
            const bool pv = (unsigned)pd < (unsigned)D;        // wave-uniform; planes outside the volume are zero padding
#pragma unroll
            for (int it = 0; it < SIT; ++it)
                dst[it] = pv ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[it], pd * in_slice_bytes, 0))
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        };

        // epilogue lane offsets (bytes inside one depth plane): rows row0, row0+1; columns 4g .. 4g+3 of the tile
        // D rows 4g + r of lane group g are the voxels pi(4g + r): 2r, 2r + 1, 2r + 9, 2r + 8 for g = 0..3
        const int ey0 = th0 + row0;
        const int ex0 = tw0 + ((ESTD_WABL & 512) ? 4 * g : (g == 0 ? 0 : g == 1 ? 1 : g == 2 ? 9 : 8));
        const int exs = (ESTD_WABL & 512) ? 1 : 2;
        auto eoff_of = [&](int m, int r) {
            const int y = ey0 + m, x = ex0 + exs * r;
            return (y < H && x < W) ? (unsigned)((y * W + x) * p.out_stride + ch) * 4u : OOB_OFFSET;
        };
        unsigned eoff[2][4];                        // held in registers by the plain instance, recomputed by the 33-channel ones
        if (!EXTRA) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) eoff[m][r] = eoff_of(m, r);
        }

        // epilogue of one plane: the 2 x 4 voxels (tile row m, column 4g + r) of channel ch this lane holds.  Every read-back stream
        // (residuals, the running sum) issues its eight loads back to back and is waited for ONCE: one load -> wait -> store per
        // element cost 6.4 us per stream per tile (1.1 ms per step on pre1 / pre2).
        auto epi_plane = [&](const f32x4 (&a)[2], int dd) {
            const int so = dd * out_plane_bytes;
            unsigned eo[2][4];
            float r1[2][4], r2[2][4], ro[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) eo[m][r] = EXTRA ? eoff_of(m, r) : eoff[m][r];
            if (plain_epi) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = __builtin_fmaxf(__builtin_fmaf(a[m][r], sc, sh), act_floor);
                        if (!(ESTD_WABL & 1)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, eo[m][r], so, 0);
                    }
                return;
            }
            if (p.residual) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) r1[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res, eo[m][r], so, 0));
            }
            if (p.residual2) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) r2[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res2, eo[m][r], so, 0));
            }
            if (p.accumulate) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ro[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_out, eo[m][r], so, 0));
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = act_apply(a[m][r] * sc + sh, act0);
                    if (p.residual) v += r1[m][r];
                    if (p.residual2) v += r2[m][r];
                    v *= p.out_scale;
                    if (p.accumulate) v += ro[m][r];
                    if (!(ESTD_WABL & 1)) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs_out, eo[m][r], so, 0);
                }
        };
        // GroupNorm(1 group) partial sums of the raw outputs of one plane: group = channel half nh.  Fixed-order reduction
        // (lanes by butterfly, the four row-pair waves of a half through LDS) -> deterministic.  Workgroup-uniform call.
        auto plane_stats = [&](const f32x4 (&a)[2], int dd) {
            double s_sum = 0.0, s_sq = 0.0;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((EXTRA ? eoff_of(m, r) : eoff[m][r]) != OOB_OFFSET) { const double v = (double)(a[m][r] * sc + sh); s_sum += v; s_sq += v * v; }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { s_sum += __shfl_xor(s_sum, o); s_sq += __shfl_xor(s_sq, o); }
            double* red = reinterpret_cast<double*>(smem + 4 * SLICE_BYTES + XSL_BYTES);
            __syncthreads();                                     // the previous plane's scratch has been consumed
            if (lane == 0) { red[wave * 2] = s_sum; red[wave * 2 + 1] = s_sq; }
            __syncthreads();
            if (tid < 4) {
                const int grp = tid >> 1, q = tid & 1;
                const double tot = red[(grp * 4 + 0) * 2 + q] + red[(grp * 4 + 1) * 2 + q] + red[(grp * 4 + 2) * 2 + q] + red[(grp * 4 + 3) * 2 + q];
                // partial index = canonical tile id (n, d, thi, twi), as the direct kernel writes it
                const size_t tile_id = (((size_t)n * D + dd) * tiles_h + thi) * tiles_w + twi;
                p.stats_partials[tile_id * 4 + tid] = tot;
            }
        };

        // raw planes in registers: xa = x[d0-1], xb = x[d0], xc = x[d0+1], xd = x[d0+2]  (+ the scalar channel's: ea..ed)
        float4 xa[SIT], xb[SIT], xc[SIT], xd[SIT];
        float ea = 0.f, eb = 0.f, ec = 0.f, ed = 0.f;
        {
            const int d0 = 2 * dp;
            load_plane(d0 - 1, xa);
            load_plane(d0, xb);
            load_plane(d0 + 1, xc);
            load_plane(d0 + 2, xd);
            if (EXTRA) { ea = load_x(d0 - 1); eb = load_x(d0); ec = load_x(d0 + 1); ed = load_x(d0 + 2); }
        }

        // input transform B^T x along depth of the planes in (xa, xb, xc, xd), straight into LDS slice sl
        auto write_slice = [&](int sl) {
#pragma unroll
            for (int it = 0; it < SIT; ++it) {
                if ((it < SIT - 1 || loff[it] >= 0) && !(ESTD_WABL & 2)) {
                    const float4 v = sl == 0 ? f4_sub(xa[it], xc[it]) : sl == 1 ? f4_add(xb[it], xc[it])
                                   : sl == 2 ? f4_sub(xc[it], xb[it]) : f4_sub(xb[it], xd[it]);
                    *reinterpret_cast<float4*>(smem + sl * SLICE_BYTES + loff[it]) = v;
                }
            }
        };
        auto write_x_slices = [&]() {
            if (tid < SL_VOX) {
                lds_x[0 * SL_VOX + tid] = ea - ec;
                lds_x[1 * SL_VOX + tid] = eb + ec;
                lds_x[2 * SL_VOX + tid] = ec - eb;
                lds_x[3 * SL_VOX + tid] = eb - ed;
            }
        };
        auto shift_planes = [&]() {                      // planes d0+1, d0+2 are planes d0'-1, d0' of the next tile
#pragma unroll
            for (int it = 0; it < SIT; ++it) { xa[it] = xc[it]; xb[it] = xd[it]; }
            if (EXTRA) { ea = ec; eb = ed; }
        };
        // PIPE: slice s is only read by taps 9s .. 9s+8, so the NEXT tile's slices 0..2 are written inside this tile's tap loop
        // (one barrier each, at taps 9 / 18 / 27) and only slice 3 between two tiles; the epilogue then runs with no barrier
        // behind it, so the waves drift apart and overlap it with the next tile's first taps.  (XOUT re-reads all four slices
        // after its epilogue: it keeps the plain order.)
        constexpr bool PIPE = !XOUT && !(ESTD_WABL & 32);
        bool first = true;

        for (; u < seg_end; ++u, ++dp) {
            const int d0 = 2 * dp;
            if (!PIPE || first) {
                lds_barrier();                          // every wave is done reading the previous tile's slices
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) write_slice(sl);
                if (EXTRA) write_x_slices();
                shift_planes();
                lds_barrier();
                first = false;
            }

            const bool has_next = (u + 1 < seg_end);     // wave-uniform
            const int nd = d0 + 3;                       // new planes of the next tile: nd, nd + 1
            const bool v0 = nd < D, v1 = nd + 1 < D;

            f32x4 acc[4][2];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int m = 0; m < 2; ++m) acc[s][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

            constexpr bool W2 = !EXTRA && !(ESTD_WABL & 64);     // weights two taps ahead (8 more registers: the plain instance only)
            float4 bcur[2], bnext[2], bnext2[2];
            auto load_w = [&](int t, int q) {               // t is a compile-time constant after unrolling
                if (t < WLDS_TAPS) return *reinterpret_cast<const float4*>(lds_w + t * 4096 + q * 1024 + wlane);
                return as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, t * 4096 + q * 1024, 0));
            };
#pragma unroll
            for (int q = 0; q < 2; ++q) bcur[q] = load_w(0, q);
            if (W2) {
#pragma unroll
                for (int q = 0; q < 2; ++q) bnext[q] = load_w(1, q);
            }
#ifdef ESTD_TIMELINE
            if (tid == 0 && p.stats_partials) {     // debug build only: per-tile start stamps instead of GroupNorm sums
                const size_t tile_id = (((size_t)n * D + d0) * tiles_h + thi) * tiles_w + twi;
                p.stats_partials[tile_id * 4 + 0] = (double)__builtin_amdgcn_s_memtime();
                p.stats_partials[tile_id * 4 + 1] = (double)blockIdx.x;
                p.stats_partials[tile_id * 4 + 2] = (double)wall_clock64();
            }
#endif
            // A fragments of (transform s, tap kh kw): two 16-byte LDS reads per M tile (channels 4g.., 16+4g..)
            auto load_a = [&](int tap, float4 (&a0)[2], float4 (&a1)[2]) {
                const int s = tap / 9, kh = (tap % 9) / 3, kw = tap % 3;
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const int vs = (row0 + m + kh) * IN_W + kw + pi;
                    const int off0 = s * SLICE_BYTES + lds_chunk_off(vs, g);
                    a0[m] = *reinterpret_cast<const float4*>(smem + off0);
                    a1[m] = *reinterpret_cast<const float4*>(smem + (off0 ^ 64));
                }
            };
            constexpr bool APF = !XOUT;                  // A fragments one tap ahead (the 33 -> 33 instance needs the registers)
            float4 a0c[2], a1c[2], a0n[2], a1n[2];
            if (APF) load_a(0, a0c, a1c);

#pragma clang loop unroll(full)
            for (int tap = 0; tap < 36; ++tap) {
                const int s = tap / 9;
                if (PIPE && has_next && tap == 27) {
                    lds_barrier();                       // slices 0..2 have been read for the last time by every wave
                    write_slice(0);
                    write_slice(1);
                    write_slice(2);
                }
                // next tap's weights (the packed buffer carries one padding tap)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (ESTD_WABL & 8) { bnext[q] = bcur[q]; asm volatile("" : "+v"(bnext[q].x)); }
                    else if (W2) { if (tap + 2 < 36) bnext2[q] = load_w(tap + 2, q); }
                    else bnext[q] = load_w(tap + 1, q);
                }
                // one 16-byte chunk of the NEXT tile's two new planes per tap
                if (has_next && tap < 2 * SIT && !(ESTD_WABL & 16)) {
                    const int it = tap % SIT;
                    if (tap < SIT) xc[it] = v0 ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[it], nd * in_slice_bytes, 0))
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
                    else           xd[it] = v1 ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[it], (nd + 1) * in_slice_bytes, 0))
                                               : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                if (EXTRA && has_next && (tap == 2 * SIT || tap == 2 * SIT + 1)) {          // the scalar channel's two new planes
                    if (tap == 2 * SIT) ec = v0 ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ex, voffx, nd * HW * 4, 0)) : 0.f;
                    else                ed = v1 ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ex, voffx, (nd + 1) * HW * 4, 0)) : 0.f;
                }
                if (APF) { if (tap + 1 < 36) load_a(tap + 1, a0n, a1n); }        // next tap's A fragments: LDS latency under this tap's MFMAs
                else load_a(tap, a0c, a1c);
                const float av0[8] = {a0c[0].x, a0c[0].y, a0c[0].z, a0c[0].w, a1c[0].x, a1c[0].y, a1c[0].z, a1c[0].w};
                const float av1[8] = {a0c[1].x, a0c[1].y, a0c[1].z, a0c[1].w, a1c[1].x, a1c[1].y, a1c[1].z, a1c[1].w};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {     // the two M tiles alternate: no MFMA waits for its own predecessor
                    const float4 bq = bcur[ks >> 2];
                    const float b = (ks & 3) == 0 ? bq.x : (ks & 3) == 1 ? bq.y : (ks & 3) == 2 ? bq.z : bq.w;
                    acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av0[ks], b, acc[s][0], 0, 0, 0);
                    acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av1[ks], b, acc[s][1], 0, 0, 0);
                }
                bcur[0] = bnext[0];
                bcur[1] = bnext[1];
                if (W2) { bnext[0] = bnext2[0]; bnext[1] = bnext2[1]; }
                if (APF) { a0c[0] = a0n[0]; a0c[1] = a0n[1]; a1c[0] = a1n[0]; a1c[1] = a1n[1]; }
                __builtin_amdgcn_sched_barrier(0);       // keep each tap's loads inside the tap (bounds live registers)
            }

            // ---- the scalar input channel: per transform three more k-steps (lane group g = tap 4k + g of the 3x3 window) ----
            if (EXTRA) {
                // (the 12 weights of this stage are fetched here, L2-resident, rather than held across the tap loop)
                float4 bx[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) bx[q] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_wx, lane * 16, (nh * 3 + q) * 1024, 0));
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const int idx = s * 3 + k;
                        const float4 bq = bx[idx >> 2];
                        const float b = (idx & 3) == 0 ? bq.x : (idx & 3) == 1 ? bq.y : (idx & 3) == 2 ? bq.z : bq.w;
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            const float a = lds_x[s * SL_VOX + (row0 + m) * IN_W + xtap_off[k]];
                            acc[s][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[s][m], 0, 0, 0);
                        }
                    }
            }

            if (PIPE && has_next) {                       // slice 3 (+ the scalar channel's slices) of the next tile
                lds_barrier();
                write_slice(3);
                if (EXTRA) write_x_slices();
                shift_planes();
                lds_barrier();
            }

            // ---- output transform A^T m and the epilogue of the two planes ----
            f32x4 y0[2], y1[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                y0[m] = acc[0][m] + acc[1][m] + acc[2][m];
                y1[m] = acc[1][m] - acc[2][m] - acc[3][m];
            }
            if (p.stats_partials) {                      // uniform; the GRU gate convolution (one volume per launch)
                plane_stats(y0, d0);
                if (d0 + 1 < D) plane_stats(y1, d0 + 1);              // (odd D: the last pair has one plane)
            }
            epi_plane(y0, d0);
            if (d0 + 1 < D) epi_plane(y1, d0 + 1);

            // ---- 33rd output channel: out[32] = sum over (s, tap, channel) of the transformed inputs x U_s[32] ----
            if (XOUT) {
                // weights (LDS copy made at kernel start): [36 taps][2 quads][4 lane groups][4] + scalar input channel [3 quads][4][4]
                float xacc[4] = {0.f, 0.f, 0.f, 0.f};
                int ix = pi, gx_ = g;                          // opaque copies: this pass's LDS addresses are formed HERE, not hoisted
                asm volatile("" : "+v"(ix), "+v"(gx_));       // over the tap loop above, whose register budget is spent
                // 36 (transform s, tap) items of 4 LDS reads each, software-pipelined XD items ahead by hand: fully unrolled with a
                // scheduling barrier per item (left to itself the scheduler hoists all 144 reads and spills 360 registers; rolled, every
                // filter row waited for its own reads -- 10 000 cycles per tile with both waves of a SIMD in this pass at the same time)
                constexpr int XD = ESTD_XOUT_DEPTH;
                float4 xb_[XD + 1][4];
                auto x_issue = [&](int it, float4 (&buf)[4]) {
                    const int s_ = it / 9, kh = (it / 3) % 3, kw = it % 3;
                    const char* wrow = lds_wxo + ((s_ * 9 + kh * 3) * 2) * 64 + gx_ * 16;
                    const int vs = (row0 + nh + kh) * IN_W + kw + ix;
                    const int off0 = s_ * SLICE_BYTES + lds_chunk_off(vs, gx_);
                    buf[0] = *reinterpret_cast<const float4*>(smem + off0);
                    buf[1] = *reinterpret_cast<const float4*>(smem + (off0 ^ 64));
                    buf[2] = *reinterpret_cast<const float4*>(wrow + (kw * 2 + 0) * 64);
                    buf[3] = *reinterpret_cast<const float4*>(wrow + (kw * 2 + 1) * 64);
                };
                f32x2 t2[4] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};     // two partial sums each: the products pair up into v_pk_fma_f32
#pragma unroll
                for (int it = 0; it < XD; ++it) x_issue(it, xb_[it]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < 36; ++it) {
                    if (it + XD < 36) x_issue(it + XD, xb_[(it + XD) % (XD + 1)]);
                    const float4 a0 = xb_[it % (XD + 1)][0], a1 = xb_[it % (XD + 1)][1], w0 = xb_[it % (XD + 1)][2], w1 = xb_[it % (XD + 1)][3];
                    f32x2& t = t2[it / 9];
                    t = __builtin_elementwise_fma((f32x2){a0.x, a0.y}, (f32x2){w0.x, w0.y}, t);
                    t = __builtin_elementwise_fma((f32x2){a0.z, a0.w}, (f32x2){w0.z, w0.w}, t);
                    t = __builtin_elementwise_fma((f32x2){a1.x, a1.y}, (f32x2){w1.x, w1.y}, t);
                    t = __builtin_elementwise_fma((f32x2){a1.z, a1.w}, (f32x2){w1.z, w1.w}, t);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_) xacc[s_] = t2[s_].x + t2[s_].y;
                {
                    float4 wx[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) wx[q] = *reinterpret_cast<const float4*>(lds_wxo + (72 + q) * 64 + gx_ * 16);
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            const int idx = s * 3 + k;
                            const float4 wq = wx[idx >> 2];
                            const float wv = (idx & 3) == 0 ? wq.x : (idx & 3) == 1 ? wq.y : (idx & 3) == 2 ? wq.z : wq.w;
                            xacc[s] = fmaf(lds_x[s * SL_VOX + (row0 + nh) * IN_W + xtap_off[k]], wv, xacc[s]);
                        }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    xacc[s] += __shfl_xor(xacc[s], 16);
                    xacc[s] += __shfl_xor(xacc[s], 32);
                }
                const float sc2 = p.scale[32], sh2 = p.shift[32];
                const int act2 = 32 < p.act_split ? p.act_a : p.act_b;
                const int y = th0 + row0 + nh, x = tw0 + pi;
                const int dd = d0 + g;                                   // lane group 0 stores plane d0, group 1 plane d0 + 1
                const float raw = g == 0 ? xacc[0] + xacc[1] + xacc[2] : xacc[1] - xacc[2] - xacc[3];
                if (g < 2 && dd < D && y < H && x < W)
                    p.out_extra[((size_t)n * D + dd) * HW + (size_t)y * W + x] = act_apply(raw * sc2 + sh2, act2);
            }
        }
    }
}

constexpr int PERSISTENT_WGS = 256;     // one 512-thread workgroup per CU (LDS-limited)

}  // namespace

extern "C" int estd_conv3d_k3_wino(const estd_conv3d_desc* dp, estd_stream_t s)
{
    if (!dp) return ESTD_ERR_ARG;
    const estd_conv3d_desc& d = *dp;
    if (d.N <= 0 || d.D <= 0 || d.H <= 0 || d.W <= 0) return ESTD_ERR_ARG;
    if (!d.in_main || !d.w_wino || !d.scale || !d.shift || !d.out_main) return ESTD_ERR_ARG;
    // 32 main input channels -> 32 output channels on the MFMA, optionally a scalar 33rd input channel (w_extra in Winograd
    // packing) and, with it, a 33rd output channel (n_tiles == 3, w_xout in Winograd packing); no fused head
    if (d.cin_main != 32 || (d.n_tiles != 2 && d.n_tiles != 3) || d.out_head) return ESTD_ERR_UNSUPPORTED;
    const bool extra = d.in_extra != nullptr, xout = d.n_tiles == 3;
    if (extra != (d.w_extra != nullptr)) return ESTD_ERR_ARG;
    if (xout && (!extra || !d.out_extra || !d.w_xout)) return ESTD_ERR_ARG;
    if (!xout && d.out_extra) return ESTD_ERR_UNSUPPORTED;
    if (d.stats_partials && extra) return ESTD_ERR_UNSUPPORTED;
    if (d.in_stride < 32 || (d.in_stride & 3) || d.out_stride < 32 || (d.act_split & 1)) return ESTD_ERR_ARG;
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
#define ESTD_WINO_LAUNCH(E, X)                                                                                                   \
    do {                                                                                                                         \
        estd_allow_dynamic_lds<conv3d_wino_kernel<E, X>>(LDS_BYTES);                                                             \
        hipLaunchKernelGGL((conv3d_wino_kernel<E, X>), dim3(grid), dim3(NTHREADS), LDS_BYTES, estd_stream(s), d, tiles_w, tiles_h, \
                           dpairs, (int)total);                                                                                  \
    } while (0)
    if (xout) ESTD_WINO_LAUNCH(true, true);
    else if (extra) ESTD_WINO_LAUNCH(true, false);
    else ESTD_WINO_LAUNCH(false, false);
#undef ESTD_WINO_LAUNCH
    return ESTD_LAUNCH_CHECK();
}
