// conv3d_split_bf16.hip -- the 32->32 3x3x3 convolution with fp32 operands split into three bf16 pieces.
//
// Same operator, descriptor and epilogue as conv3d_mfma.hip (networks/layers_op.py:16-39 stacks,
// transformer/epipolar_transformer.py:21); different arithmetic for the products:
//
//     a = a1 + a2 + a3,  b = b1 + b2 + b3      (round-to-nearest bf16 pieces; the split is exact: 3 x 8 >= 24 bits)
//     a*b ~= a1b3 + a3b1 + a2b2 + a1b2 + a2b1 + a1b1      (dropped terms <= 2^-26 |ab|; fp32 accumulation in the MFMA)
//
// i.e. six v_mfma_f32_16x16x32_bf16 (K = 32 = all input channels of one tap, 16 cycles each) replace eight
// v_mfma_f32_16x16x4_f32 (32 cycles each): 96 instead of 256 matrix-pipe cycles per (16 voxels x 16 channels x tap),
// and far fewer joules -- the fp32 kernel is DVFS-limited.  The result carries fp32-level error
// (tests/test_gpu_split_conv.py measures both kernels against an fp64 convolution).
//
// Design (CDNA4), differences from the fp32 kernel:
//   * The operands are needed faster than a wave-private stream from L2 can supply them, so BOTH operands come from
//     LDS: 512-thread workgroup (8 waves, one per tile row), tile 1 x 8 x 32 voxels, one workgroup per CU.
//   * Input slices are split into bf16 pieces ONCE, when they enter LDS (9 VALU ops per channel pair per slice
//     element; every element is then read 27 times).  LDS slice layout [piece][chunk of 8 channels][voxel] x 16 B with
//     the voxel index rotated by 2*chunk: the A-fragment read (16 consecutive voxels, chunk = k-group) and the fill
//     (chunk fastest across lanes) are both bank-conflict free.
//   * 2-slot slice ring: taps kd=0 read slice d-1; once they are done its slot is refilled with slice d+1 (prefetched
//     into registers during the previous tile) while the kd=1 taps run; kd=2 then reads it.  2 x 66 KB.
//   * Weights: host-split [tap][piece][n-tile][lane][8] bf16 (6 KB per tap, packing.py::pack_conv3d_split), streamed
//     L2 -> registers -> a 2-slot LDS buffer one tap ahead; fragments for tap t+1 are read into registers during tap t.
//     One LDS-only barrier per tap publishes the weight slot, the ring refill and the GroupNorm scratch.
//   * 33rd INPUT channel (dres2, key|value convs): its 27 taps are one more K = 32 block ("tap 27"): the A fragment is
//     gathered from a 3-slot LDS ring of the scalar slices and split in registers; its weights are record 27.
//   * 33rd OUTPUT channel (dres2): a third N tile whose only live column is 0; its B fragment (64 B per piece) rides in
//     the padding of the 8 KB weight record, lanes of the other columns read a zero block.
//   * The body of a tap is ONE basic block (no exec-mask or uniform branches: out-of-range work is redirected to
//     out-of-bounds buffer offsets / an LDS dump area) so that the issue order can be prescribed: every MFMA is
//     followed by a few of the tap's other instructions (LDS reads of the next tap, weight hand-over, slice split,
//     the previous tile's epilogue), which then execute in the shadow of the 16-cycle matrix operation.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "estd_hip.h"
#include "estd_common.h"

#ifdef ESTD_TIMELINE
#define ESTD_SPLIT_STATS_ON 0   // debug build: stats_partials receives per-tile time stamps instead of GroupNorm sums
#else
#define ESTD_SPLIT_STATS_ON 1
#endif
#ifndef ESTD_SABL
#define ESTD_SABL 0   // timing ablations only (tools/ablate_split.sh); results are wrong when != 0
#endif
#ifndef ESTD_SPIPE
#define ESTD_SPIPE 1  // 1: prescribe the per-tap issue pipeline with sched_group_barrier
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((__vector_size__(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
typedef unsigned int u32x2 __attribute__((__vector_size__(8)));

constexpr int TH = 8, TW = 32;
constexpr int IN_H = TH + 2, IN_W = TW + 2;
constexpr int SL_VOX = IN_H * IN_W;            // 340 voxels per slice (with halo)
constexpr int PLANE = 352;                     // 16-byte entries per chunk plane: multiple of 16, >= 340 + 6 (rotation)
constexpr int CHUNK_BYTES = PLANE * 16;        // 5632
constexpr int PIECE_BYTES = 4 * CHUNK_BYTES;   // 22528
constexpr int SLICE_BYTES = 3 * PIECE_BYTES;   // 67584
constexpr int WMAIN_BYTES = 3 * 2 * 64 * 16;   // 6144 bytes of split weights per tap: [piece][n-tile][lane] x 16 B
constexpr int WTAP_BYTES = 512 * 16;           // weight record of one tap, global and LDS: every thread hands over 16 B;
                                               // bytes 6144..6335 = column of the 33rd output channel [piece][k-group] x 16 B, rest zero
constexpr int WX_OFF = WMAIN_BYTES;            // 33rd-output column inside a record
constexpr int WZERO_OFF = WMAIN_BYTES + 256;   // 16 zero bytes inside every record
constexpr int LDS_W = 2 * SLICE_BYTES;
constexpr int LDS_DUMP = LDS_W + 2 * WTAP_BYTES;       // 176 x 16 B: target of the writes of threads without an item
constexpr int XSLICE_BYTES = SL_VOX * 4;               // scalar (33rd input channel) slice, fp32
constexpr int LDS_XRING = LDS_DUMP + 176 * 16;         // 3 slots
constexpr int LDS_RED = LDS_XRING + 4096;
constexpr int LDS_TOTAL = LDS_RED + 8 * 4 * 8;         // 158720 of 163840
constexpr int FILL_E = SL_VOX * 4;             // (voxel, chunk) items per slice: 1360
constexpr int FIT = 3;                         // items per thread (512 threads)
constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;

__device__ __forceinline__ float4 as_float4(u32x4 v)
{
    float4 f;
    __builtin_memcpy(&f, &v, 16);
    return f;
}

__device__ __forceinline__ float2 as_float2(u32x2 v)
{
    float2 f;
    __builtin_memcpy(&f, &v, 8);
    return f;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// x0, x1 -> three packed bf16 pairs with x = h + m + l (exact unless x is within 2^-24 of the bf16 range limits)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l)
{
    const bf16x2 hb = {(__bf16)x0, (__bf16)x1};
    h = __builtin_bit_cast(unsigned, hb);
    float r0 = x0 - __builtin_bit_cast(float, h << 16);
    float r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
    const bf16x2 mb = {(__bf16)r0, (__bf16)r1};
    m = __builtin_bit_cast(unsigned, mb);
    r0 -= __builtin_bit_cast(float, m << 16);
    r1 -= __builtin_bit_cast(float, m & 0xffff0000u);
    const bf16x2 lb = {(__bf16)r0, (__bf16)r1};
    l = __builtin_bit_cast(unsigned, lb);
}

// 8 channels (two float4) of one voxel -> LDS, one 16-byte write per piece (o1 - o0 = o2 - o1 = PIECE_BYTES for real items)
__device__ __forceinline__ void fill_item(char* smem, int o0, int o1, int o2, float4 a, float4 b)
{
    u32x4 h, m, l;
    unsigned th, tm, tl;
    split2(a.x, a.y, th, tm, tl); h[0] = th; m[0] = tm; l[0] = tl;
    split2(a.z, a.w, th, tm, tl); h[1] = th; m[1] = tm; l[1] = tl;
    split2(b.x, b.y, th, tm, tl); h[2] = th; m[2] = tm; l[2] = tl;
    split2(b.z, b.w, th, tm, tl); h[3] = th; m[3] = tm; l[3] = tl;
    *reinterpret_cast<u32x4*>(smem + o0) = h;
    *reinterpret_cast<u32x4*>(smem + o1) = m;
    *reinterpret_cast<u32x4*>(smem + o2) = l;
}

// compile-time tap index: a "#pragma unroll" loop gets peeled (some taps are special) and is then no longer fully unrolled
template <typename F, int... I>
__device__ __forceinline__ void for_each_tap(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}

// Issue order of one tap: one MFMA, then a few of the other instructions of the tap (IGroupLP pipeline).
//   0x008 MFMA   0x100 DS read   0x200 DS write   0x020 VMEM read   0x040 VMEM write   0x002 VALU   0x004 SALU
template <int TAP, int NT, bool EXTRA>
__device__ __forceinline__ void tap_pipeline()
{
#if ESTD_SPIPE
    constexpr int NM = NT == 3 ? 30 : 12 * NT;      // MFMAs of the tap (third N tile: 3 per M tile)
    constexpr int NR = NT == 3 ? 13 : 6 + 3 * NT;   // LDS fragment reads of the next tap
    constexpr bool VALU_HEAVY = TAP <= 2 || (TAP >= 8 && TAP <= 10) || (EXTRA && TAP == 26);
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (k < NR + (EXTRA && TAP == 26 ? 4 : 0)) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (VALU_HEAVY) __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);    // epilogue / slice split / extra-channel split
        if (k >= NR && k < NR + 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        if (k >= NR) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        if (TAP >= 1 && TAP <= 2 && k >= NR + 4 && k < NR + 8) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);
    }
#endif
}

// NT: 16-channel output tiles on the MFMA (1 = 16 channels; 2 = 32 channels; 3 = 32 channels + the 33rd in a third tile)
// RES: some of residual / residual2 / accumulate is present (otherwise the epilogue issues no loads at all)
template <int NT, bool EXTRA, bool TANH, bool STATS, bool RES>
__global__ __launch_bounds__(512, 1) void conv3d_k3_split_kernel(const estd_conv3d_desc p, int tiles_w, int tiles_h, int total_tiles)
{
    constexpr int NTAPS = EXTRA ? 28 : 27;          // tap 27 = the scalar input channel's 27 taps as one K = 32 block
    constexpr int NB = NT == 1 ? 1 : 2;             // full N tiles (the third tile of NT == 3 is handled apart)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = tile row of this wave
    const int g = lane >> 4;            // k group (A, B) / row group (D)
    const int i = lane & 15;            // M row (A) / N column (B, D)
    const int D = p.D, H = p.H, W = p.W;

    int u, u_end;
    {
        const int G = gridDim.x, bid = blockIdx.x;
        const int r = ((G & 7) == 0) ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;     // XCD x owns a contiguous block of ranges
        u = (int)((long long)total_tiles * r / G);
        u_end = (int)((long long)total_tiles * (r + 1) / G);
    }
    if (u >= u_end) return;

    const int cbase = NB * i;           // output channels of this lane: 2i (n-tile 0), 2i+1 (n-tile 1); NT == 1: channel i
    const float sc0 = p.scale[cbase], sh0 = p.shift[cbase], sc1 = p.scale[cbase + NB - 1], sh1 = p.shift[cbase + NB - 1];
    const int act0 = cbase < p.act_split ? p.act_a : p.act_b;
    const float relu_floor = act0 == ESTD_ACT_RELU ? 0.0f : -__builtin_huge_valf();    // max(v, floor): branch-free ReLU / identity
    const float out_scale = p.out_scale;
    float sc2 = 0.f, sh2 = 0.f, relu_floor2 = 0.f;
    if (NT == 3) { sc2 = p.scale[32]; sh2 = p.shift[32]; relu_floor2 = p.act_b == ESTD_ACT_RELU ? 0.0f : -__builtin_huge_valf(); }

    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w_split, (size_t)NTAPS * WTAP_BYTES);
    const size_t vol = (size_t)D * H * W;
    const int HW = H * W;
    const int tiles_w16 = (W + 15) / 16;                 // GroupNorm partials keep the 8x16-tile numbering of estd_conv3d_k3_grid
    const int a_lane = g * CHUNK_BYTES + (i + 2 * g) * 16;     // lane part of an A-fragment address
    const int b_lane = lane * 16;
    // third N tile: column j < 3 holds PIECE j of the 33rd output channel's weights, so one MFMA per A piece puts
    // a_p*b_1, a_p*b_2, a_p*b_3 into columns 0..2: 3 MFMAs give all nine products, summed over the columns in the epilogue
    const int bx_lane = i < 3 ? WX_OFF + i * 64 + g * 16 : WZERO_OFF;
    const int w_lane = tid * 16;
    const int dump16 = LDS_DUMP + (tid >= 336 ? tid - 336 : 0) * 16;
    double* red = reinterpret_cast<double*>(smem + LDS_RED);

    // extra (scalar) input channel: lane (i, g) contracts taps 8g..8g+7 of "tap 27"; taps >= 27 are padding
    int xo[8], xkd[8];
    if (EXTRA) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tp = 8 * g + j;
            const int tq = tp > 26 ? 26 : tp;
            xkd[j] = tp > 26 ? -1 : tq / 9;
            xo[j] = (((tq / 3) % 3) * IN_W + tq % 3) * 4;
        }
    }

    while (u < u_end) {
        // ---- column segment [u, seg_end): same (n, h-tile, w-tile), consecutive d ----
        const int col = u / D;
        int d = u - col * D;
        const int twi = col % tiles_w, c2 = col / tiles_w;
        const int thi = c2 % tiles_h, n = c2 / tiles_h;
        const int tw0 = twi * TW, th0 = thi * TH;
        const int seg_end = min(u_end, (col + 1) * D);

        // absent optional operands get an EMPTY descriptor: every load returns 0, so the epilogue needs no branches
        const size_t out_bytes = vol * p.out_stride * 4;
        const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in_main + (size_t)n * vol * p.in_stride, vol * p.in_stride * 4);
        const __amdgpu_buffer_rsrc_t rs_ex = make_rsrc(EXTRA ? p.in_extra + (size_t)n * vol : p.in_main, EXTRA ? vol * 4 : 0);
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(p.out_main + (size_t)n * vol * p.out_stride, out_bytes);
        const __amdgpu_buffer_rsrc_t rs_xout = make_rsrc(NT == 3 ? p.out_extra + (size_t)n * vol : p.out_main, NT == 3 ? vol * 4 : 0);
        const __amdgpu_buffer_rsrc_t rs_res = make_rsrc(p.residual ? p.residual + (size_t)n * vol * p.out_stride : p.out_main, p.residual ? out_bytes : 0);
        const __amdgpu_buffer_rsrc_t rs_res2 = make_rsrc(p.residual2 ? p.residual2 + (size_t)n * vol * p.out_stride : p.out_main, p.residual2 ? out_bytes : 0);
        const __amdgpu_buffer_rsrc_t rs_acc = make_rsrc(p.out_main + (size_t)n * vol * p.out_stride, p.accumulate ? out_bytes : 0);
        const int in_slice_bytes = HW * p.in_stride * 4;
        const int out_plane_bytes = HW * p.out_stride * 4;

        // slice-fill items of this thread: item e = (voxel e/4, chunk e%4): 32 contiguous bytes of the voxel record
        unsigned voff[FIT];
        int loff[FIT];
#pragma unroll
        for (int it = 0; it < FIT; ++it) {
            const int e = tid + it * 512;
            const int vs = e >> 2, c = e & 3;
            const int zy = vs / IN_W, zx = vs - zy * IN_W;
            const int gy = th0 - 1 + zy, gx = tw0 - 1 + zx;
            const bool ok = e < FILL_E && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            voff[it] = ok ? (unsigned)((gy * W + gx) * p.in_stride + c * 8) * 4u : OOB_OFFSET;
            loff[it] = e < FILL_E ? c * CHUNK_BYTES + (vs + 2 * c) * 16 : -1;
        }
        const bool last_item = loff[FIT - 1] >= 0;       // items 0..FIT-2 exist for every thread
        auto fill = [&](int slot_bytes, int it, float4 a, float4 b) {
            const bool real = it < FIT - 1 || last_item;
            const int o0 = real ? slot_bytes + loff[it] : dump16;
            const int st = real ? PIECE_BYTES : 0;
            fill_item(smem, o0, o0 + st, o0 + 2 * st, a, b);
        };
        // scalar-channel slice element of this thread (threads >= 340 have none)
        unsigned voffx = OOB_OFFSET;
        if (EXTRA) {
            const int zy = tid / IN_W, zx = tid - zy * IN_W;
            const int gy = th0 - 1 + zy, gx = tw0 - 1 + zx;
            if (tid < SL_VOX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) voffx = (unsigned)(gy * W + gx) * 4u;
        }
        auto xslot = [&](int sd) { return LDS_XRING + ((sd + 3) % 3) * XSLICE_BYTES; };     // slice number -> ring slot (sd >= -1)
        auto xstore = [&](int sd, float v) {
            const int o = tid < SL_VOX ? xslot(sd) + tid * 4 : dump16;
            *reinterpret_cast<float*>(smem + o) = v;
        };

        // epilogue lane offsets: this wave owns tile row `wave`; M tile m covers columns 16m..16m+15; lane rows 4g..4g+3
        const int ey = th0 + wave;
        unsigned eoff[2][4], eoffx[2][4];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int x = tw0 + 16 * m + 4 * g + r;
                const bool ok = ey < H && x < W;
                eoff[m][r] = ok ? (unsigned)((ey * W + x) * p.out_stride + cbase) * 4u : OOB_OFFSET;
                eoffx[m][r] = (NT == 3 && ok && i == 0) ? (unsigned)(ey * W + x) * 4u : OOB_OFFSET;
            }

        // ---- epilogue of one M tile, in two halves so that its loads are a tap ahead of their use ----
        struct EpiLoads { u32x2 r1[4], r2[4], ac[4]; };
        auto epi_load = [&](EpiLoads& L, int m, int dd, bool live) {
            if (!RES) return;
            const int so = dd * out_plane_bytes;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned eo = live ? eoff[m][r] : OOB_OFFSET;
                if (NB == 2) {
                    L.r1[r] = __builtin_amdgcn_raw_buffer_load_b64(rs_res, eo, so, 0);
                    L.r2[r] = __builtin_amdgcn_raw_buffer_load_b64(rs_res2, eo, so, 0);
                    L.ac[r] = __builtin_amdgcn_raw_buffer_load_b64(rs_acc, eo, so, 0);
                } else {
                    L.r1[r] = (u32x2){__builtin_amdgcn_raw_buffer_load_b32(rs_res, eo, so, 0), 0u};
                    L.r2[r] = (u32x2){__builtin_amdgcn_raw_buffer_load_b32(rs_res2, eo, so, 0), 0u};
                    L.ac[r] = (u32x2){__builtin_amdgcn_raw_buffer_load_b32(rs_acc, eo, so, 0), 0u};
                }
            }
        };
        double s_sum = 0.0, s_sq = 0.0;
        auto epi_finish = [&](const EpiLoads& L, const f32x4 (&a)[2][NT], int m, int dd, bool live) {
            const int so = dd * out_plane_bytes;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned eo = live ? eoff[m][r] : OOB_OFFSET;
                float v0 = a[m][0][r] * sc0 + sh0;
                float v1 = NB == 2 ? a[m][NB - 1][r] * sc1 + sh1 : 0.0f;
                if (STATS) {
                    const double w = eo != OOB_OFFSET ? 1.0 : 0.0;
                    s_sum += w * ((double)v0 + (double)v1);
                    s_sq += w * ((double)v0 * (double)v0 + (double)v1 * (double)v1);
                }
                if (TANH) {
                    if (act0 == ESTD_ACT_TANH) { v0 = tanhf(v0); v1 = tanhf(v1); }
                }
                v0 = fmaxf(v0, relu_floor); v1 = fmaxf(v1, relu_floor);
                if (RES) {
                    const float2 q1 = as_float2(L.r1[r]), q2 = as_float2(L.r2[r]), qa = as_float2(L.ac[r]);
                    v0 = (v0 + q1.x + q2.x) * out_scale + qa.x;
                    v1 = (v1 + q1.y + q2.y) * out_scale + qa.y;
                } else {
                    v0 *= out_scale; v1 *= out_scale;
                }
                if (NB == 2) {
                    const float2 ov = make_float2(v0, v1);
                    u32x2 od; __builtin_memcpy(&od, &ov, 8);
                    __builtin_amdgcn_raw_buffer_store_b64(od, rs_out, eo, so, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), rs_out, eo, so, 0);
                }
                if (NT == 3) {      // 33rd output channel: columns 0..2 of the third N tile hold the b1/b2/b3 partial sums
                    float x2 = a[m][NT - 1][r];
                    x2 += __shfl_down(x2, 1, 16) + __shfl_down(x2, 2, 16);     // lane i = 0 of each 16-lane row gets c0 + c1 + c2
                    const float v2 = fmaxf(x2 * sc2 + sh2, relu_floor2);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v2), rs_xout, live ? eoffx[m][r] : OOB_OFFSET, dd * HW * 4, 0);
                }
            }
        };
        // GroupNorm partials of one tile: lane sums -> wave sums -> LDS scratch (published by the next barrier)
        auto stats_to_lds = [&]() {
            const int grp = (cbase >= 16) ? 1 : 0;
            double a0 = grp == 0 ? s_sum : 0.0, q0 = grp == 0 ? s_sq : 0.0;
            double a1 = grp == 1 ? s_sum : 0.0, q1 = grp == 1 ? s_sq : 0.0;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                a0 += __shfl_xor(a0, o); q0 += __shfl_xor(q0, o);
                a1 += __shfl_xor(a1, o); q1 += __shfl_xor(q1, o);
            }
            if (lane == 0) { red[wave * 4 + 0] = a0; red[wave * 4 + 1] = q0; red[wave * 4 + 2] = a1; red[wave * 4 + 3] = q1; }
            s_sum = 0.0; s_sq = 0.0;
        };
        auto stats_store = [&](int dd, bool live) {
            if (tid < 4 && live) {
                double tot = 0.0;
#pragma unroll
                for (int w8 = 0; w8 < 8; ++w8) tot += red[w8 * 4 + tid];
                const size_t t16 = (((size_t)n * D + dd) * tiles_h + thi) * tiles_w16 + 2 * twi;
                p.stats_partials[t16 * 4 + tid] = tot;
                if (2 * twi + 1 < tiles_w16) p.stats_partials[(t16 + 1) * 4 + tid] = 0.0;
            }
        };

        // ---- prologue: prime ring (slot 0 = slice d-1, slot 1 = slice d), weights of taps 0 and 1, prefetch slice d+1 ----
        lds_barrier();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int sd = d - 1 + s;
            const bool sv = (unsigned)sd < (unsigned)D;
            float4 t0[FIT], t1[FIT];
#pragma unroll
            for (int it = 0; it < FIT; ++it) {
                const unsigned vo = sv ? voff[it] : OOB_OFFSET;
                t0[it] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, sd * in_slice_bytes, 0));
                t1[it] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, sd * in_slice_bytes + 16, 0));
            }
#pragma unroll
            for (int it = 0; it < FIT; ++it) fill(s * SLICE_BYTES, it, t0[it], t1[it]);
        }
        if (EXTRA) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int sd = d - 1 + s;
                const unsigned vo = (unsigned)sd < (unsigned)D ? voffx : OOB_OFFSET;
                xstore(sd, __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ex, vo, sd * HW * 4, 0)));
            }
        }
        float4 pf[2 * FIT];
        {
            const int sd = d + 1;
#pragma unroll
            for (int it = 0; it < FIT; ++it) {
                const unsigned vo = sd < D ? voff[it] : OOB_OFFSET;
                pf[2 * it] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, sd * in_slice_bytes, 0));
                pf[2 * it + 1] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, vo, sd * in_slice_bytes + 16, 0));
            }
        }
        float pfx = 0.f;
        u32x4 wreg;
        {
            const u32x4 w0 = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, 0, 0);
            const u32x4 w1 = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, WTAP_BYTES, 0);
            wreg = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, 2 * WTAP_BYTES, 0);
            *reinterpret_cast<u32x4*>(smem + LDS_W + w_lane) = w0;
            *reinterpret_cast<u32x4*>(smem + LDS_W + WTAP_BYTES + w_lane) = w1;
        }
        lds_barrier();
        bf16x8 acur[2][3], bcur[3][NB], bxcur;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                acur[m][pc] = *reinterpret_cast<const bf16x8*>(smem + pc * PIECE_BYTES + a_lane + (wave * IN_W + 16 * m) * 16);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
            for (int nn = 0; nn < NB; ++nn)
                bcur[pc][nn] = *reinterpret_cast<const bf16x8*>(smem + LDS_W + (pc * NB + nn) * 1024 + b_lane);
        }
        bxcur = *reinterpret_cast<const bf16x8*>(smem + LDS_W + bx_lane);
        lds_barrier();      // nobody overwrites weight slot 0 (tap 2) before every wave has its tap-0 fragments

        int q = 0;          // ring parity: slot q holds slice d-1 (later d+1), slot q^1 holds slice d
        int wsel = 1;       // weight slot holding tap t+1 at the start of tap t
        f32x4 pend[2][NT] = {};
        int pend_d = 0;
        bool have_pend = false;
        EpiLoads el;

        for (; u < seg_end; ++u, ++d) {
            const int sb_even = q * SLICE_BYTES, sb_odd = (q ^ 1) * SLICE_BYTES;
            const bool fetch = (u + 1 < seg_end) && (d + 2 < D);     // slice d+2 exists and is ours to use
            const int next_soff = (d + 2) * in_slice_bytes;

            f32x4 acc[2][NT];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nn = 0; nn < NT; ++nn) acc[m][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef ESTD_TIMELINE
            if (tid == 0 && p.stats_partials) {
                const size_t t16 = (((size_t)n * D + d) * tiles_h + thi) * tiles_w16 + 2 * twi;
                p.stats_partials[t16 * 4 + 0] = (double)__builtin_amdgcn_s_memtime();
                p.stats_partials[t16 * 4 + 1] = (double)blockIdx.x;
                p.stats_partials[t16 * 4 + 2] = (double)wall_clock64();
            }
#endif

            for_each_tap([&](auto tap_c) __attribute__((always_inline)) {
                constexpr int tap = decltype(tap_c)::value;
                constexpr int nt = tap == NTAPS - 1 ? 0 : tap + 1;
                constexpr bool next_is_extra = EXTRA && nt == 27;
                constexpr int nkd = nt / 9, nkh = (nt / 3) % 3, nkw = nt % 3;
                // slot of the NEXT tap's slice: kd 0/2 -> even slot, kd 1 -> odd slot; tap 0 of the next tile -> its even slot = our odd
                const int nsb = (nt == 0) ? sb_odd : (nkd == 1 ? sb_odd : sb_even);
                const int wrd = LDS_W + wsel * WTAP_BYTES, wwr = LDS_W + (wsel ^ 1) * WTAP_BYTES;

                // 1. fragments of the next tap
                bf16x8 anext[2][3], bnext[3][NB], bxnext;
                if constexpr (next_is_extra) {
                    // gather the 8 taps of this lane's k group from the scalar ring and split them in registers
                    const int xs0 = xslot(d - 1), xs1 = xslot(d), xs2 = xslot(d + 1);
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        float xv[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int base = xkd[j] <= 0 ? xs0 : (xkd[j] == 1 ? xs1 : xs2);
                            const float v = *reinterpret_cast<const float*>(smem + base + xo[j] + (wave * IN_W + 16 * m + i) * 4);
                            xv[j] = xkd[j] < 0 ? 0.f : v;
                        }
                        u32x4 h, md, l;
                        unsigned th, tm, tl;
                        split2(xv[0], xv[1], th, tm, tl); h[0] = th; md[0] = tm; l[0] = tl;
                        split2(xv[2], xv[3], th, tm, tl); h[1] = th; md[1] = tm; l[1] = tl;
                        split2(xv[4], xv[5], th, tm, tl); h[2] = th; md[2] = tm; l[2] = tl;
                        split2(xv[6], xv[7], th, tm, tl); h[3] = th; md[3] = tm; l[3] = tl;
                        __builtin_memcpy(&anext[m][0], &h, 16);
                        __builtin_memcpy(&anext[m][1], &md, 16);
                        __builtin_memcpy(&anext[m][2], &l, 16);
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) {
                            if (ESTD_SABL & 2) { anext[m][pc] = acur[m][pc]; asm volatile("" : "+v"(anext[m][pc])); } else
                            anext[m][pc] = *reinterpret_cast<const bf16x8*>(smem + nsb + pc * PIECE_BYTES + a_lane + ((wave + nkh) * IN_W + nkw + 16 * m) * 16);
                        }
                }
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) {
#pragma unroll
                    for (int nn = 0; nn < NB; ++nn) {
                        if (ESTD_SABL & 4) { bnext[pc][nn] = bcur[pc][nn]; asm volatile("" : "+v"(bnext[pc][nn])); } else
                        bnext[pc][nn] = *reinterpret_cast<const bf16x8*>(smem + wrd + (pc * NB + nn) * 1024 + b_lane);
                    }
                }
                bxnext = bxcur;
                if (NT == 3) bxnext = *reinterpret_cast<const bf16x8*>(smem + wrd + bx_lane);
                // 2. weights of tap+2 -> the free weight slot; 3. weights of tap+3 -> registers
                if (!(ESTD_SABL & 8)) {
                    *reinterpret_cast<u32x4*>(smem + wwr + w_lane) = wreg;
                    wreg = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, ((tap + 3) % NTAPS) * WTAP_BYTES, 0);
                }
                // 4. ring: refill the even slot with slice d+1 (taps 8..10), then prefetch slice d+2 (taps 11..16)
                if constexpr (tap >= 8 && tap < 8 + FIT) {
                    if (!(ESTD_SABL & 16)) fill(sb_even, tap - 8, pf[2 * (tap - 8)], pf[2 * (tap - 8) + 1]);
                }
                if constexpr (tap >= 11 && tap < 11 + 2 * FIT) {
                    constexpr int k = tap - 11;
                    if (!(ESTD_SABL & 16))
                        pf[k] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, fetch ? voff[k >> 1] : OOB_OFFSET, next_soff + (k & 1) * 16, 0));
                }
                if constexpr (EXTRA && tap == 17)      // scalar channel: slice d+2 -> register, stored once tap 27's gather (issued in tap 26) is done
                    pfx = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_ex, fetch ? voffx : OOB_OFFSET, (d + 2) * HW * 4, 0));
                if constexpr (EXTRA && tap == 27) xstore(d + 2, pfx);
                // 5. deferred epilogue of the previous tile (all stores are dropped while there is none)
                if (!(ESTD_SABL & 32)) {
                    if constexpr (tap == 0) epi_load(el, 0, pend_d, have_pend);
                    if constexpr (tap == 1) { epi_finish(el, pend, 0, pend_d, have_pend); epi_load(el, 1, pend_d, have_pend); }
                    if constexpr (tap == 2) { epi_finish(el, pend, 1, pend_d, have_pend); if (STATS && ESTD_SPLIT_STATS_ON) stats_to_lds(); }
                    if constexpr (tap == 3) { if (STATS && ESTD_SPLIT_STATS_ON) stats_store(pend_d, have_pend); }
                }
                // 6. this tap's products, smallest terms first; the accumulator chains are interleaved
                {
                    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                    for (int t = 0; t < 6; ++t)
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#pragma unroll
                            for (int nn = 0; nn < NB; ++nn)
                                acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(acur[m][PA[t]], bcur[PB[t]][nn], acc[m][nn], 0, 0, 0);
                    if (NT == 3) {
#pragma unroll
                        for (int pc = 2; pc >= 0; --pc)
#pragma unroll
                            for (int m = 0; m < 2; ++m)
                                acc[m][NT - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(acur[m][pc], bxcur, acc[m][NT - 1], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) acur[m][pc] = anext[m][pc];
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                    for (int nn = 0; nn < NB; ++nn) bcur[pc][nn] = bnext[pc][nn];
                bxcur = bxnext;
                wsel ^= 1;
                tap_pipeline<tap, NT, EXTRA>();
                __builtin_amdgcn_sched_barrier(0);       // nothing of this tap moves past the barrier (asm volatile does not order register-only ops)
                if (ESTD_SABL & 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else
                lds_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }, std::make_integer_sequence<int, NTAPS>{});

#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int nn = 0; nn < NT; ++nn) pend[m][nn] = acc[m][nn];
            pend_d = d;
            have_pend = true;
            q ^= 1;
        }
        if (have_pend) {       // last tile of the segment
            epi_load(el, 0, pend_d, true);
            epi_finish(el, pend, 0, pend_d, true);
            epi_load(el, 1, pend_d, true);
            epi_finish(el, pend, 1, pend_d, true);
            if (STATS && ESTD_SPLIT_STATS_ON) {
                stats_to_lds();
                __syncthreads();
                stats_store(pend_d, true);
            }
        }
    }
}

template <int NT, bool EXTRA, bool TANH, bool STATS, bool RES>
int launch_inst(const estd_conv3d_desc& d, hipStream_t stream, int tiles_w, int tiles_h, int total)
{
    const int slots = estd_persistent_wgs(1);      // one workgroup per CU (LDS-limited)
    int grid = total < slots ? total : slots;
    if (grid >= 8) grid &= ~7;
    estd_allow_dynamic_lds<conv3d_k3_split_kernel<NT, EXTRA, TANH, STATS, RES>>((int)LDS_TOTAL);
    hipLaunchKernelGGL((conv3d_k3_split_kernel<NT, EXTRA, TANH, STATS, RES>), dim3(grid), dim3(512), LDS_TOTAL, stream, d, tiles_w, tiles_h, total);
    return hipGetLastError() == hipSuccess ? ESTD_OK : ESTD_ERR_LAUNCH;
}

template <int NT, bool EXTRA, bool TANH, bool STATS>
int launch(const estd_conv3d_desc& d, hipStream_t stream, int tiles_w, int tiles_h, int total)
{
    // the plain 32 -> 32 convolutions dominate: they get an instance without epilogue loads; the rarer shapes always use RES
    if (NT == 2 && !EXTRA && !d.residual && !d.residual2 && !d.accumulate)
        return launch_inst<NT, EXTRA, TANH, STATS, false>(d, stream, tiles_w, tiles_h, total);
    return launch_inst<NT, EXTRA, TANH, STATS, true>(d, stream, tiles_w, tiles_h, total);
}

}  // namespace

extern "C" int estd_conv3d_k3_split(const estd_conv3d_desc* dp, estd_stream_t s)
{
    if (!dp) return ESTD_ERR_ARG;
    const estd_conv3d_desc& d = *dp;
    hipStream_t stream = static_cast<hipStream_t>(s);
    if (d.N <= 0 || d.D <= 0 || d.H <= 0 || d.W <= 0) return ESTD_ERR_ARG;
    if (!d.in_main || !d.w_split || !d.scale || !d.shift || !d.out_main) return ESTD_ERR_ARG;
    if (d.cin_main != 32 || d.n_tiles < 1 || d.n_tiles > 3 || d.head_w) return ESTD_ERR_UNSUPPORTED;
    if (d.n_tiles == 3 && (!d.out_extra || !d.in_extra)) return ESTD_ERR_ARG;
    if (d.n_tiles < 3 && d.out_extra) return ESTD_ERR_ARG;
    if (d.in_stride < 32 || (d.in_stride & 3) || d.out_stride < (d.n_tiles == 1 ? 16 : 32) || (d.act_split & 1)) return ESTD_ERR_ARG;
    if (d.n_tiles >= 2 && (d.out_stride & 1)) return ESTD_ERR_ARG;
    const int tiles_w = (d.W + TW - 1) / TW, tiles_h = (d.H + TH - 1) / TH;
    const long long total = (long long)d.N * d.D * tiles_h * tiles_w;
    if (total > 0x7fffffffLL) return ESTD_ERR_ARG;
    {
        const long long vox = (long long)d.D * d.H * d.W;
        const int widest = d.in_stride > d.out_stride ? d.in_stride : d.out_stride;
        if (vox * widest * 4 >= 0x7fffff00LL) return ESTD_ERR_UNSUPPORTED;
    }
    const bool tanh_used = (d.act_split > 0 && d.act_a == ESTD_ACT_TANH) || d.act_b == ESTD_ACT_TANH;
    const bool stats = d.stats_partials != nullptr;
    const bool extra = d.in_extra != nullptr;
    const int t = (int)total;
    // instantiated combinations = the ones the decoder / transformer use (hybrid_depth_decoder.py:84-112, epipolar_transformer.py:21)
    if (d.n_tiles == 3) return (!tanh_used && !stats) ? launch<3, true, false, false>(d, stream, tiles_w, tiles_h, t) : ESTD_ERR_UNSUPPORTED;
    if (d.n_tiles == 1) {        // 32 -> 16 (GRU output convolution)
        if (extra || tanh_used) return ESTD_ERR_UNSUPPORTED;
        return stats ? launch<1, false, false, true>(d, stream, tiles_w, tiles_h, t) : launch<1, false, false, false>(d, stream, tiles_w, tiles_h, t);
    }
    if (extra) return stats ? ESTD_ERR_UNSUPPORTED : launch<2, true, true, false>(d, stream, tiles_w, tiles_h, t);
    if (tanh_used) return ESTD_ERR_UNSUPPORTED;
    return stats ? launch<2, false, false, true>(d, stream, tiles_w, tiles_h, t) : launch<2, false, false, false>(d, stream, tiles_w, tiles_h, t);
}
