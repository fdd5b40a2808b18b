// conv2d_wino.hip -- 3x3 (stride 1, dilation 1 or 2) 2D convolution on NHWC maps with the ROW axis in Winograd F(2,3) form, on gfx950 fp32 MFMA,
// with the folded BatchNorm / ReLU / residual epilogue of csrc/conv2d_mfma.hip (same operator, same descriptor).
//
// the PSMNet matching-feature extractor (networks/psm_submodule.py:14-60,112-114: 50 of these
// convolutions per 5-frame step, ~6.5 ms on the critical path in front of the plane sweep), the ResNet-50 stride-1 3x3
// convolutions (resnet_encoder.py:43-49) and the large 2D-decoder blocks (hybrid_depth_decoder.py:17-30).
//
// F(2,3) along h: two output rows y_a, y_a+1 from four input rows d0..d3 and the three row taps g0, g1, g2 of a kw filter column:
//     t0 = d0 - d2    U0 = g0                   y_a   = m0 + m1 + m2
//     t1 = d1 + d2    U1 = (g0 + g1 + g2) / 2   y_a+1 = m1 - m2 - m3          m_i = conv1d_3(t_i, U_i) along w
//     t2 = d2 - d1    U2 = (g0 - g1 + g2) / 2
//     t3 = d1 - d3    U3 = g2
// 4 x 3 tap products instead of 2 x 9: 2/3 of the MFMA work, every product still a v_mfma_f32_16x16x4_f32 with fp32
// accumulation (U packed on the host in fp64, estdepth_amd/packing.py::pack_conv2d_wino).
//
// Structure = the direct kernel's: 256-thread persistent workgroups (2 per CU), 8x16-pixel output tiles, the input consumed in
// 32-channel chunks through a 2-slot LDS ring with ONE LDS-only barrier per chunk, next chunk (possibly of the next tile)
// prefetched into registers during the MFMAs.  Differences:
//   * wave w owns output rows 2w, 2w+1; its four transformed rows t0..t3 are four 16-pixel M tiles that meet U_i[kw] in
//     accumulators m0..m3 (x NT channel tiles): the output transform happens once, after the last chunk;
//   * the brick is held by the threads as COLUMNS (thread = (column, 16-byte chunk), ten rows in registers) so that the input
//     transform is register arithmetic when the chunk is written to LDS: the LDS slot holds 4 row pairs x 4 transformed rows.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>
#include <type_traits>

#include "estd_hip.h"
#include "estd_common.h"

#ifndef ESTD_W2ABL
#define ESTD_W2ABL 0    // timing ablations only (wrong results when != 0): 1 no output stores, 2 no transform writes, 8 no weight stream,
#endif                  // 16 no brick prefetch
#ifndef ESTD_C2TIME
#define ESTD_C2TIME 0    // debug build: s_memtime stamps of the first work items of the first workgroups into the RESIDUAL buffer
#endif                   // (tools/conv2d_timeline.py; results are wrong)
#ifndef ESTD_C2_PFR
#define ESTD_C2_PFR 1    // brick rows requested per tap (A/B: 2 = all rows in taps 0-4: no gain alone, +0.3 ms in the step)
#endif
#ifndef ESTD_C2SCHED
#define ESTD_C2SCHED 1   // explicit issue order inside a tap (A/B switch)
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
typedef unsigned int u32x2 __attribute__((__vector_size__(8)));
constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;

constexpr int TH = 8, TW = 16;
constexpr int TROWS = 16;                            // 4 row pairs x 4 transformed rows

__device__ __forceinline__ float4 as_float4(u32x4 v) { float4 f; __builtin_memcpy(&f, &v, 16); return f; }
__device__ __forceinline__ u32x4 as_u32x4(float4 f) { u32x4 v; __builtin_memcpy(&v, &f, 16); return v; }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, size_t elems)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(elems * 4), 0x00020000);
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#if ESTD_C2TIME
#define C2STAMP(slot) do { if (lane == 0 && blockIdx.x < 16 && item_no < 8)                                                       \
        reinterpret_cast<unsigned long long*>(const_cast<float*>(p.residual))[((blockIdx.x * 8 + item_no) * 4 + wave) * 8 + (slot)] = \
            __builtin_amdgcn_s_memtime(); } while (0)
#else
#define C2STAMP(slot) do { } while (0)
#endif

__device__ __forceinline__ int lds_chunk_off(int v, int c) { return v * 128 + ((c ^ ((v >> 1) & 7)) << 4); }

// DIL = 2: rows of equal parity form the F(2,3) sequences -- wave w owns output rows a, a + 2 with a = (w & 1) + 4 (w >> 1),
// fed by brick rows a, a+2, a+4, a+6; column taps are two pixels apart.
template <int NT, int DIL>
__global__ __launch_bounds__(256, 2) void conv2d_wino_kernel(const estd_conv2d_desc p, int tiles_w, int tiles_h, int total_items)
{
    constexpr int QN = 2 * NT;                        // weight quads per lane per tap
    constexpr int IN_H = TH + 2 * DIL, IN_W = TW + 2 * DIL;      // haloed brick: 10 x 18 (dilation 1) / 12 x 20 (dilation 2)
    constexpr int SLOT_BYTES = TROWS * IN_W * 128;               // 36 864 / 40 960
    constexpr int LOADERS = IN_W * 8;                            // 144 / 160 threads hold the brick: (column, 16-byte chunk)
    constexpr int PFR = ESTD_C2_PFR;                             // brick rows requested per tap
    static_assert(IN_H <= 12 * PFR && 2 * (QN + PFR) <= 8 * NT, "prefetch schedule");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // row pair: output rows a_w, a_w + DIL
    const int a_w = DIL == 1 ? 2 * wave : (wave & 1) + 4 * (wave >> 1);
    const int g = lane >> 4, col = lane & 15;
    // MFMA row <-> pixel of a tile row (see csrc/conv3d_wino.hip): rows {0-3,12-15} = even pixels, rows {4-11} = odd pixels, so that the
    // two halves of a ds_read_b128 lane group never meet in a bank row -- conflict-free A reads for every tap parity.
    const int pcol = (ESTD_W2ABL & 32) ? col : (col < 4 ? 2 * col : col < 12 ? 2 * col - 7 : 2 * col - 16);
    const int px0 = (ESTD_W2ABL & 32) ? 4 * g : (g == 0 ? 0 : g == 1 ? 1 : g == 2 ? 9 : 8);      // pixel of D row 4g + r = px0 + pxs * r
    const int pxs = (ESTD_W2ABL & 32) ? 1 : 2;
    const int H = p.H, W = p.W, Cin = p.cin, Cout = p.cout;
    const int nchunks = Cin >> 5;
    const int tiles_per_group = p.N * tiles_h * tiles_w;
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
    const bool loader = tid < LOADERS;
    const int lzx = tid >> 3, lc = tid & 7;           // this loader thread's brick column and 16-byte chunk

    auto decode = [&](int item, int& grp, int& n, int& th0, int& tw0) {
        grp = item / tiles_per_group;
        int t = item - grp * tiles_per_group;
        const int twi = t % tiles_w; t /= tiles_w;
        const int thi = t % tiles_h; n = t / tiles_h;
        th0 = thi * TH; tw0 = twi * TW;
    };
    // per-thread source offsets (bytes inside one image, chunk 0) of the ten brick rows of a tile; OOB -> zeros
    auto brick_offsets = [&](int th0, int tw0, unsigned (&voff)[IN_H], bool enable) {
        // one vector multiply-add for the whole brick: the row term is wave-uniform (scalar unit), the row validity too
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
    // LDS byte offset of (row pair w, transformed row i) at this loader thread's (column, chunk)
    int loff[4][4];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int i = 0; i < 4; ++i) loff[w][i] = lds_chunk_off((w * 4 + i) * IN_W + (loader ? lzx : 0), lc);

    int grp, n, th0, tw0;
    decode(u, grp, n, th0, tw0);
    unsigned voff[IN_H];
    brick_offsets(th0, tw0, voff, true);
    __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in + (size_t)n * img_in, img_in);
    float4 pf[IN_H];
#pragma unroll
    for (int zy = 0; zy < IN_H; ++zy) pf[zy] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[zy], 0, 0));

    // Weight stream: WD taps ahead of the MFMAs through a ring of WD + 1 register sets, CONTINUOUS across chunks and work items -- the
    // first WD taps of the next chunk (or of the next item's first chunk) are requested during the last WD taps of this one, so no
    // chunk starts by waiting for an L2 round trip.  One tap = 8 NT MFMAs = 256 NT matrix cycles: NT = 2 needs two taps of cover.
    constexpr int WD = (NT == 2) ? 2 : 1, WR = WD + 1;
    const size_t wgrp_elems = (size_t)nchunks * 13 * QN * 256;          // packed floats per output group (12 taps + 1 pad)
    float4 bq[WR][QN];
    {
        const __amdgpu_buffer_rsrc_t rs_w0 = make_rsrc(p.w_wino + (size_t)grp * wgrp_elems, wgrp_elems);
#pragma unroll
        for (int t = 0; t < WD; ++t)
#pragma unroll
            for (int q = 0; q < QN; ++q) bq[t][q] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w0, wlane, (t * QN + q) * 1024, 0));
    }
    const float floor_b = p.relu_before_residual ? 0.f : ESTD_NO_FLOOR;
    const float floor_a = p.relu_after_residual ? 0.f : ESTD_NO_FLOOR;

    int k = 0;                                              // global chunk counter -> LDS slot
    int item_no = 0;
    while (true) {
        C2STAMP(0);
        const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w_wino + (size_t)grp * wgrp_elems, wgrp_elems);

        // folded BatchNorm of this lane's channels: requested HERE, a whole item ahead of the epilogue that uses it (requested there,
        // every item ended with an exposed L2 round trip: 900 cycles alone, 4 000-7 000 next to a busy memory pipeline)
        float sc[NT], sh[NT];
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) { sc[nn] = p.scale[grp * 16 * NT + NT * col + nn]; sh[nn] = p.shift[grp * 16 * NT + NT * col + nn]; }
        // m0..m3 of this wave's row pair, summed over the input chunks.  No zero fill (8 NT register moves per item, paid in matrix
        // time): the first chunk is a separate instance of the chunk body whose first product per accumulator takes C = 0.
        f32x4 acc[4][NT];

        const bool has_next_item = (u + 1 < u_end);
        int ngrp = grp, nn_ = n, nth0 = th0, ntw0 = tw0;
        if (has_next_item) decode(u + 1, ngrp, nn_, nth0, ntw0);
        const __amdgpu_buffer_rsrc_t rs_wni = make_rsrc(p.w_wino + (size_t)ngrp * wgrp_elems, wgrp_elems);

        auto chunk_body = [&](auto first_c, const int c) {
            constexpr bool FIRST = decltype(first_c)::value;
            char* slot = smem + (k & 1) * SLOT_BYTES;
            // ---- input transform B^T d along rows, from the column registers straight into the slot ----
            if (loader && !(ESTD_W2ABL & 2)) {
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const int a = DIL == 1 ? 2 * w : (w & 1) + 4 * (w >> 1);           // compile-time after unrolling
                    const float4 d0 = pf[a], d1 = pf[a + DIL], d2 = pf[a + 2 * DIL], d3 = pf[a + 3 * DIL];
                    *reinterpret_cast<float4*>(slot + loff[w][0]) = f4_sub(d0, d2);
                    *reinterpret_cast<float4*>(slot + loff[w][1]) = f4_add(d1, d2);
                    *reinterpret_cast<float4*>(slot + loff[w][2]) = f4_sub(d2, d1);
                    *reinterpret_cast<float4*>(slot + loff[w][3]) = f4_sub(d1, d3);
                }
            }
            if (c == 0) C2STAMP(1);
            lds_barrier();
            if (c == 0) C2STAMP(2);

            // what to prefetch while this chunk computes: the next 32-channel chunk of the same pixels, or chunk 0 of the next item.
            // Branch-free: with nothing left to fetch every offset is out of bounds (the loads return zeros without touching memory).
            const bool last_chunk = (c + 1 == nchunks);
            int pf_soff = (c + 1) * 128;
            if (last_chunk) {
                brick_offsets(nth0, ntw0, voff, has_next_item);
                if (nn_ != n) rs_in = make_rsrc(p.in + (size_t)nn_ * img_in, img_in);
                pf_soff = 0;
            }
            const int wbase = c * 13 * QN;                                 // quads of this chunk's first tap
            const __amdgpu_buffer_rsrc_t rs_wx = last_chunk ? rs_wni : rs_w;     // where the stream goes on after tap 11
            const int wnext = last_chunk ? 0 : wbase + 13 * QN;

            // A fragment of (transformed row i, column tap kw): two 16-byte LDS reads (channels 4g.., 16+4g..)
            auto load_a = [&](int tap, float4& a0, float4& a1) {
                const int i = tap / 3, kw = tap % 3;
                const int off0 = lds_chunk_off((wave * 4 + i) * IN_W + kw * DIL + pcol, g);
                a0 = *reinterpret_cast<const float4*>(slot + off0);
                a1 = *reinterpret_cast<const float4*>(slot + (off0 ^ 64));
            };
            float4 af[2][2];
            load_a(0, af[0][0], af[0][1]);

#pragma clang loop unroll(full)
            for (int tap = 0; tap < 12; ++tap) {
                const int i = tap / 3;
                const int tgt = tap + WD;
#pragma unroll
                for (int q = 0; q < QN; ++q) {
                    if (ESTD_W2ABL & 8) continue;
                    bq[tgt % WR][q] = tgt < 12
                        ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane, (wbase + tgt * QN + q) * 1024, 0))
                        : as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_wx, wlane, (wnext + (tgt - 12) * QN + q) * 1024, 0));
                }
                // the next brick: PFR rows per tap from tap 0 on.  (tools/conv2d_timeline.py shows the transform of the next chunk waiting
                // thousands of cycles for these rows; requesting them earlier -- PFR = 2, 4 -- only moves that wait into the tap loop or
                // the epilogue: the two workgroups of a CU take turns on the matrix pipe and the sum per work item does not change)
                if (tap * PFR < IN_H && !(ESTD_W2ABL & 16)) {
#pragma unroll
                    for (int r = 0; r < PFR; ++r)
                        if (tap * PFR + r < IN_H)
                            pf[tap * PFR + r] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, voff[tap * PFR + r], pf_soff, 0));
                }
                if (tap + 1 < 12) load_a(tap + 1, af[(tap + 1) & 1][0], af[(tap + 1) & 1][1]);   // LDS latency under this tap's MFMAs
                const float4 a0c = af[tap & 1][0], a1c = af[tap & 1][1];
                const float av[8] = {a0c.x, a0c.y, a0c.z, a0c.w, a1c.x, a1c.y, a1c.z, a1c.w};
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if ((ESTD_W2ABL & 64) && tap % 3 == 2) continue;      // timing ablation: 2/3 of the MFMAs (what a second Winograd axis would leave)
#pragma unroll
                    for (int nn = 0; nn < NT; ++nn) {
                        const int idx = ks * NT + nn;
                        const float4 bv = bq[tap % WR][idx >> 2];
                        const float b = (idx & 3) == 0 ? bv.x : (idx & 3) == 1 ? bv.y : (idx & 3) == 2 ? bv.z : bv.w;
                        const f32x4 cin_ = (FIRST && tap % 3 == 0 && ks == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[i][nn];
                        acc[i][nn] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], b, cin_, 0, 0, 0);
                    }
                }
#if ESTD_C2SCHED
                // issue order of a tap: the next tap's two LDS reads first, then one memory request per two MFMAs
                __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int q = 0; q < QN + PFR; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 8 * NT - 2 * (QN + PFR), 0);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        chunk_body(std::true_type{}, 0);
        ++k;
        for (int c = 1; c < nchunks; ++c, ++k) chunk_body(std::false_type{}, c);

        C2STAMP(3);
        // ---- output transform A^T m, then the direct kernel's epilogue: lane = column j (N index) x pixels 4g..4g+3 of the two
        //      output rows; its channels are grp*16*NT + NT*j .. +NT-1.  Two instances (with / without residual), the activations as
        //      floors (0 or -inf): no branch per pixel ----
        auto epilogue = [&](auto has_res_c) {
            constexpr bool HAS_RES = decltype(has_res_c)::value;
            const int cb = grp * 16 * NT + NT * col;
            const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(p.out + (size_t)n * img_out, img_out);
            const __amdgpu_buffer_rsrc_t rs_res = make_rsrc((HAS_RES ? p.residual : p.out) + (size_t)n * img_out, img_out);
            const float floor_1 = HAS_RES ? floor_b : fmaxf(floor_b, floor_a);
            unsigned eo[2][4];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int y = th0 + a_w + m * DIL;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int x = tw0 + px0 + pxs * r;
                    eo[m][r] = (y < H && x < W) ? (unsigned)((y * W + x) * Cout + cb) * 4u : OOB_OFFSET;
                }
            }
            // the residual pixels: loads issued back to back and waited for once per batch (one load -> wait -> store per pixel
            // serialises eight memory round trips per tile).  NT = 2: both rows in one batch; NT = 4: a row at a time (registers).
            float rres[2][4][NT];
            auto load_res_row = [&](int m) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (NT == 4) {
                        const float4 rr = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_res, eo[m][r], 0, 0));
                        rres[m][r][0] = rr.x; rres[m][r][1] = rr.y; rres[m][r][NT - 2] = rr.z; rres[m][r][NT - 1] = rr.w;
                    } else {
                        const u32x2 rv = __builtin_amdgcn_raw_buffer_load_b64(rs_res, eo[m][r], 0, 0);
                        float2 rr; __builtin_memcpy(&rr, &rv, 8);
                        rres[m][r][0] = rr.x; rres[m][r][1] = rr.y;
                    }
                }
            };
            if (NT == 2 && HAS_RES) { load_res_row(0); load_res_row(1); }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (NT == 4 && HAS_RES) load_res_row(m);
                f32x4 yv[NT];
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
                    yv[nn] = m == 0 ? acc[0][nn] + acc[1][nn] + acc[2][nn] : acc[1][nn] - acc[2][nn] - acc[3][nn];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v[NT];
#pragma unroll
                    for (int nn = 0; nn < NT; ++nn) {
                        v[nn] = __builtin_fmaxf(__builtin_fmaf(yv[nn][r], sc[nn], sh[nn]), floor_1);
                        if (HAS_RES) v[nn] = __builtin_fmaxf(v[nn] + rres[m][r][nn], floor_a);
                    }
                    if (ESTD_W2ABL & 1) {
                        asm volatile("" :: "v"(v[0]), "v"(v[NT - 1]));
                    } else if (NT == 4) {
                        __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(make_float4(v[0], v[1], v[2], v[3])), rs_out, eo[m][r], 0, 0);
                    } else {
                        const float2 ov = make_float2(v[0], v[1]);
                        u32x2 od; __builtin_memcpy(&od, &ov, 8);
                        __builtin_amdgcn_raw_buffer_store_b64(od, rs_out, eo[m][r], 0, 0);
                    }
                }
            }
        };
        if (p.residual && !ESTD_C2TIME) epilogue(std::true_type{}); else epilogue(std::false_type{});
        C2STAMP(4);
        ++item_no;

        if (!has_next_item) break;
        ++u;
        grp = ngrp; n = nn_; th0 = nth0; tw0 = ntw0;
    }
}

template <int NT, int DIL>
int launch2d(const estd_conv2d_desc& d, hipStream_t stream)
{
    const int tiles_w = (d.W + TW - 1) / TW, tiles_h = (d.H + TH - 1) / TH;
    const int groups = d.cout / (16 * NT);
    const int total = groups * d.N * tiles_h * tiles_w;
    const size_t lds = (size_t)2 * TROWS * (TW + 2 * DIL) * 128;      // two slots: 72 KB / 80 KB -> two workgroups per CU
    static const int grid_mult = [] { const char* e = getenv("ESTD_C2_GRID_MULT"); const int v = e ? atoi(e) : 1; return v >= 1 ? v : 1; }();
    const int slots = estd_persistent_wgs(2) * grid_mult;   // > 1: more workgroups than fit at once, the hardware dispatcher balances
    int grid = total < slots ? total : slots;
    if (grid >= 8) grid &= ~7;
    estd_allow_dynamic_lds<conv2d_wino_kernel<NT, DIL>>((int)lds);
    hipLaunchKernelGGL((conv2d_wino_kernel<NT, DIL>), dim3(grid), dim3(256), lds, stream, d, tiles_w, tiles_h, total);
    return hipGetLastError() == hipSuccess ? ESTD_OK : ESTD_ERR_LAUNCH;
}

}  // namespace

extern "C" int estd_conv2d_k3_wino(const estd_conv2d_desc* dp, estd_stream_t s)
{
    if (!dp) return ESTD_ERR_ARG;
    const estd_conv2d_desc& d = *dp;
    if (d.N <= 0 || d.H <= 0 || d.W <= 0 || !d.in || !d.w_wino || !d.scale || !d.shift || !d.out) return ESTD_ERR_ARG;
    if (d.cin < 32 || (d.cin & 31) || d.cout < 32 || (d.cout & 31)) return ESTD_ERR_ARG;
    if (d.dilation != 1 && d.dilation != 2) return ESTD_ERR_UNSUPPORTED;
    if (d.group_tiles != 2 && d.group_tiles != 4) return ESTD_ERR_ARG;
    if (d.cout % (16 * d.group_tiles)) return ESTD_ERR_ARG;
    const long long widest = (long long)d.H * d.W * (d.cin > d.cout ? d.cin : d.cout) * 4;
    if (widest >= 0x7fffff00LL) return ESTD_ERR_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(s);
    if (d.dilation == 1) return d.group_tiles == 4 ? launch2d<4, 1>(d, stream) : launch2d<2, 1>(d, stream);
    return d.group_tiles == 4 ? launch2d<4, 2>(d, stream) : launch2d<2, 2>(d, stream);
}
