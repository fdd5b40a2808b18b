// conv3d_wino2_c16.hip -- the 16 -> 16 3x3x3 convolution + folded BatchNorm + activation + fused 1x1x1 head of the two stereo heads
// (hybrid_models/hybrid_depth_decoder.py:96-112: stereo_head0 / stereo_head1 = convbnrelu_3d(16, 16) + Conv3d(16, 1, 1), called at
// :200 and :256 / :377), with depth AND image rows in Winograd F(2,3) form on gfx950 fp32 MFMA -- F(2x2, 3x3) over (d, h), a 3-tap
// direct convolution along w, exactly the arithmetic of csrc/conv3d_wino2.hip (48 tap products per 2 x 2 outputs = 12/27 of the
// direct kernel's MFMA work; U = G g G^T packed on the host in float64, packing.py::pack_conv3d_wino2_c16).  Only the head's
// logit volume leaves the kernel.  Same operator and descriptor as estd_conv3d_k3 with cin_main = 16, n_tiles = 1, head_w set and
// out_main = NULL; reached through estd_conv3d_k3_wino2.
//
// What a 16-channel instance changes against the 32-channel kernel:
//   * a voxel is a 64-byte record: the four depth-transformed slices of an 18 x 18-voxel halo are 81 KB, so the tile is
//     2 planes x 16 rows x 16 columns and every one of the eight waves owns ONE row pair with ALL FOUR depth transforms and the
//     full K = 16: 12 steps (sd, kw) of 16 MFMAs, no cross-wave reduction or exchange (the 16-output-channel instance of the
//     32-channel kernel splits K over the two waves of a SIMD and swaps halves through LDS);
//   * the whole transformed filter is 48 taps x 1 KB = 48 KB: it lives in LDS for the lifetime of the workgroup -- no weight
//     stream through L2 at all (vector-memory traffic = the plane prefetch and the logit stores);
//   * LDS swizzle for 64-byte records: chunk c of voxel v at v * 64 + ((c ^ ((v >> 2) & 3)) << 4).  A 256-byte bank row holds four
//     voxels; with the MFMA column <-> voxel permutation of the other kernels (rows {0-3,12-15} = even voxels, {4-11} = odd) the
//     eight even voxels of a ds_read_b128 lane group fall on two of the four voxel slots with four distinct chunk positions each,
//     the eight odd ones on the other two: conflict-free for every column tap.
//   * epilogue: lane (g, i) holds output channels 4g .. 4g+3 of voxel i of a tile row (transposed MFMAs): BN + activation floor, the
//     head's dot product as 4 in-lane FMAs + two cross-row exchanges (ds_bpermute), one 4-byte store per voxel from lane group 0.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "estd_hip.h"
#include "estd_common.h"

#ifndef ESTD_C16ABL
#define ESTD_C16ABL 0   // timing ablations only (results wrong): 1 no stores, 2 no slice writes, 16 no plane prefetch, 128 no row transform
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));

constexpr int TH = 16, TW = 16;
constexpr int IN_H = TH + 2, IN_W = TW + 2;
constexpr int SL_VOX = IN_H * IN_W;                 // 324 voxels per haloed slice
constexpr int SLICE_BYTES = SL_VOX * 64;            // 16 channels: 20 736
constexpr int SL_CHUNKS = SL_VOX * 4;               // 1 296 16-byte chunks
constexpr int NTHREADS = 512;
constexpr int SIT = (SL_CHUNKS + NTHREADS - 1) / NTHREADS;     // 3 chunks per thread and slice (the last one for 272 threads)
constexpr int NTAPS = 48, TAP_BYTES = 1024;
constexpr int SS_BYTES = 4 * 16 * 4;                // scale | shift | activation floor | head weights of the 16 output channels
constexpr int VTAB_BYTES = SIT * NTHREADS * 4;      // per-thread global offsets of the slice chunks
constexpr int W_OFF = 4 * SLICE_BYTES + SS_BYTES + VTAB_BYTES;
constexpr int LDS_BYTES = W_OFF + NTAPS * TAP_BYTES;            // 138 496
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
static_assert(3 * SLICE_BYTES < 65536 && (NTAPS - 1) * TAP_BYTES < 65536, "ds_read immediate offsets");
constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;
constexpr int NSTEPS = 12;                          // (sd, kw)
constexpr int RB_STEP = 7;                          // slices 0..2 are rewritten in front of this step (rows of step 8 = (sd 2, kw 2) are fetched at the end of step 6)

__device__ __forceinline__ float4 as_float4(u32x4 v) { float4 f; __builtin_memcpy(&f, &v, 16); return f; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, size_t elems)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)(elems * 4), 0x00020000);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int lds_chunk_off(int v, int c) { return v * 64 + ((c ^ ((v >> 2) & 3)) << 4); }
__device__ __forceinline__ float4 f4_sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 f4_add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }

__global__ __launch_bounds__(NTHREADS, 1) void conv3d_wino2_c16_kernel(const estd_conv3d_desc p, int tiles_w, int tiles_h, int dpairs, int total_tiles)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // row pair: tile rows 2 wave, 2 wave + 1 (halo rows 2 wave .. 2 wave + 3)
    const int g = lane >> 4, i = lane & 15;
    const int pi = i < 4 ? 2 * i : i < 12 ? 2 * i - 7 : 2 * i - 16;  // MFMA column <-> voxel of a tile row
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

    float* lds_ss = reinterpret_cast<float*>(smem + 4 * SLICE_BYTES);
    if (tid < 16) {
        lds_ss[tid] = p.scale[tid];
        lds_ss[16 + tid] = p.shift[tid];
        lds_ss[32 + tid] = (tid < p.act_split ? p.act_a : p.act_b) == ESTD_ACT_RELU ? 0.0f : ESTD_NO_FLOOR;
        lds_ss[48 + tid] = p.head_w[tid];
    }
    const float hb = p.head_b[0];
    unsigned* lds_vt = reinterpret_cast<unsigned*>(smem + 4 * SLICE_BYTES + SS_BYTES);            // [it][thread]
    char* lds_w = smem + W_OFF;
    for (int e = tid; e < NTAPS * TAP_BYTES / 16; e += NTHREADS)                                 // visible after the first tile's barriers
        reinterpret_cast<float4*>(lds_w)[e] = reinterpret_cast<const float4*>(p.w_wino2)[e];

    const int row0 = 2 * wave;
    // per-lane LDS offsets of the four halo rows at the three column taps (the depth slice is an immediate)
    int roff[4][3];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) roff[r][kw] = lds_chunk_off((row0 + r) * IN_W + kw + pi, g);
    const int woff = W_OFF + lane * 16;

    while (u < u_end) {
        // ---- column segment [u, seg_end): same (n, h-tile, w-tile), consecutive depth pairs ----
        const int col = u / dpairs;
        int dp = u - col * dpairs;
        const int twi = col % tiles_w, c2 = col / tiles_w;
        const int thi = c2 % tiles_h, n = c2 / tiles_h;
        const int tw0 = twi * TW, th0 = thi * TH;
        const int seg_end = min(u_end, (col + 1) * dpairs);

        const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in_main + (size_t)n * vol * p.in_stride, vol * p.in_stride);
        const __amdgpu_buffer_rsrc_t rs_head = make_rsrc(p.out_head + (size_t)n * vol, vol);
        const int in_slice_bytes = HW * p.in_stride * 4;

        // chunk it of a slice for this thread = chunk tid + it * 512: voxel vs = e >> 2, 16-byte chunk e & 3.  Its LDS offset is
        // loff0 + it * 8 KB exactly (the swizzle key (vs >> 2) & 3 does not change when vs advances by 128).
#pragma unroll
        for (int it = 0; it < SIT; ++it) {
            const int e = tid + it * NTHREADS;
            const int vs = e >> 2, c = e & 3;
            const int zy = vs / IN_W, zx = vs % IN_W;
            const int gy = th0 - 1 + zy, gx = tw0 - 1 + zx;
            const bool ok = e < SL_CHUNKS && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            lds_vt[it * NTHREADS + tid] = ok ? (unsigned)((gy * W + gx) * p.in_stride + c * 4) * 4u : OOB_OFFSET;
        }
        auto chunk_voff = [&](int it) {       // the thread's table slot, re-formed from the lane id (nothing held across the tap loop)
            const int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
            return lds_vt[it * NTHREADS + wave * 64 + l];
        };
        const int loff0 = lds_chunk_off(tid >> 2, tid & 3);
        const bool last_ok = tid + (SIT - 1) * NTHREADS < SL_CHUNKS;
        auto load_plane = [&](int pd, float4 (&dst)[SIT]) {
            const bool pv = (unsigned)pd < (unsigned)D;        // wave-uniform; planes outside the volume are zero padding
#pragma unroll
            for (int it = 0; it < SIT; ++it)
                dst[it] = pv ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, chunk_voff(it), pd * in_slice_bytes, 0))
                             : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        // head output offsets of this lane's voxel in the two tile rows (lane group 0 stores; the others hold an out-of-range offset)
        unsigned hoff[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int y = th0 + row0 + m, x = tw0 + pi;
            hoff[m] = (g == 0 && y < H && x < W) ? (unsigned)(y * W + x) * 4u : OOB_OFFSET;
        }

        float4 xa[SIT], xb[SIT], xc[SIT], xd[SIT];      // raw planes d0-1, d0, d0+1, d0+2
        {
            const int d0 = 2 * dp;
            load_plane(d0 - 1, xa);
            load_plane(d0, xb);
            load_plane(d0 + 1, xc);
            load_plane(d0 + 2, xd);
        }
        auto write_slice = [&](int sl) {                 // depth transform B^T x of the planes in (xa, xb, xc, xd), straight into LDS slice sl
#pragma unroll
            for (int it = 0; it < SIT; ++it) {
                if ((it < SIT - 1 || last_ok) && !(ESTD_C16ABL & 2)) {
                    const float4 v = sl == 0 ? f4_sub(xa[it], xc[it]) : sl == 1 ? f4_add(xb[it], xc[it])
                                   : sl == 2 ? f4_sub(xc[it], xb[it]) : f4_sub(xb[it], xd[it]);
                    *reinterpret_cast<float4*>(smem + loff0 + sl * SLICE_BYTES + it * NTHREADS * 16) = v;
                }
            }
        };
        auto shift_planes = [&]() {
#pragma unroll
            for (int it = 0; it < SIT; ++it) { xa[it] = xc[it]; xb[it] = xd[it]; }
        };
        bool first = true;

        for (; u < seg_end; ++u, ++dp) {
            const int d0 = 2 * dp;
            if (first) {
                lds_barrier();                          // every wave is done reading the previous segment's slices
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) write_slice(sl);
                shift_planes();
                lds_barrier();
                first = false;
            }
            const bool has_next = (u + 1 < seg_end);     // wave-uniform
            const int nd = d0 + 3;                       // new planes of the next tile: nd, nd + 1
            const bool v0 = nd < D, v1 = nd + 1 < D;

            f32x4 acc[4][4];                             // m[sd][sh]; the first product of every accumulator takes C = 0

            auto load_w = [&](int st, float4 (&bq)[4]) { // the four taps (sh) of step st = (sd, kw): tap = st * 4 + sh
#pragma unroll
                for (int t = 0; t < 4; ++t) bq[t] = *reinterpret_cast<const float4*>(smem + woff + (st * 4 + t) * TAP_BYTES);
            };
            auto load_rows = [&](int st, float4 (&Rr)[4]) {
                const int sd = st / 3, kw = st % 3;
#pragma unroll
                for (int r = 0; r < 4; ++r) Rr[r] = *reinterpret_cast<const float4*>(smem + roff[r][kw] + sd * SLICE_BYTES);
            };
            auto xform2 = [&](const float4 (&Rr)[4], int h, f32x2 (&o)[4]) {
                f32x2 r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) r[k] = h == 0 ? (f32x2){Rr[k].x, Rr[k].y} : (f32x2){Rr[k].z, Rr[k].w};
                if (ESTD_C16ABL & 128) { o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; }
                else { o[0] = r[0] - r[2]; o[1] = r[1] + r[2]; o[2] = r[2] - r[1]; o[3] = r[1] - r[3]; }
            };

            // software pipeline inside the wave (as csrc/conv3d_wino2.hip): step s multiplies the fragments T (transformed during step
            // s-1) while the VALU transforms the raw rows of step s+1 (read during step s-1) and then reads the raw rows of step s+2;
            // weights one step ahead, from LDS.
            float4 bq[2][4];
            f32x2 T[2][4];
            float4 R[4];
            load_w(0, bq[0]);
            load_rows(0, R);
            xform2(R, 0, T[0]);
            xform2(R, 1, T[1]);
            load_rows(1, R);
            unsigned vo_next = 0;
            if (has_next && !(ESTD_C16ABL & 16)) vo_next = chunk_voff(0);
            __builtin_amdgcn_sched_barrier(0);

#pragma clang loop unroll(full)
            for (int step = 0; step < NSTEPS; ++step) {
                const int sd = step / 3, kw = step % 3;
                if (has_next && step == RB_STEP) {
                    // slices 0..2 have been read for the last time by every wave; this barrier also publishes slice 3, rewritten behind
                    // the previous tile's loop and first read at the end of this step
                    lds_barrier();
                    write_slice(0);
                    write_slice(1);
                    write_slice(2);
                }
                if (step + 1 < NSTEPS) load_w(step + 1, bq[(step + 1) & 1]);
                // the next tile's two new planes: one chunk per step in steps 0..5
                if (has_next && step < 2 * SIT && !(ESTD_C16ABL & 16)) {
                    const int it = step % SIT;
                    if (step < SIT) {
                        if (it < SIT - 1 || last_ok)
                            xc[it] = v0 ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, vo_next, nd * in_slice_bytes, 0)) : make_float4(0.f, 0.f, 0.f, 0.f);
                    } else {
                        if (it < SIT - 1 || last_ok)
                            xd[it] = v1 ? as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, vo_next, (nd + 1) * in_slice_bytes, 0)) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);       // the loads above are issued BEFORE this step's MFMAs
                f32x2 Tn[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int e = 0; e < 2; ++e)
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            const float4 b4 = bq[step & 1][t];
                            const float b = h == 0 ? (e == 0 ? b4.x : b4.y) : (e == 0 ? b4.z : b4.w);
                            const bool first_product = kw == 0 && h == 0 && e == 0;
                            const f32x4 c_in = first_product ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[sd][t];
                            acc[sd][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, T[h][t][e], c_in, 0, 0, 0);
                        }
                    if (step + 1 < NSTEPS) xform2(R, h, Tn[h]);
                }
                if (step + 2 < NSTEPS) load_rows(step + 2, R);
                if (has_next && step + 1 < 2 * SIT && !(ESTD_C16ABL & 16)) vo_next = chunk_voff((step + 1) % SIT);
#pragma unroll
                for (int h = 0; h < 2; ++h) {            // order of the region: the MFMAs of two components, then the next step's 4 transforms
                    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                if (step + 1 < NSTEPS) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int t = 0; t < 4; ++t) T[h][t] = Tn[h][t];
                }
                __builtin_amdgcn_sched_barrier(0);
            }

            // ---- output transform A^T m A ----
            f32x4 y0[2], y1[2];
            {
                f32x4 z[4][2];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    z[s][0] = acc[s][0] + acc[s][1] + acc[s][2];
                    z[s][1] = acc[s][1] - acc[s][2] - acc[s][3];
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    y0[m] = z[0][m] + z[1][m] + z[2][m];
                    y1[m] = z[1][m] - z[2][m] - z[3][m];
                }
            }
            if (has_next) {
                lds_barrier();                            // every wave has read slice 3 for the last time; slices 0..2 (rewritten in the loop) are visible
                write_slice(3);                           // published by the next tile's in-loop barrier (first read at the end of its step 7)
                shift_planes();
            }
            // ---- epilogue: BN + activation floor, the 1x1x1 head over the 16 channels, one logit per voxel ----
            {
                const float4 sc4 = *reinterpret_cast<const float4*>(lds_ss + 4 * g), sh4 = *reinterpret_cast<const float4*>(lds_ss + 16 + 4 * g);
                const float4 lo4 = *reinterpret_cast<const float4*>(lds_ss + 32 + 4 * g), hw4 = *reinterpret_cast<const float4*>(lds_ss + 48 + 4 * g);
                float part[2][2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const f32x4 a = pl == 0 ? y0[m] : y1[m];
                        float s = fmaxf(fmaf(a[0], sc4.x, sh4.x), lo4.x) * hw4.x;
                        s = fmaf(fmaxf(fmaf(a[1], sc4.y, sh4.y), lo4.y), hw4.y, s);
                        s = fmaf(fmaxf(fmaf(a[2], sc4.z, sh4.z), lo4.z), hw4.z, s);
                        s = fmaf(fmaxf(fmaf(a[3], sc4.w, sh4.w), lo4.w), hw4.w, s);
                        part[pl][m] = s;
                    }
                // sum over the four lane groups (channels 4g..): two exchange levels, all four values of a level back to back
#pragma unroll
                for (int o = 16; o <= 32; o <<= 1) {
                    float t[2][2];
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                        for (int m = 0; m < 2; ++m) t[pl][m] = __shfl_xor(part[pl][m], o);
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                        for (int m = 0; m < 2; ++m) part[pl][m] += t[pl][m];
                }
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    if (d0 + pl < D) {                    // (odd D: the last pair has one plane) wave-uniform
#pragma unroll
                        for (int m = 0; m < 2; ++m)
                            if (!(ESTD_C16ABL & 1))
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, part[pl][m] + hb), rs_head, hoff[m], (d0 + pl) * HW * 4, 0);
                    }
                }
            }
        }
    }
}

}  // namespace

// 16 -> 16 + head instance behind estd_conv3d_k3_wino2 (validated there: cin_main == 16, n_tiles == 1, head, no out_main)
int estd_wino2_c16_launch(const estd_conv3d_desc& d, hipStream_t stream)
{
    const int tiles_w = (d.W + TW - 1) / TW, tiles_h = (d.H + TH - 1) / TH, dpairs = (d.D + 1) / 2;
    const long long total = (long long)d.N * dpairs * tiles_h * tiles_w;
    if (total > 0x7fffffffLL) return ESTD_ERR_ARG;
    if ((long long)d.D * d.H * d.W * d.in_stride * 4 >= 0x7fffff00LL) return ESTD_ERR_UNSUPPORTED;   // 32-bit byte offsets inside one volume
    const int slots = estd_persistent_wgs(1);
    int grid = total < slots ? (int)total : slots;
    if (grid >= 8) grid &= ~7;
    estd_allow_dynamic_lds<conv3d_wino2_c16_kernel>(LDS_BYTES);
    hipLaunchKernelGGL(conv3d_wino2_c16_kernel, dim3(grid), dim3(NTHREADS), LDS_BYTES, stream, d, tiles_w, tiles_h, dpairs, (int)total);
    return ESTD_LAUNCH_CHECK();
}
