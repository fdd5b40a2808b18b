// conv3d_wino2x.hip -- the 32 -> 32 3x3x3 convolution in F(2x2, 3x3) Winograd form over (depth, image rows) on gfx950 fp32 MFMA,
// rebuilt around OPERAND REUSE: one 512-register wave per SIMD, v_mfma_f32_32x32x2_f32 (32 voxels x 32 output channels per
// product: every A and every B register feeds twice the multiply-adds of the 16x16x4 form), both Winograd transforms applied
// BEFORE the LDS (the tap loop is loads + MFMAs only) and wave-private LDS blocks (no barrier inside the tap loop).
//
// Same operator and descriptor as estd_conv3d_k3_wino2's plain 32 -> 32 instance (networks/layers_op.py:16-39 as used at
// hybrid_models/model_hybrid.py:59-60,:95, hybrid_models/hybrid_depth_decoder.py:84-95,:190-191 and the gate convolution of
// transformer/epipolar_transformer.py:21); same arithmetic (csrc/conv3d_wino2.hip header: T = B^T x B on the 4 x 4 (d, h) patch,
// U = G g G^T in float64 on the host, m[sd][sh] = conv1d_w(T[sd][sh], U[sd][sh][kw]), y = A^T m A; 12/27 of the direct products).
//
// What was wrong with the 8-wave form (profiles/r4_wino2_ablation.txt): every v_mfma_f32_16x16x4_f32 consumed a
// fresh A register (weights, through L1) and a fresh B register (voxels, through LDS + a row transform in registers that each of
// the two channel-half waves and each of the three column taps repeated: 6x redundant), two barriers per tile kept the two waves
// of a SIMD in the same phase, and 1 240 VALU instructions per SIMD and tile were paid in matrix-pipe time (the fp32 MFMA hides
// no VALU work).  Here, per tile of 2 planes x 8 rows x 16 columns and workgroup of 4 waves:
//   * wave (rpp, sdh) owns the row pairs 2 rpp, 2 rpp + 1 (MFMA column = (row pair, column): 32 voxels), ALL 32 output channels
//     (MFMA rows) and the depth transforms sd = 2 sdh, 2 sdh + 1: 8 products m[sd][sh] x 16 registers = 128 accumulators;
//     12 steps (sd, kw, 16-channel chunk) of 32 MFMAs = 384 x 64 cycles = the same 24 576 matrix cycles per SIMD and tile;
//   * per step 8 ds_read_b128 (B: voxels) + 8 buffer_load_b128 (A: weights) feed 32 MFMAs of 2 048 multiply-adds: half the LDS
//     bytes and half the weight bytes per multiply-add of the 8-wave form, no VALU instruction between LDS and MFMA;
//   * the FULLY transformed operands (depth AND row transform) are written to LDS by the wave that consumes them:
//     block (sd, row pair) = [4 sh][18 columns][32 channels] = 9 KB, 16 blocks = 144 KB, each read by exactly one wave -- the
//     only workgroup synchronisation left is the pairwise exchange behind the tap loop.  Transform work per wave and tile: 288
//     items (row pair, column, 16-byte piece) x 2 depth transforms x (4 depth adds + 4 row adds) on float4 = 320 VALU, spread over
//     the MFMA stream one round (64 items) per step, its plane loads requested one step (2 048 matrix cycles) ahead;
//   * the depth half of the output transform crosses the two waves of a row-pair pair: wave sdh = 0 holds z0, z1 and finishes plane
//     d0 (y0 = z0 + z1 + z2), wave sdh = 1 holds z2, z3 and finishes plane d0 + 1 (y1 = z1 - z2 - z3): each sends ONE z (32
//     registers, 8 KB) through the LDS block it has just finished reading.
// Raw planes are re-read per tile (3 of the 4 planes of a tile were read by the previous tile of the column segment: L1 / L2
// hits) instead of being carried in registers: HBM traffic is unchanged, L2 -> L1 traffic per tile 624 KB (8-wave form: 590 KB).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "estd_hip.h"
#include "estd_common.h"

#ifndef ESTD_W2X_TOUCH
#define ESTD_W2X_TOUCH 0     // measured: +5 % (0.869 -> 0.913 ms): the touch loads themselves sit in the in-order return queue in front of the next tile's weights
#endif
#ifndef ESTD_W2X_SCHED
#define ESTD_W2X_SCHED 1    // 0: requests clustered in front of the MFMAs of a half step; 1: one request / LDS instruction / VALU group per MFMA gap
#endif
#ifndef ESTD_W2X_PK
#define ESTD_W2X_PK 1       // production transforms as packed fp32 instructions (inline assembly)
#endif
#ifndef ESTD_W2X_VPG
#define ESTD_W2X_VPG 3
#endif
#ifndef ESTD_W2XABL
#define ESTD_W2XABL 0   // timing ablations only (results are wrong): 1 no output stores, 2 no operand production (plane loads, transforms,
#endif                  // LDS writes), 4 no weight loads, 8 no B reads, 16 no exchange / barriers, 32 no epilogue arithmetic

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));

constexpr int TH = 8, TW = 16, IN_W = TW + 2;
constexpr int ROW_B = IN_W * 128;              // one transformed row: 18 voxels x 32 channels
constexpr int BLK_B = 4 * ROW_B;               // block (sd, row pair): 4 row-transform indices
constexpr int HALF_B = 2 * BLK_B;              // the two row pairs of one depth transform
constexpr int WAVE_B = 2 * HALF_B;             // a wave's two depth transforms: 36 864 bytes
constexpr int SL_BYTES = 4 * WAVE_B;           // 147 456
constexpr int SS_BYTES = 3 * 32 * 4;           // folded BN scale | shift | activation floor
constexpr int RED_BYTES = 2 * 2 * 2 * 2 * 2 * 8;   // GroupNorm scratch: [tile parity][plane][row-pair pair][group][sum, sumsq] doubles
constexpr int LDS_BYTES = SL_BYTES + SS_BYTES + RED_BYTES;
static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
constexpr int ITEMS = 2 * IN_W * 8;            // (row pair, column, piece) per wave: 288
constexpr int ROUNDS = 5;                      // of 64 items (the last one half full)
constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;   // beyond num_records of any descriptor: loads return 0, stores are dropped
constexpr int HALF_W_BYTES = 4 * 1024;         // weights of one half step: [4 sh][64 lanes][4]

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
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__device__ __forceinline__ int colkey_off(int col, int piece) { return col * 128 + ((piece ^ ((col >> 1) & 7)) << 4); }

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

// RBK: read-back streams of the epilogue.  0 none; 1 running sum (out += result); 2 residual(s) + scale; 3 both.
// STATS: GroupNorm(1 group per 16 channels) partial sums of the raw outputs (the ConvGRU gate convolution).
template <int RBK, bool STATS>
__global__ __launch_bounds__(256, 1) void conv3d_wino2x_kernel(const estd_conv3d_desc p, int tiles_w, int tiles_h, int dpairs, int total_tiles)
{
    constexpr bool RB_ACC = RBK == 1 || RBK == 3, RB_RES = RBK == 2 || RBK == 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rpp = wave & 1;                   // row pairs 2 rpp, 2 rpp + 1: tile rows 4 rpp .. 4 rpp + 3, halo rows 4 rpp .. 4 rpp + 5
    const int sdh = wave >> 1;                  // depth transforms 2 sdh, 2 sdh + 1; finishes output plane d0 + sdh
    const int k2 = lane >> 5;                   // k index of the 32x32x2 MFMA
    const int iv = lane & 31;                   // MFMA column: voxel (row pair rpl, column pi)
    const int rpl = iv >> 4, ii = iv & 15;
    // MFMA column <-> voxel column of a tile row (conflict-free ds_read_b128 for every column tap: the hardware services a
    // ds_read_b128 in the lane groups {0-3,12-15,20-27}, ...; with this permutation a group reads 8 even columns of one row pair
    // and 8 odd columns of the other, and the column-keyed swizzle gives the 8 columns of equal parity 8 distinct 16-byte slots)
    const int pi = ii < 4 ? 2 * ii : ii < 12 ? 2 * ii - 7 : 2 * ii - 16;
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

    float* lds_ss = reinterpret_cast<float*>(smem + SL_BYTES);
    if (tid < 64) lds_ss[tid] = tid < 32 ? p.scale[tid] : p.shift[tid - 32];
    if (tid >= 64 && tid < 96) lds_ss[tid] = ((tid - 64) < p.act_split ? p.act_a : p.act_b) == ESTD_ACT_RELU ? 0.0f : ESTD_NO_FLOOR;
    double* lds_red = reinterpret_cast<double*>(smem + SL_BYTES + SS_BYTES);

    const int wbase = wave * WAVE_B;
    // B fragment reads: byte offset of (row pair rpl, column pi + kw, piece 4 c + 2 q + k2) in block (sdl = 0, sh = 0)
    int rd[3][2][2];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q) rd[kw][c][q] = wbase + rpl * BLK_B + colkey_off(pi + kw, 4 * c + 2 * q + k2);
    // operand production: item = (row pair, column, piece); lane's item of round t
    int wr[ROUNDS];
#pragma unroll
    for (int t = 0; t < ROUNDS; ++t) {
        int id = t * 64 + lane;
        if (id >= ITEMS) id -= 32;              // the idle lanes of the last round repeat lanes 0..31 (same value to the same address)
        const int rp_i = id >= 144 ? 1 : 0, rem = id - 144 * rp_i;
        wr[t] = wbase + rp_i * BLK_B + colkey_off(rem >> 3, rem & 7);
    }

    // packed weights [4 sd][3 kw][2 c][2 q][4 sh][64 lanes][4] (packing.pack_conv3d_wino2x)
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w_wino2, (size_t)4 * 3 * 2 * 8 * 256);
    const __amdgpu_buffer_rsrc_t rs_null = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w_wino2), 0, 0, 0x00020000);
    const int wlane = lane * 16;
    const int w_wave = sdh * (2 * 3 * 2 * 2 * HALF_W_BYTES);

    const int in_slice_bytes = HW * p.in_stride * 4;
    const int out_plane_bytes = HW * p.out_stride * 4;

    // ---- production cursor (tile whose operands are being formed) ----
    int pcol = -1, pdp = 0, pn = 0, pthi = 0, ptwi = 0;
    unsigned gvo[ROUNDS][4];
    unsigned tvo[2];                            // L2 touch: this wave's quarter of the 2 x 10 x 18 voxel records two new planes bring
    __amdgpu_buffer_rsrc_t rs_in = rs_null;
    auto set_prod = [&](int uu) {
        const int col = uu / dpairs;
        pdp = uu - col * dpairs;
        if (col != pcol) {
            pcol = col;
            ptwi = col % tiles_w;
            const int c2 = col / tiles_w;
            pthi = c2 % tiles_h;
            pn = c2 / tiles_h;
            rs_in = make_rsrc(p.in_main + (size_t)pn * vol * p.in_stride, vol * p.in_stride);
            const int th0 = pthi * TH, tw0 = ptwi * TW;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int l = lane + 64 * j, L = 90 * wave + l;
                const int pl = L >= 180 ? 1 : 0, rem = L - 180 * pl;
                const int gy = th0 - 1 + rem / IN_W, gx = tw0 - 1 + rem % IN_W;
                const bool ok = l < 90 && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                tvo[j] = ok ? (unsigned)((gy * W + gx) * p.in_stride) * 4u + (unsigned)(pl * in_slice_bytes) : OOB_OFFSET;
            }
#pragma unroll
            for (int t = 0; t < ROUNDS; ++t) {
                int id = t * 64 + lane;
                if (id >= ITEMS) id -= 32;
                const int rp_i = id >= 144 ? 1 : 0, rem = id - 144 * rp_i;
                const int gx = tw0 - 1 + (rem >> 3);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gy = th0 - 1 + 4 * rpp + 2 * rp_i + r;
                    const bool ok = (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
                    gvo[t][r] = ok ? (unsigned)((gy * W + gx) * p.in_stride + (rem & 7) * 4) * 4u : OOB_OFFSET;
                }
            }
        }
    };
    // planes of depth transform sd = 2 sdh + sdl of the tile with first output plane d0:
    //   sd 0: x[d0-1] - x[d0+1]   sd 1: x[d0] + x[d0+1]   sd 2: x[d0+1] - x[d0]   sd 3: x[d0] - x[d0+2]
    struct Round { float4 P[4], Q[4]; };
    auto issue_round = [&](int t, int sdl, bool live, Round& L) {
        const int d0 = 2 * pdp;
        const int pP = sdl == 0 ? (sdh ? d0 + 1 : d0 - 1) : d0;
        const int pQ = sdl == 0 ? (sdh ? d0 : d0 + 1) : (sdh ? d0 + 2 : d0 + 1);
        const bool vP = live && (unsigned)pP < (unsigned)D, vQ = live && (unsigned)pQ < (unsigned)D;     // wave-uniform
        const __amdgpu_buffer_rsrc_t rP = vP ? rs_in : rs_null, rQ = vQ ? rs_in : rs_null;
        const int oP = vP ? pP * in_slice_bytes : 0, oQ = vQ ? pQ * in_slice_bytes : 0;
        if (ESTD_W2XABL & (2 | 128)) return;       // (128: the transforms and LDS writes without the plane loads)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (ESTD_W2XABL & 256) {             // (ablation: the same number of loads from 8 KB that stay in the L1)
                L.P[r] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rP, lane * 16 + r * 1024, 0, 0));
                if (!(ESTD_W2XABL & 512)) L.Q[r] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rQ, lane * 16 + r * 1024 + 4096, 0, 0));
                continue;
            }
            L.P[r] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rP, gvo[t][r], oP, 0));
            if (!(ESTD_W2XABL & 512)) L.Q[r] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rQ, gvo[t][r], oQ, 0));   // (ablation: half the plane loads)
        }
        if (ESTD_W2XABL & 512) {
#pragma unroll
            for (int r = 0; r < 4; ++r) L.Q[r] = L.P[(r + 1) & 3];
        }
    };
    auto finish_round = [&](int t, int sdl, const Round& L) {
        if (ESTD_W2XABL & 2) return;
        const float sg = (sdl == 1 && sdh == 0) ? 1.0f : -1.0f;      // wave-uniform
        // packed arithmetic (v_pk_fma_f32 / v_pk_add_f32: two operands per VALU slot -- beside the 64-cycle MFMA a VALU instruction costs
        // 5.5 .. 7 matrix cycles packed or not, profiles/r5_mfma32_filler_cost.txt)
        f32x2 X[4][2];
        const f32x2 sg2 = {sg, sg};
        f32x2 Tq[4][2];
        if (ESTD_W2X_PK) {
            // inline assembly: the compiler's pre-emit peephole splits every packed fp32 operation that follows an MFMA into two plain ones
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const f32x2 q0 = {L.Q[r].x, L.Q[r].y}, q1 = {L.Q[r].z, L.Q[r].w}, p0 = {L.P[r].x, L.P[r].y}, p1 = {L.P[r].z, L.P[r].w};
                asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(X[r][0]) : "v"(q0), "v"(sg2), "v"(p0));
                asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(X[r][1]) : "v"(q1), "v"(sg2), "v"(p1));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(Tq[0][h]) : "v"(X[0][h]), "v"(X[2][h]));
                asm("v_pk_add_f32 %0, %1, %2" : "=v"(Tq[1][h]) : "v"(X[1][h]), "v"(X[2][h]));
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(Tq[2][h]) : "v"(X[2][h]), "v"(X[1][h]));
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(Tq[3][h]) : "v"(X[1][h]), "v"(X[3][h]));
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                X[r][0] = __builtin_elementwise_fma((f32x2){L.Q[r].x, L.Q[r].y}, sg2, (f32x2){L.P[r].x, L.P[r].y});
                X[r][1] = __builtin_elementwise_fma((f32x2){L.Q[r].z, L.Q[r].w}, sg2, (f32x2){L.P[r].z, L.P[r].w});
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Tq[0][h] = X[0][h] - X[2][h];
                Tq[1][h] = X[1][h] + X[2][h];
                Tq[2][h] = X[2][h] - X[1][h];
                Tq[3][h] = X[1][h] - X[3][h];
            }
        }
        const float4 T0 = make_float4(Tq[0][0][0], Tq[0][0][1], Tq[0][1][0], Tq[0][1][1]);
        const float4 T1 = make_float4(Tq[1][0][0], Tq[1][0][1], Tq[1][1][0], Tq[1][1][1]);
        const float4 T2 = make_float4(Tq[2][0][0], Tq[2][0][1], Tq[2][1][0], Tq[2][1][1]);
        const float4 T3 = make_float4(Tq[3][0][0], Tq[3][0][1], Tq[3][1][0], Tq[3][1][1]);
        char* dst = smem + wr[t] + sdl * HALF_B;
        if (ESTD_W2XABL & 64) {                  // (ablation: the arithmetic without the LDS writes)
            asm volatile("" :: "v"(T0.x), "v"(T0.y), "v"(T0.z), "v"(T0.w), "v"(T1.x), "v"(T1.y), "v"(T1.z), "v"(T1.w));
            asm volatile("" :: "v"(T2.x), "v"(T2.y), "v"(T2.z), "v"(T2.w), "v"(T3.x), "v"(T3.y), "v"(T3.z), "v"(T3.w));
            return;
        }
        *reinterpret_cast<float4*>(dst + 0 * ROW_B) = T0;
        *reinterpret_cast<float4*>(dst + 1 * ROW_B) = T1;
        *reinterpret_cast<float4*>(dst + 2 * ROW_B) = T2;
        *reinterpret_cast<float4*>(dst + 3 * ROW_B) = T3;
    };

    // operands of one HALF step (sd, kw, chunk c, q): 16 MFMAs = 4 sh x 4 k-pairs (the four floats of a 16-byte piece)
    struct Half { float4 v[4]; };                // [sh]
    auto load_a = [&](int hk, Half& A) {         // weights of half step hk = 2 step + q of the tile (sd = 2 sdh + step / 6, kw = (step % 6) / 2, c = step & 1)
        if (ESTD_W2XABL & 4) return;
#pragma unroll
        for (int sh = 0; sh < 4; ++sh)
            A.v[sh] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane + sh * 1024, w_wave + hk * HALF_W_BYTES, 0));
    };
    auto load_b = [&](int hk, Half& B) {
        if (ESTD_W2XABL & 8) return;
        const int step = hk >> 1, q = hk & 1;
        const int sdl = step / 6, kw = (step % 6) >> 1, c = step & 1;
#pragma unroll
        for (int sh = 0; sh < 4; ++sh) B.v[sh] = *reinterpret_cast<const float4*>(smem + rd[kw][c][q] + sdl * HALF_B + sh * ROW_B);
    };

    // ---- prologue: the first tile's depth transform sdl = 0, unoverlapped; round 0 of its sdl = 1 requested ----
    Round RD[2];
    Half A[3], B[2];                              // weights two half steps ahead (ring of three), fragments one half step ahead
    if (ESTD_W2XABL & 12) {
#pragma unroll
        for (int sh = 0; sh < 4; ++sh) {
            A[0].v[sh] = A[1].v[sh] = A[2].v[sh] = make_float4(0.5f, -0.25f, 0.125f, 1.0f);
            B[0].v[sh] = B[1].v[sh] = make_float4(lane * 0.01f, 1.0f, -1.0f, 0.5f);
        }
    }
    if (ESTD_W2XABL & (2 | 128)) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) RD[k].P[r] = RD[k].Q[r] = make_float4(lane * 0.5f, r * 1.0f, k * 2.0f, 1.0f);
    }
    set_prod(u);
    load_a(0, A[0]);
    load_a(1, A[1]);
#pragma unroll
    for (int t = 0; t < ROUNDS; ++t) {
        issue_round(t, 0, true, RD[0]);
        finish_round(t, 0, RD[0]);
    }
    issue_round(0, 1, true, RD[0]);
    load_b(0, B[0]);
    __syncthreads();                              // lds_ss visible

    // L2 touch (ESTD_W2X_TOUCH): the vector memory unit returns loads in order, so a plane load that misses the L2 holds back every weight
    // load requested after it, and a production round has one step (2 048 cycles) of cover -- less than an HBM round trip under load
    // (measured: production cost 18 % of the kernel).  One dword per 128-byte voxel record of the two planes that are new to the tile
    // AFTER the next one is requested behind the tap loop (the epilogue covers its latency); the operand loads then hit the L2.
    float tch[2] = {0.f, 0.f};
    int tile_parity = 0;
    for (; u < u_end; ++u, tile_parity ^= 1) {
        // consumption cursor = the production cursor's tile (it moves on at step 5)
        const int cn = pn, cthi = pthi, ctwi = ptwi, d0 = 2 * pdp;
        const int th0 = cthi * TH, tw0 = ctwi * TW;
        const bool has_next = u + 1 < u_end;
        unsigned eoff[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int y = th0 + 4 * rpp + 2 * rpl + m, x = tw0 + pi;
            eoff[m] = (y < H && x < W) ? (unsigned)((y * W + x) * p.out_stride + 4 * k2) * 4u : OOB_OFFSET;
        }

        f32x16 acc[2][4];
#pragma clang loop unroll(full)
        for (int hk = 0; hk < 24; ++hk) {
            const int step = hk >> 1, q = hk & 1, sdl = step / 6;
            // requests first: the weights of half step hk + 2 and the fragments of half step hk + 1 (beyond 23: of the next tile -- the
            // weights do not depend on the tile, and block sdl = 0 of the next tile is complete since step 10; without a next tile the
            // fragment reads return stale LDS bytes nobody uses), then the next production round's planes
            load_a((hk + 2) % 24, A[(hk + 2) % 3]);
            load_b((hk + 1) % 24, B[(hk + 1) & 1]);
            if (q == 0) {
                if (step == 5) {
                    if (has_next) set_prod(u + 1);
                    issue_round(0, 0, has_next, RD[1]);                  // phase 1 (steps 6..10): depth transform sdl = 0 of the NEXT tile
                } else if (step == 11) {
                    issue_round(0, 1, has_next, RD[0]);                  // phase 0 of the next tile: its depth transform sdl = 1
                } else {
                    const int ph = step / 6, t = step % 6;               // round t of phase ph is finished in this step, round t + 1 requested
                    if (t + 1 < ROUNDS) issue_round(t + 1, ph == 0 ? 1 : 0, ph == 0 || has_next, RD[(5 * ph + t + 1) & 1]);
                }
            }
            if (ESTD_W2X_SCHED == 0) __builtin_amdgcn_sched_barrier(0);
            if (ESTD_W2X_TOUCH && hk == 12) asm volatile("" :: "v"(tch[0]), "v"(tch[1]));      // (the touch loads of the previous tile end here)
            if (q == 0 && step != 5 && step != 11) {
                const int ph = step / 6, t = step % 6;
                finish_round(t, ph == 0 ? 1 : 0, RD[(5 * ph + t) & 1]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int sh = 0; sh < 4; ++sh) {
                    const float4 a4 = A[hk % 3].v[sh], b4 = B[hk & 1].v[sh];
                    const float a = e == 0 ? a4.x : e == 1 ? a4.y : e == 2 ? a4.z : a4.w;
                    const float b = e == 0 ? b4.x : e == 1 ? b4.y : e == 2 ? b4.z : b4.w;
                    const bool first_product = hk % 12 == 0 && e == 0;
                    f32x16 c_in;
                    if (first_product) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) c_in[k] = 0.0f;
                    } else c_in = acc[sdl][sh];
                    acc[sdl][sh] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c_in, 0, 0, 0);
                }
            // issue order of the half step: the memory instructions and the transform arithmetic one group per MFMA gap (an MFMA occupies the
            // matrix pipe for 64 cycles: a cluster of 16 requests in front of the MFMAs drains it -- measured with the requests clustered:
            // 6 400 cycles per wave and tile for the operand production, of which 1 750 are its VALU instructions)
            if (ESTD_W2X_SCHED == 1) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // one vector memory read
                    __builtin_amdgcn_sched_group_barrier(0x080, 1, 0);      // one LDS instruction
                    __builtin_amdgcn_sched_group_barrier(0x002, ESTD_W2X_VPG, 0);   // transform arithmetic
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- row half of the output transform: z[sdl][m] ----
        f32x16 z[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            z[s][0] = acc[s][0] + acc[s][1] + acc[s][2];
            z[s][1] = acc[s][1] - acc[s][2] - acc[s][3];
        }
        // ---- depth half across the wave pair: wave sdh = 0 sends z1 (its z[1]) and receives z2, wave sdh = 1 sends z2 (its z[0]) and
        // receives z1; buffer = this wave's block sdl = 1 (read for the last time in step 11, rewritten from step 0 of the next tile) ----
        float4* xs = reinterpret_cast<float4*>(smem + wbase + HALF_B) + lane;
        const float4* xr = reinterpret_cast<const float4*>(smem + (wave ^ 2) * WAVE_B + HALF_B) + lane;
        if (!(ESTD_W2XABL & 16)) {
            if (sdh == 0) {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) xs[(m * 4 + qq) * 64] = make_float4(z[1][m][4 * qq], z[1][m][4 * qq + 1], z[1][m][4 * qq + 2], z[1][m][4 * qq + 3]);
            } else {
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) xs[(m * 4 + qq) * 64] = make_float4(z[0][m][4 * qq], z[0][m][4 * qq + 1], z[0][m][4 * qq + 2], z[0][m][4 * qq + 3]);
            }
            lds_barrier();
        }
        // y = z0 + z1 + z2 (plane d0, wave sdh = 0: own z[0] + z[1] + received) | y = z1 - z2 - z3 (plane d0 + 1: received - own z[0] - z[1])
        const float ysg = sdh ? -1.0f : 1.0f;
        f32x16 y[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const f32x16 s = z[0][m] + z[1][m];
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (!(ESTD_W2XABL & 16)) o = xr[(m * 4 + qq) * 64];
                y[m][4 * qq + 0] = fmaf(s[4 * qq + 0], ysg, o.x);
                y[m][4 * qq + 1] = fmaf(s[4 * qq + 1], ysg, o.y);
                y[m][4 * qq + 2] = fmaf(s[4 * qq + 2], ysg, o.z);
                y[m][4 * qq + 3] = fmaf(s[4 * qq + 3], ysg, o.w);
            }
        }

        // ---- epilogue of plane d0 + sdh: rows 4 rpp + 2 rpl + m, column pi, channels 8 qq + 4 k2 .. + 3 ----
        const int dd = d0 + sdh;
        const bool plane_ok = dd < D;                                   // (odd D: the last pair has one plane)
        const __amdgpu_buffer_rsrc_t rs_out = plane_ok ? make_rsrc(p.out_main + (size_t)cn * vol * p.out_stride, vol * p.out_stride) : rs_null;
        const int so = plane_ok ? dd * out_plane_bytes : 0;
        // read-back streams: decided by the INSTANCE (no runtime flag inside: a conditionally loaded register array becomes a loop-carried
        // value with a select per element); an absent second residual reads through the null descriptor (zeros, no memory access)
        float4 r1[2][4], r2[2][4], ro[2][4];
        if (RB_RES) {
            const __amdgpu_buffer_rsrc_t rs_res = (plane_ok && p.residual) ? make_rsrc(p.residual + (size_t)cn * vol * p.out_stride, vol * p.out_stride) : rs_null;
            const __amdgpu_buffer_rsrc_t rs_res2 = (plane_ok && p.residual2) ? make_rsrc(p.residual2 + (size_t)cn * vol * p.out_stride, vol * p.out_stride) : rs_null;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) {
                    r1[m][qq] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_res, eoff[m], so + 32 * qq, 0));
                    r2[m][qq] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_res2, eoff[m], so + 32 * qq, 0));
                }
        }
        if (RB_ACC) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int qq = 0; qq < 4; ++qq) ro[m][qq] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_out, eoff[m], so + 32 * qq, 0));
        }
        if (ESTD_W2X_TOUCH && has_next) {
            // production cursor = the next tile (first output plane 2 pdp): the tile after it brings planes 2 pdp + 3 and 2 pdp + 4
            const int tp = 2 * pdp + 3;
            const __amdgpu_buffer_rsrc_t rt = tp < D ? rs_in : rs_null;
            const int to = tp < D ? tp * in_slice_bytes : 0;
#pragma unroll
            for (int j = 0; j < 2; ++j) tch[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rt, tvo[j], to, 0));
        }
        if (STATS) {
            // GroupNorm(1 group per 16 channels) partial sums of the raw (BN-folded, pre-activation) outputs of this wave's plane: group =
            // qq >> 1.  Fixed-order reduction: lanes of a DPP row, the four rows, then the two row-pair pairs through LDS.
            double v[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int cb = 8 * qq + 4 * k2;
                const float4 sc4 = *reinterpret_cast<const float4*>(lds_ss + cb), sh4 = *reinterpret_cast<const float4*>(lds_ss + 32 + cb);
                const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    if (eoff[m] != OOB_OFFSET) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const double uu = (double)(y[m][4 * qq + e] * scv[e] + shv[e]);
                            v[qq >> 1][0] += uu;
                            v[qq >> 1][1] += uu * uu;
                        }
                    }
            }
#pragma unroll
            for (int gq = 0; gq < 2; ++gq)
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    double t = row16_sum_f64(v[gq][k]);
                    t += __shfl_xor(t, 16);
                    t += __shfl_xor(t, 32);
                    v[gq][k] = t;
                }
            double* red = lds_red + tile_parity * 16;                   // [plane sdh][rpp][group][sum, sumsq]
            if (lane == 0) {
#pragma unroll
                for (int gq = 0; gq < 2; ++gq)
#pragma unroll
                    for (int k = 0; k < 2; ++k) red[((sdh * 2 + rpp) * 2 + gq) * 2 + k] = v[gq][k];
            }
        }
        // second barrier of the tile: every wave has read its partner's exchange buffer (the partner rewrites that block from step 0 of
        // the next tile on); STATS: the partial sums of the four waves are visible
        if (!(ESTD_W2XABL & 16)) lds_barrier();
        if (STATS && tid < 8) {                                          // (plane, group, {sum, sumsq})
            const int pl_ = tid >> 2, gq = (tid >> 1) & 1, k = tid & 1;
            if (d0 + pl_ < D) {
                const double* red = lds_red + tile_parity * 16;
                const double tot = red[((pl_ * 2 + 0) * 2 + gq) * 2 + k] + red[((pl_ * 2 + 1) * 2 + gq) * 2 + k];
                const size_t tile_id = (((size_t)cn * D + d0 + pl_) * tiles_h + cthi) * tiles_w + ctwi;      // canonical tile id, as the direct kernel writes it
                p.stats_partials[tile_id * 4 + gq * 2 + k] = tot;
            }
        }
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int cb = 8 * qq + 4 * k2;
                float4 v = make_float4(y[m][4 * qq], y[m][4 * qq + 1], y[m][4 * qq + 2], y[m][4 * qq + 3]);
                if (!(ESTD_W2XABL & 32)) {
                    const float4 sc4 = *reinterpret_cast<const float4*>(lds_ss + cb), sh4 = *reinterpret_cast<const float4*>(lds_ss + 32 + cb);
                    const float4 lo = *reinterpret_cast<const float4*>(lds_ss + 64 + cb);
                    v.x = fmaxf(fmaf(v.x, sc4.x, sh4.x), lo.x);
                    v.y = fmaxf(fmaf(v.y, sc4.y, sh4.y), lo.y);
                    v.z = fmaxf(fmaf(v.z, sc4.z, sh4.z), lo.z);
                    v.w = fmaxf(fmaf(v.w, sc4.w, sh4.w), lo.w);
                    if (RB_RES) {
                        v.x = (v.x + r1[m][qq].x + r2[m][qq].x) * p.out_scale;
                        v.y = (v.y + r1[m][qq].y + r2[m][qq].y) * p.out_scale;
                        v.z = (v.z + r1[m][qq].z + r2[m][qq].z) * p.out_scale;
                        v.w = (v.w + r1[m][qq].w + r2[m][qq].w) * p.out_scale;
                    }
                    if (RB_ACC) { v.x += ro[m][qq].x; v.y += ro[m][qq].y; v.z += ro[m][qq].z; v.w += ro[m][qq].w; }
                }
                u32x4 bits;
                __builtin_memcpy(&bits, &v, 16);
                if (!(ESTD_W2XABL & 1)) __builtin_amdgcn_raw_buffer_store_b128(bits, rs_out, eoff[m], so + 32 * qq, 0);
            }
    }
}

}  // namespace

// 32 -> 32 instance of estd_conv3d_k3_wino2 on the operand-reuse kernel; desc->w_wino2 in packing.pack_conv3d_wino2x form.
// ESTD_ERR_UNSUPPORTED for everything else (the caller falls back to csrc/conv3d_wino2.hip).
extern "C" int estd_conv3d_k3_wino2x(const estd_conv3d_desc* dp, estd_stream_t s)
{
    if (!dp) return ESTD_ERR_ARG;
    const estd_conv3d_desc& d = *dp;
    if (d.N <= 0 || d.D <= 0 || d.H <= 0 || d.W <= 0) return ESTD_ERR_ARG;
    if (!d.in_main || !d.w_wino2 || !d.scale || !d.shift || !d.out_main) return ESTD_ERR_ARG;
    if (d.cin_main != 32 || d.n_tiles != 2 || d.out_head || d.in_extra || d.w_extra || d.out_extra) return ESTD_ERR_UNSUPPORTED;
    if (d.act_a == ESTD_ACT_TANH || d.act_b == ESTD_ACT_TANH) return ESTD_ERR_UNSUPPORTED;
    if (d.in_stride < 32 || (d.in_stride & 3) || d.out_stride < 32 || (d.out_stride & 3)) return ESTD_ERR_ARG;
    const int tiles_w = (d.W + TW - 1) / TW, tiles_h = (d.H + TH - 1) / TH, dpairs = (d.D + 1) / 2;
    const long long total = (long long)d.N * dpairs * tiles_h * tiles_w;
    if (total > 0x7fffffffLL) return ESTD_ERR_ARG;
    {
        const long long vox = (long long)d.D * d.H * d.W;
        const int widest = d.in_stride > d.out_stride ? d.in_stride : d.out_stride;
        if (vox * widest * 4 >= 0x7fffff00LL) return ESTD_ERR_UNSUPPORTED;
    }
    const int slots = estd_persistent_wgs(1);
    int grid = total < slots ? (int)total : slots;
    if (grid >= 8) grid &= ~7;
    const bool res_any = d.residual || d.residual2 || d.out_scale != 1.0f;
    const int rbk = (!res_any && !d.accumulate) ? 0 : (!res_any ? 1 : (!d.accumulate ? 2 : 3));
#define ESTD_W2X_LAUNCH(RBV, STV)                                                                                               \
    do {                                                                                                                        \
        estd_allow_dynamic_lds<conv3d_wino2x_kernel<RBV, STV>>(LDS_BYTES);                                                      \
        hipLaunchKernelGGL((conv3d_wino2x_kernel<RBV, STV>), dim3(grid), dim3(256), LDS_BYTES, estd_stream(s), d, tiles_w,      \
                           tiles_h, dpairs, (int)total);                                                                        \
    } while (0)
    if (d.stats_partials) {
        if (rbk != 0) return ESTD_ERR_UNSUPPORTED;
        ESTD_W2X_LAUNCH(0, true);
    } else {
        switch (rbk) {
        case 0: ESTD_W2X_LAUNCH(0, false); break;
        case 1: ESTD_W2X_LAUNCH(1, false); break;
        case 2: ESTD_W2X_LAUNCH(2, false); break;
        default: ESTD_W2X_LAUNCH(3, false); break;
        }
    }
#undef ESTD_W2X_LAUNCH
    return ESTD_LAUNCH_CHECK();
}
