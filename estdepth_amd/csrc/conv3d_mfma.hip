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

#include "estd_hip.h"
#include "estd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TH = 8;    // tile rows
constexpr int TW = 16;   // tile columns (= one MFMA M tile)
constexpr int IN_D = 3, IN_H = TH + 2, IN_W = TW + 2;
constexpr int NVOX_IN = IN_D * IN_H * IN_W;   // 540
constexpr int MT = 2;    // M tiles (rows) per wave: 4 waves x 2 = 8 rows

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

template <int CM, int NT, bool EXTRA>
__global__ __launch_bounds__(256, 2) void conv3d_k3_kernel(const estd_conv3d_desc p, int tiles_w, int tiles_h)
{
    constexpr int CH = CM / 4;          // 16-byte chunks per voxel
    constexpr int KS = CM / 4;          // MFMA k-steps per tap
    constexpr int QN = (KS * NT) / 4;   // float4 weight quads per lane per tap
    constexpr int XS = 7;               // k-steps of the extra (scalar) input channel: 27 taps padded to 28
    constexpr int XQ = (XS * NT + 3) / 4;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_main = smem;
    float* lds_extra = reinterpret_cast<float*>(smem + NVOX_IN * CM * 4);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4;            // k index inside an MFMA
    const int i = lane & 15;            // M row (A) / N column (B, D)

    // ---- XCD-aware tile id (bijective) ----
    const int nwg = gridDim.x;
    int tile;
    {
        const int bid = blockIdx.x;
        const int q = nwg >> 3, r = nwg & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int t = tile;
    const int twi = t % tiles_w; t /= tiles_w;
    const int thi = t % tiles_h; t /= tiles_h;
    const int d0 = t % p.D;
    const int n = t / p.D;
    const int tw0 = twi * TW, th0 = thi * TH;
    const int D = p.D, H = p.H, W = p.W;

    // ---- stage the input brick in LDS (zero padded) ----
    {
        const float* __restrict__ in = p.in_main + (size_t)n * D * H * W * p.in_stride;
        constexpr int NEL = NVOX_IN * CH;
        constexpr int ITER = (NEL + 255) / 256;
        float4 vals[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int e = tid + it * 256;
            const int v = e / CH, c = e % CH;
            const int zx = v % IN_W;
            const int t2 = v / IN_W;
            const int zy = t2 % IN_H, zd = t2 / IN_H;
            const int gx = tw0 - 1 + zx, gy = th0 - 1 + zy, gd = d0 - 1 + zd;
            float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < NEL && gx >= 0 && gx < W && gy >= 0 && gy < H && gd >= 0 && gd < D)
                val = *reinterpret_cast<const float4*>(in + ((size_t)(gd * H + gy) * W + gx) * p.in_stride + c * 4);
            vals[it] = val;
        }
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int e = tid + it * 256;
            if (e < NEL) {
                const int v = e / CH, c = e % CH;
                *reinterpret_cast<float4*>(lds_main + lds_chunk_off<CM>(v, c)) = vals[it];
            }
        }
        if (EXTRA) {
            const float* __restrict__ ex = p.in_extra + (size_t)n * D * H * W;
            for (int v = tid; v < NVOX_IN; v += 256) {
                const int zx = v % IN_W;
                const int t2 = v / IN_W;
                const int zy = t2 % IN_H, zd = t2 / IN_H;
                const int gx = tw0 - 1 + zx, gy = th0 - 1 + zy, gd = d0 - 1 + zd;
                float val = 0.f;
                if (gx >= 0 && gx < W && gy >= 0 && gy < H && gd >= 0 && gd < D)
                    val = ex[(size_t)(gd * H + gy) * W + gx];
                lds_extra[v] = val;
            }
        }
    }
    __syncthreads();

    // ---- main loop: 27 taps x KS k-steps ----
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int nn = 0; nn < NT; ++nn) acc[m][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float4* __restrict__ wq = reinterpret_cast<const float4*>(p.w_main) + lane;
    float4 bcur[QN], bnext[QN];
#pragma unroll
    for (int q = 0; q < QN; ++q) bcur[q] = wq[q * 64];

    const int row0 = wave * MT;   // first tile row of this wave
    int tap = 0;
    for (int kd = 0; kd < 3; ++kd) {
        for (int kh = 0; kh < 3; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 3; ++kw, ++tap) {
                // prefetch next tap's weights (the packed buffer carries one padding tap)
#pragma unroll
                for (int q = 0; q < QN; ++q) bnext[q] = wq[((tap + 1) * QN + q) * 64];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int v = (kd * IN_H + (row0 + m + kh)) * IN_W + kw + i;
                    const int off0 = lds_chunk_off<CM>(v, g);
                    const float4 a0 = *reinterpret_cast<const float4*>(lds_main + off0);
                    float4 a1 = a0;
                    if (CM == 32) a1 = *reinterpret_cast<const float4*>(lds_main + (off0 ^ 64));
                    const float av[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
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
            }
        }
    }

    // ---- extra scalar input channel: its 27 taps form one more K chunk (28 = 7 x 4) ----
    if (EXTRA) {
        const float4* __restrict__ xq = reinterpret_cast<const float4*>(p.w_extra) + lane;
        float4 bx[XQ];
#pragma unroll
        for (int q = 0; q < XQ; ++q) bx[q] = xq[q * 64];
#pragma unroll
        for (int s = 0; s < XS; ++s) {
            int tp = 4 * s + g;
            tp = tp > 26 ? 26 : tp;           // tap 27 is padding (zero weight)
            const int kd = tp / 9, kh = (tp / 3) % 3, kw = tp % 3;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int v = (kd * IN_H + (row0 + m + kh)) * IN_W + kw + i;
                const float a = lds_extra[v];
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

    // ---- epilogue ----
    // D layout: lane holds column j = i (N index) and rows 4g..4g+3 (M index = voxel along W).
    // channel of (tile nn, column j): NT==1 -> j ; NT>=2 -> 2j+nn for nn<2 ; nn==2 -> 32 (only j==0).
    const int cbase = (NT == 1) ? i : 2 * i;
    float sc[2], sh[2];
    sc[0] = p.scale[cbase]; sh[0] = p.shift[cbase];
    sc[1] = sc[0]; sh[1] = sh[0];
    if (NT >= 2) { sc[1] = p.scale[cbase + 1]; sh[1] = p.shift[cbase + 1]; }
    const int act0 = cbase < p.act_split ? p.act_a : p.act_b;   // both channels of a lane share the range (split is even)
    float sc2 = 0.f, sh2 = 0.f;
    if (NT == 3) { sc2 = p.scale[32]; sh2 = p.shift[32]; }
    float hw = 0.f, hb = 0.f;
    if (NT == 1 && p.head_w) { hw = p.head_w[i]; hb = p.head_b[0]; }

    double s_sum = 0.0, s_sq = 0.0;     // GroupNorm partials of this lane (its channels are in one group)
    const size_t vol_base = (size_t)n * D * H * W;

#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int y = th0 + row0 + m;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int x = tw0 + 4 * g + r;
            const bool valid = (y < H) && (x < W);
            const size_t vox = vol_base + ((size_t)d0 * H + y) * W + x;
            float v0 = acc[m][0][r] * sc[0] + sh[0];
            float v1 = 0.f;
            if (NT >= 2) v1 = acc[m][1][r] * sc[1] + sh[1];
            if (p.stats_partials && valid) {
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
            if (p.out_main && valid) {
                float* o = p.out_main + vox * p.out_stride + cbase;
                if (NT == 1) {
                    if (p.residual) v0 += p.residual[vox * p.out_stride + cbase];
                    v0 *= p.out_scale;
                    if (p.accumulate) v0 += *o;
                    *o = v0;
                } else {
                    if (p.residual) {
                        const float2 rr = *reinterpret_cast<const float2*>(p.residual + vox * p.out_stride + cbase);
                        v0 += rr.x; v1 += rr.y;
                    }
                    v0 *= p.out_scale; v1 *= p.out_scale;
                    if (p.accumulate) {
                        const float2 pr = *reinterpret_cast<const float2*>(o);
                        v0 += pr.x; v1 += pr.y;
                    }
                    *reinterpret_cast<float2*>(o) = make_float2(v0, v1);
                }
            }
            if (NT == 3) {
                if (valid && i == 0 && p.out_extra) {
                    float v2 = acc[m][2][r] * sc2 + sh2;
                    p.out_extra[vox] = act_apply(v2, p.act_b);
                }
            }
        }
    }

    if (p.stats_partials) {
        // group 0 = channels 0..15, group 1 = channels 16..31.  Lane's channels: cbase(,+1).
        const int grp = (cbase >= 16) ? 1 : 0;
        double a0 = grp == 0 ? s_sum : 0.0, q0 = grp == 0 ? s_sq : 0.0;
        double a1 = grp == 1 ? s_sum : 0.0, q1 = grp == 1 ? s_sq : 0.0;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            a0 += __shfl_xor(a0, o); q0 += __shfl_xor(q0, o);
            a1 += __shfl_xor(a1, o); q1 += __shfl_xor(q1, o);
        }
        __syncthreads();   // LDS brick no longer needed: reuse it for the cross-wave reduction
        double* red = reinterpret_cast<double*>(smem);
        if (lane == 0) { red[wave * 4 + 0] = a0; red[wave * 4 + 1] = q0; red[wave * 4 + 2] = a1; red[wave * 4 + 3] = q1; }
        __syncthreads();
        if (tid < 4) {
            const double tot = red[tid] + red[4 + tid] + red[8 + tid] + red[12 + tid];
            p.stats_partials[(size_t)tile * 4 + tid] = tot;
        }
    }
}

template <int CM, int NT, bool EXTRA>
int launch(const estd_conv3d_desc& d, hipStream_t stream)
{
    const int tiles_w = (d.W + TW - 1) / TW, tiles_h = (d.H + TH - 1) / TH;
    const int grid = d.N * d.D * tiles_h * tiles_w;
    const size_t lds = (size_t)NVOX_IN * CM * 4 + (EXTRA ? NVOX_IN * 4 : 0);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv3d_k3_kernel<CM, NT, EXTRA>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3d_k3_kernel<CM, NT, EXTRA>), dim3(grid), dim3(256), lds, stream, d, tiles_w, tiles_h);
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
    if (d.n_tiles == 3 && !d.out_extra) return ESTD_ERR_ARG;
    if ((long long)d.N * d.D * ((d.H + TH - 1) / TH) * ((d.W + TW - 1) / TW) > 0x7fffffffLL) return ESTD_ERR_ARG;

    if (d.cin_main == 32 && d.n_tiles == 2 && !extra) return launch<32, 2, false>(d, stream);
    if (d.cin_main == 32 && d.n_tiles == 2 && extra)  return launch<32, 2, true>(d, stream);
    if (d.cin_main == 32 && d.n_tiles == 3 && extra)  return launch<32, 3, true>(d, stream);
    if (d.cin_main == 32 && d.n_tiles == 1 && !extra) return launch<32, 1, false>(d, stream);
    if (d.cin_main == 16 && d.n_tiles == 1 && !extra) return launch<16, 1, false>(d, stream);
    return ESTD_ERR_UNSUPPORTED;
}
