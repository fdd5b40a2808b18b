// conv1x1.hip -- 1x1 convolution (stride 1 | 2) of an NHWC map with the folded BatchNorm, the residual add and the ReLU in its
// epilogue, on gfx950 fp32 MFMA: the bottleneck convolutions of the semantic branch's ResNet (hybrid_models/resnet_encoder.py:40-51
// over torchvision's Bottleneck: conv1 / conv3 / downsample[0], each followed by BatchNorm2d, conv3's sum with the shortcut and
// the block's ReLU).  SURVEY.md §8(f) rank 3.  One launch replaces a library GEMM + a separate BatchNorm / add / ReLU pass over the
// output map.
//
// A GEMM D[cout][pixel] = W[cout][cin] . X[pixel][cin]^T on v_mfma_f32_16x16x4_f32, TRANSPOSED like the other kernels of this
// library (weights as the A operand, pixels as B): lane (g, i) of an accumulator holds output channels n0 + 4g .. 4g+3 of pixel
// m0 + i -- the epilogue is one 16-byte residual load, four FMAs / maxes and one 16-byte store per accumulator.
//   * operands come STRAIGHT from L1 / L2 in MFMA layout, no LDS and no barrier: the k index of lane group g at k-step e of a
//     16-channel chunk is channel 4g + e on both sides, so a lane's operand quad is the 16 bytes at X[pixel][k0 + 4g] (NHWC rows
//     are contiguous in the channel) resp. W[cout][k0 + 4g] (the Conv2d weight exactly as it lies in memory: nothing to pack);
//   * a wave owns a (16 TM pixels) x (16 TN channels) block: per 16-channel chunk TM + TN 16-byte loads feed 4 TM TN MFMAs
//     (TM = TN = 4: 64 MFMAs per 8 loads); the next chunk's quads are requested before the current chunk's MFMAs;
//   * the four waves of a workgroup take neighbouring channel blocks of the same pixels (their pixel loads meet in L1); maps with
//     few pixels (layer3 / layer4 at 30x40 / 15x20) take 32 x 32 blocks so that there are enough waves to go round;
//   * stride 2 (the downsample convolutions): output pixel (y, x) reads input pixel (2y, 2x) -- an address computation, no gather pass.
// The layers with few input channels at full resolution (64 -> 256 at 120x160) are HBM-bound: 74 MB per launch of 3 images.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "estd_hip.h"
#include "estd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;

__device__ __forceinline__ float4 as_float4(u32x4 v) { float4 f; __builtin_memcpy(&f, &v, 16); return f; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}

// SK = 1: the four waves of a workgroup own four neighbouring blocks.  SK = 4 (K-heavy layers on small maps: few blocks, long K): the
// four waves of a workgroup share ONE block and split its input channels into four contiguous ranges; waves 1..3 hand their partial
// sums to wave 0 through LDS, which adds them in a fixed order (deterministic) and runs the epilogue.  Larger blocks per wave = fewer
// operand bytes per MFMA from L2, which is what bounds this kernel.
template <int TM, int TN, int PF, int SK>
__global__ __launch_bounds__(256) void conv1x1_nhwc_kernel(const estd_conv1x1_desc p, int Ho, int Wo, int tiles_m, int tiles_n)
{
    __shared__ float4 red[SK > 1 ? (SK - 1) * TN * TM * 64 : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int wt = SK > 1 ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;      // wave tile: channel block fastest
    if (wt >= tiles_m * tiles_n) return;                        // (workgroup-uniform when SK > 1)
    const int tn = wt % tiles_n, tm = wt / tiles_n;
    const int m0 = tm * 16 * TM, n0 = tn * 16 * TN;
    const int Mtot = p.N * Ho * Wo;
    const int cin = p.cin, cout = p.cout;

    const __amdgpu_buffer_rsrc_t rs_x = make_rsrc(p.in, (size_t)p.N * p.H * p.W * cin * 4);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w, (size_t)cout * cin * 4);
    unsigned xoff[TM], woff[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) {
        const int m = m0 + 16 * t + i;
        const int x = m % Wo, r = m / Wo;
        const int y = r % Ho, n = r / Ho;
        xoff[t] = m < Mtot ? (unsigned)((((size_t)n * p.H + y * p.stride) * p.W + x * p.stride) * cin + 4 * g) * 4u : OOB_OFFSET;
    }
#pragma unroll
    for (int t = 0; t < TN; ++t) woff[t] = (unsigned)((n0 + 16 * t + i) * cin + 4 * g) * 4u;       // cout is a multiple of 16 TN

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ring of PF chunks: PF - 1 chunks of operand quads are in flight while one is multiplied (small blocks have few MFMAs per chunk
    // and few waves per SIMD: they need several L2 round trips of cover; the registers are there)
    float4 xq[PF][TM], wq[PF][TN];
    auto load_chunk = [&](int k0, float4 (&xd)[TM], float4 (&wd)[TN]) {
#pragma unroll
        for (int t = 0; t < TN; ++t) wd[t] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_w, woff[t], k0 * 4, 0));
#pragma unroll
        for (int t = 0; t < TM; ++t) xd[t] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff[t], k0 * 4, 0));
    };
    auto mfma_chunk = [&](const float4 (&xs)[TM], const float4 (&ws)[TN]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const float wv = e == 0 ? ws[a].x : e == 1 ? ws[a].y : e == 2 ? ws[a].z : ws[a].w;
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const float xv = e == 0 ? xs[b].x : e == 1 ? xs[b].y : e == 2 ? xs[b].z : xs[b].w;
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, acc[a][b], 0, 0, 0);
                }
            }
    };
    const int nchunks = (cin >> 4) / SK;                        // this wave's share of the input channels: chunks kbase .. kbase + nchunks
    const int kbase = SK > 1 ? wave * nchunks : 0;
#pragma unroll
    for (int j = 0; j < PF - 1; ++j)
        if (j < nchunks) load_chunk((kbase + j) * 16, xq[j], wq[j]);
    for (int c = 0; c < nchunks; c += PF) {                     // PF chunks per trip: the ring slots are compile-time constants
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int cc = c + j;                                // (wave-uniform conditions)
            if (cc + PF - 1 < nchunks) load_chunk((kbase + cc + PF - 1) * 16, xq[(j + PF - 1) % PF], wq[(j + PF - 1) % PF]);
            if (cc < nchunks) mfma_chunk(xq[j], wq[j]);
        }
    }
    if (SK > 1) {
        if (wave > 0) {
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b)
                    red[(((wave - 1) * TN + a) * TM + b) * 64 + lane] = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < SK - 1; ++w)
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const float4 r = red[((w * TN + a) * TM + b) * 64 + lane];
                    acc[a][b] += (f32x4){r.x, r.y, r.z, r.w};
                }
    }

    // ---- epilogue: folded BatchNorm, + residual, ReLU; 16-byte stores ----
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, (size_t)Mtot * cout * 4);
    const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.residual ? p.residual : p.out, (size_t)Mtot * cout * 4);
    const float floor_ = p.relu ? 0.0f : ESTD_NO_FLOOR;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int cb = n0 + 16 * a + 4 * g;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale) sc = *reinterpret_cast<const float4*>(p.scale + cb);
        if (p.shift) sh = *reinterpret_cast<const float4*>(p.shift + cb);
        unsigned ooff[TM];
        float4 res[TM];
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const int m = m0 + 16 * b + i;
            ooff[b] = m < Mtot ? (unsigned)((size_t)m * cout + cb) * 4u : OOB_OFFSET;
        }
        if (p.residual) {
#pragma unroll
            for (int b = 0; b < TM; ++b) res[b] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_r, ooff[b], 0, 0));
        }
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            float4 v;
            v.x = fmaf(acc[a][b][0], sc.x, sh.x); v.y = fmaf(acc[a][b][1], sc.y, sh.y);
            v.z = fmaf(acc[a][b][2], sc.z, sh.z); v.w = fmaf(acc[a][b][3], sc.w, sh.w);
            if (p.residual) { v.x += res[b].x; v.y += res[b].y; v.z += res[b].z; v.w += res[b].w; }
            v.x = fmaxf(v.x, floor_); v.y = fmaxf(v.y, floor_); v.z = fmaxf(v.z, floor_); v.w = fmaxf(v.w, floor_);
            u32x4 bits;
            __builtin_memcpy(&bits, &v, 16);
            __builtin_amdgcn_raw_buffer_store_b128(bits, rs_o, ooff[b], 0, 0);
        }
    }
}

#ifndef ESTD_C1X1_COUNTED
#define ESTD_C1X1_COUNTED 1     // 1: counted vmcnt + raw s_barrier per stage (stages stay in flight across the barrier); 0: __syncthreads() (A/B)
#endif
// ---------------------------------------------------------------------------------------------------------------------------------
// LDS-tiled form (round 6).  The direct form above streams every operand quad of every WAVE from L1 / L2: a 64 x 64 wave block reads
// 128 bytes per MFMA, the smaller blocks of the small maps 256-384, and the K-heavy layers lose to the library.  Here a WORKGROUP owns a
// BN (output channels) x BM (pixels) tile, its four waves WN x WM sub-tiles of it, and both operand panels are staged ONCE per workgroup:
//   * global -> LDS with global_load_lds_dwordx4 (no staging registers, no ds_write pass): one wave-instruction moves a 1 KiB block of
//     16 rows x 16 channels; a stage holds U sub-chunks of 16 channels of the BN weight rows and the BM pixel rows, NS stages rotate;
//   * the LDS destination of a wave-instruction is linear (base + 16 lane), so the bank swizzle is applied on the GLOBAL side: lane j of a
//     block fetches row j / 4, 16-byte k-slot (j % 4) ^ F[(j / 4) / 4] with F = {0, 3, 2, 1}; a fragment read (lane (g, i) wants row i,
//     k-slot g) is the 16 bytes at i * 64 + ((g ^ F[i / 4]) * 16): conflict-free in every one of ds_read_b128's four 16-lane groups
//     ({0-3, 12-15, 20-27}, ... -- MI355X_MICROARCH.md LDS table): per row residue i % 4 the four lanes of a group land on the slots
//     {F0, F1^1, F2^1, F3} (+ the group's constant) = {0, 2, 3, 1};
//   * as above, the k index a lane group multiplies at step e of a 16-channel chunk is channel 4 g + e on both sides (any bijection is
//     a valid dot product), so one ds_read_b128 per 16 x 16 operand tile and chunk feeds four MFMAs per partner tile;
//   * one __syncthreads() per stage: it publishes stage c (requested two iterations ago: a whole iteration of MFMAs to land) and frees
//     stage c + NS - 1 (read in iteration c - 1) for the request that follows it;
//   * workgroup -> tile mapping in XCD-sized runs (workgroup b runs on XCD b % 8): the tiles of one XCD are consecutive, so the channel
//     tiles that share a pixel panel meet in one L2.
// Pixel rows beyond the map (the last pixel tile) are clamped to the last pixel for the loads and never stored.
template <int BM, int BN, int WM, int WN, int U, int NS>
__global__ __launch_bounds__(256) void conv1x1_lds_kernel(const estd_conv1x1_desc p, int Ho, int Wo, int tiles_m, int tiles_n, int per_xcd)
{
    constexpr int TM = BM / (16 * WM), TN = BN / (16 * WN);
    constexpr int NBLK = (BM + BN) / 16;                  // 1 KiB blocks per 16-channel sub-chunk: weight rows first, then pixel rows
    constexpr int STAGE = NBLK * U * 1024;
    constexpr int LPW = NBLK * U / 4;                     // global_load_lds instructions per wave and stage
    static_assert(WM * WN == 4 && (NBLK * U) % 4 == 0 && TM >= 1 && TN >= 1, "tile shape");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int t = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (t >= tiles_m * tiles_n) return;                   // (workgroup-uniform)
    const int tn = t % tiles_n, tm = t / tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int Mtot = p.N * Ho * Wo;
    const int cin = p.cin, cout = p.cout;

    // ---- this lane's global sources: LPW blocks, the same rows in every stage (only the channel offset moves) ----
    const float* src[LPW];
    const int jr = lane >> 2, ps = lane & 3;              // row of the block, physical 16-byte slot
    const int ks = ps ^ ((4 - (jr >> 2)) & 3);            // k-slot that lands there (F = {0, 3, 2, 1})
#pragma unroll
    for (int j = 0; j < LPW; ++j) {
        const int q = wave * LPW + j;                     // block of the stage (wave-uniform): sub-chunk q / NBLK, block q % NBLK
        const int u = q / NBLK, blk = q % NBLK;
        if (blk < BN / 16) {
            int n = n0 + blk * 16 + jr;
            n = n < cout ? n : cout - 1;
            src[j] = p.w + (size_t)n * cin + u * 16 + ks * 4;
        } else {
            int m = m0 + (blk - BN / 16) * 16 + jr;
            m = m < Mtot ? m : Mtot - 1;
            const int x = m % Wo, r = m / Wo;
            const int y = r % Ho, n = r / Ho;
            src[j] = p.in + (((size_t)n * p.H + y * p.stride) * p.W + x * p.stride) * cin + u * 16 + ks * 4;
        }
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)lds;       // LDS byte address of the stage ring
    auto request = [&](int c) {                           // chunk c (16 U channels) -> stage c % NS
#if ESTD_C1X1_COUNTED
        // The request as inline assembly: behind the BUILTIN hipcc puts an s_waitcnt vmcnt(0) in front of the next ds_read of this loop (it cannot
        // see that the stage being written is not the stage being read: one extern LDS array), i.e. every iteration would wait for the requests it
        // has just issued -- no prefetch at all (what the first version of this kernel did: MfmaUtil 27-41 %).  The counted waits at the top of
        // the loop are the only synchronisation of these loads.  m0 = wave-uniform LDS destination, one wait state before its use.
        const unsigned dst = lds_base + (unsigned)((c % NS) * STAGE + wave * LPW * 1024);
#pragma unroll
        for (int j = 0; j < LPW; ++j) {
            const float* g_ = src[j] + (size_t)c * 16 * U;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"          // (m0 is a reserved register: nothing else in this kernel depends on it)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g_), "s"(dst + j * 1024) : "memory", "m0");
#pragma clang diagnostic pop
        }
#else
        unsigned char* base = lds + (c % NS) * STAGE + wave * LPW * 1024;
#pragma unroll
        for (int j = 0; j < LPW; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[j] + (size_t)c * 16 * U),
                                             (__attribute__((address_space(3))) void*)(base + j * 1024), 16, 0, 0);
#endif
    };

    f32x4 acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int wn = wave % WN, wm = wave / WN;
    const int frag = i * 64 + ((g ^ ((4 - (i >> 2)) & 3)) << 4);        // this lane's 16 bytes inside a 1 KiB block
    const int woff = wn * TN * 1024 + frag, xoff = (BN / 16 + wm * TM) * 1024 + frag;
    const int nchunks = cin / (16 * U);
#pragma unroll
    for (int c = 0; c < NS - 1; ++c)
        if (c < nchunks) request(c);
    // wait for stage c (requests of the stages behind it may stay in flight: a wave's requests complete in order) + workgroup barrier.  A COUNTED
    // wait keeps NS - 2 stages in flight across the barrier where __syncthreads() drains them all.  Raw barrier: nothing else in the loop
    // touches vector memory; the memory clobber keeps the fragment reads behind it.
    auto publish = [&](int c) {
#if ESTD_C1X1_COUNTED
        const int ahead = (c + NS - 1 < nchunks ? c + NS - 1 : nchunks) - (c + 1);      // stages requested beyond stage c (wave-uniform)
        if (NS >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(2 * LPW) : "memory");
        else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::"n"(LPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#else
        __syncthreads();
#endif
    };
    auto fragments = [&](int c, int u, float4 (&wq)[TN], float4 (&xq)[TM]) {
        const unsigned char* st = lds + (c % NS) * STAGE + u * NBLK * 1024;
#pragma unroll
        for (int a = 0; a < TN; ++a) wq[a] = *reinterpret_cast<const float4*>(st + woff + a * 1024);
#pragma unroll
        for (int b = 0; b < TM; ++b) xq[b] = *reinterpret_cast<const float4*>(st + xoff + b * 1024);
    };
    auto multiply = [&](const float4 (&wq)[TN], const float4 (&xq)[TM]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const float wv = e == 0 ? wq[a].x : e == 1 ? wq[a].y : e == 2 ? wq[a].z : wq[a].w;
#pragma unroll
                for (int b = 0; b < TM; ++b) {
                    const float xv = e == 0 ? xq[b].x : e == 1 ? xq[b].y : e == 2 ? xq[b].z : xq[b].w;
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, xv, acc[a][b], 0, 0, 0);
                }
            }
    };
    // Software pipeline over (stage, sub-chunk): the fragments of the NEXT sub-chunk are read before the MFMAs of the current one -- across
    // the stage boundary too: the wait + barrier that publishes stage c + 1 sits in front of the LAST sub-chunk's MFMAs of stage c (every wave
    // has read all of stage c by then: its slot is free for the request at the top of iteration c + 1), so neither the LDS latency nor the
    // barrier skew is exposed once per stage (one wave per SIMD on the small maps: nobody else would hide it).
    float4 wq[2][TN], xq[2][TM];
    publish(0);
    fragments(0, 0, wq[0], xq[0]);
    for (int c = 0; c < nchunks; ++c) {
        if (c + NS - 1 < nchunks) request(c + NS - 1);   // into the slot of stage c - 1 (released by the barrier of the previous iteration)
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cur = u & 1, nxt = cur ^ 1;
            if (u + 1 < U) {
                fragments(c, u + 1, wq[nxt], xq[nxt]);
            } else if (c + 1 < nchunks) {                 // (wave-uniform)
                publish(c + 1);
                fragments(c + 1, 0, wq[(U & 1) ? nxt : 0], xq[(U & 1) ? nxt : 0]);
            }
            multiply(wq[cur], xq[cur]);
        }
        if (U & 1) {                                      // odd U: the buffers have swapped roles for the next stage -- swap them back (register moves)
#pragma unroll
            for (int a = 0; a < TN; ++a) wq[0][a] = wq[1][a];
#pragma unroll
            for (int b = 0; b < TM; ++b) xq[0][b] = xq[1][b];
        }
    }

    // ---- epilogue: folded BatchNorm, + residual, ReLU; 16-byte stores (as in the direct form) ----
    const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(p.out, (size_t)Mtot * cout * 4);
    const __amdgpu_buffer_rsrc_t rs_r = make_rsrc(p.residual ? p.residual : p.out, (size_t)Mtot * cout * 4);
    const float floor_ = p.relu ? 0.0f : ESTD_NO_FLOOR;
    const int mw = m0 + wm * TM * 16, nw = n0 + wn * TN * 16;
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int cb = nw + 16 * a + 4 * g;
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        const bool cok = cb < cout;                       // (cout is a multiple of 32: whole 16-channel tiles)
        if (p.scale && cok) sc = *reinterpret_cast<const float4*>(p.scale + cb);
        if (p.shift && cok) sh = *reinterpret_cast<const float4*>(p.shift + cb);
        unsigned ooff[TM];
        float4 res[TM];
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            const int m = mw + 16 * b + i;
            ooff[b] = (m < Mtot && cok) ? (unsigned)((size_t)m * cout + cb) * 4u : OOB_OFFSET;
        }
        if (p.residual) {
#pragma unroll
            for (int b = 0; b < TM; ++b) res[b] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_r, ooff[b], 0, 0));
        }
#pragma unroll
        for (int b = 0; b < TM; ++b) {
            float4 v;
            v.x = fmaf(acc[a][b][0], sc.x, sh.x); v.y = fmaf(acc[a][b][1], sc.y, sh.y);
            v.z = fmaf(acc[a][b][2], sc.z, sh.z); v.w = fmaf(acc[a][b][3], sc.w, sh.w);
            if (p.residual) { v.x += res[b].x; v.y += res[b].y; v.z += res[b].z; v.w += res[b].w; }
            v.x = fmaxf(v.x, floor_); v.y = fmaxf(v.y, floor_); v.z = fmaxf(v.z, floor_); v.w = fmaxf(v.w, floor_);
            u32x4 bits;
            __builtin_memcpy(&bits, &v, 16);
            __builtin_amdgcn_raw_buffer_store_b128(bits, rs_o, ooff[b], 0, 0);
        }
    }
}

template <int BM, int BN, int WM, int WN, int U, int NS>
int launch1x1_lds(const estd_conv1x1_desc& d, int Ho, int Wo, hipStream_t stream)
{
    constexpr int LDS_BYTES = (BM + BN) / 16 * U * 1024 * NS;
    const long long Mtot = (long long)d.N * Ho * Wo;
    const int tiles_m = (int)((Mtot + BM - 1) / BM), tiles_n = (d.cout + BN - 1) / BN;
    const int per_xcd = (tiles_m * tiles_n + 7) / 8;
    estd_allow_dynamic_lds<conv1x1_lds_kernel<BM, BN, WM, WN, U, NS>>(LDS_BYTES);
    hipLaunchKernelGGL((conv1x1_lds_kernel<BM, BN, WM, WN, U, NS>), dim3((unsigned)per_xcd * 8), dim3(256), LDS_BYTES, stream, d, Ho, Wo, tiles_m, tiles_n,
                       per_xcd);
    return ESTD_LAUNCH_CHECK();
}

template <int TM, int TN, int PF, int SK>
int launch1x1(const estd_conv1x1_desc& d, int Ho, int Wo, hipStream_t stream)
{
    const long long Mtot = (long long)d.N * Ho * Wo;
    const int tiles_m = (int)((Mtot + 16 * TM - 1) / (16 * TM)), tiles_n = d.cout / (16 * TN);
    const long long wts = (long long)tiles_m * tiles_n;
    const unsigned grid = SK > 1 ? (unsigned)wts : (unsigned)((wts + 3) / 4);
    hipLaunchKernelGGL((conv1x1_nhwc_kernel<TM, TN, PF, SK>), dim3(grid), dim3(256), 0, stream, d, Ho, Wo, tiles_m, tiles_n);
    return ESTD_LAUNCH_CHECK();
}

}  // namespace

extern "C" int estd_conv1x1_nhwc(const estd_conv1x1_desc* dp, estd_stream_t s)
{
    if (!dp) return ESTD_ERR_ARG;
    const estd_conv1x1_desc& d = *dp;
    if (d.N <= 0 || d.H <= 0 || d.W <= 0 || !d.in || !d.w || !d.out) return ESTD_ERR_ARG;
    if (d.stride != 1 && d.stride != 2) return ESTD_ERR_UNSUPPORTED;
    if (d.cin < 16 || (d.cin & 15) || d.cout < 32 || (d.cout & 31)) return ESTD_ERR_UNSUPPORTED;
    const int Ho = (d.H - 1) / d.stride + 1, Wo = (d.W - 1) / d.stride + 1;           // kernel 1, padding 0
    const long long Mtot = (long long)d.N * Ho * Wo;
    // 32-bit byte offsets inside the input map, the output map and the weight matrix
    if ((long long)d.N * d.H * d.W * d.cin * 4 >= 0x7fffff00LL || Mtot * d.cout * 4 >= 0x7fffff00LL || (long long)d.cin * d.cout * 4 >= 0x7fffff00LL)
        return ESTD_ERR_UNSUPPORTED;
    hipStream_t stream = estd_stream(s);
    // Block per wave: the largest one that still gives the device ~one wave per SIMD (tools/conv1x1_bench.py).  ESTD_C1X1_CFG = 100 TM + 10 TN + SK
    // forces one (A/B).
    static const int cfg_env = [] { const char* e = getenv("ESTD_C1X1_CFG"); return e ? atoi(e) : 0; }();
    const long long want = (long long)estd_device_cus() * 7 / 2;                        // 896 on 256 CUs
    auto tiles = [&](int tm, int tn) { return ((Mtot + 16 * tm - 1) / (16 * tm)) * (d.cout / (16 * tn)); };
    const bool sk4 = (d.cin & 63) == 0 && d.cin >= 256;                                 // four K ranges of whole chunks, long enough to be worth the exchange
    int cfg = cfg_env;
    // LDS-tiled form: cfg = 1000 + 100 (BM / 32) + 10 (BN / 32) + U  (ESTD_C1X1_CFG forces one; ESTD_C1X1_LDS=0: direct form only)
    static const int lds_env = [] { const char* e = getenv("ESTD_C1X1_LDS"); return e ? atoi(e) : 1; }();
    if (cfg == 0 && lds_env && (d.cout & 63) == 0) {
        // Which form, from the sweep of tools/conv1x1_cfg_sweep.py over the sixteen ResNet-50 shapes (profiles/r6_conv1x1_sweep.txt; with the counted
        // waits the tiled form is at or ahead of the direct form on every one of them): 64 x 64 workgroup tiles wherever there is about one per CU
        // -- 16-channel stages up to 256 input channels, 32-channel stages from 512 on; maps with fewer tiles take 32-pixel tiles and 64-channel
        // stages (1024 -> 256 at 30x40, 2048 -> 512 at 15x20); below that the direct form (K split over the four waves of a workgroup) stays.
        auto wgs = [&](int bm, int bn) { return ((Mtot + bm - 1) / bm) * ((d.cout + bn - 1) / bn); };
        const long long cus = estd_device_cus();
        const int c64 = (d.cin >= 512 && (d.cin & 31) == 0) ? 1222 : 1221;
        if (wgs(64, 64) >= cus * 7 / 4) cfg = c64;
        else if (wgs(32, 64) >= cus * 7 / 4 && (d.cin & 63) == 0) cfg = 1124;
        else if (wgs(64, 64) >= cus * 7 / 8) cfg = c64;
        else if (wgs(32, 64) >= cus * 3 / 4 && (d.cin & 63) == 0) cfg = 1124;
    }
    if (cfg >= 1000 && (d.cin % (16 * (cfg % 10)))) cfg = 0;                              // whole stages only
    switch (cfg) {
    case 1441: return launch1x1_lds<128, 128, 2, 2, 1, 3>(d, Ho, Wo, stream);
    case 1442: return launch1x1_lds<128, 128, 2, 2, 2, 3>(d, Ho, Wo, stream);
    case 1421: return launch1x1_lds<128, 64, 2, 2, 1, 3>(d, Ho, Wo, stream);
    case 1422: return launch1x1_lds<128, 64, 2, 2, 2, 3>(d, Ho, Wo, stream);
    case 1241: return launch1x1_lds<64, 128, 2, 2, 1, 3>(d, Ho, Wo, stream);
    case 1242: return launch1x1_lds<64, 128, 2, 2, 2, 3>(d, Ho, Wo, stream);
    case 1221: return launch1x1_lds<64, 64, 2, 2, 1, 4>(d, Ho, Wo, stream);
    case 1222: return launch1x1_lds<64, 64, 2, 2, 2, 3>(d, Ho, Wo, stream);
    case 1224: return launch1x1_lds<64, 64, 2, 2, 4, 3>(d, Ho, Wo, stream);
    case 1122: return launch1x1_lds<32, 64, 1, 4, 2, 3>(d, Ho, Wo, stream);
    case 1124: return launch1x1_lds<32, 64, 1, 4, 4, 3>(d, Ho, Wo, stream);
    default: break;
    }
    if (cfg >= 1000) cfg = 0;
    if (cfg == 0) {
        const bool c64 = (d.cout & 63) == 0;
        if (c64 && tiles(4, 4) >= want) cfg = 441;
        else if (sk4 && d.cin >= 1024 && c64 && tiles(4, 4) >= want / 4) cfg = 444;     // few blocks, long K: split K over the four waves
        else if (tiles(4, 2) >= want) cfg = 421;
        else if (sk4 && c64 && tiles(4, 4) >= want / 4) cfg = 444;
        else if (sk4 && c64 && tiles(2, 4) >= want / 4) cfg = 244;
        else if (tiles(2, 2) >= want) cfg = 221;
        else if (sk4 && tiles(2, 2) >= want / 4) cfg = 224;
        else cfg = 121;
    }
    if ((cfg % 10) == 4 && !((d.cin & 63) == 0)) cfg = cfg - 3;                         // SK needs cin % 64 == 0
    if ((cfg / 10) % 10 == 4 && (d.cout & 63)) cfg -= 20;                               // TN = 4 needs cout % 64 == 0
    switch (cfg) {
    case 441: return launch1x1<4, 4, 2, 1>(d, Ho, Wo, stream);
    case 421: return launch1x1<4, 2, 3, 1>(d, Ho, Wo, stream);
    case 241: return launch1x1<2, 4, 3, 1>(d, Ho, Wo, stream);
    case 221: return launch1x1<2, 2, 4, 1>(d, Ho, Wo, stream);
    case 444: return launch1x1<4, 4, 2, 4>(d, Ho, Wo, stream);
    case 424: return launch1x1<4, 2, 3, 4>(d, Ho, Wo, stream);
    case 244: return launch1x1<2, 4, 3, 4>(d, Ho, Wo, stream);
    case 224: return launch1x1<2, 2, 4, 4>(d, Ho, Wo, stream);
    case 124: return launch1x1<1, 2, 4, 4>(d, Ho, Wo, stream);
    default: return launch1x1<1, 2, 8, 1>(d, Ho, Wo, stream);
    }
}
