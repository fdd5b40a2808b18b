// conv2d_split_bf16.hip -- the 3x3 / stride 1 / dilation 1|2 NHWC convolution of conv2d_mfma.hip with every fp32 product
// evaluated as six bf16 MFMA products of exactly 3-way split operands (see conv3d_split_bf16.hip for the arithmetic and
// the error bound).  Same descriptor and epilogue (folded BatchNorm, ReLU before/after the residual add) as
// estd_conv2d_k3 (networks/layers_op.py:10-27, psm_submodule.py:14-37); reads ``w_split`` instead of ``w``.
//
// Structure = the 3D split kernel with "input-channel chunk" in the role of "depth slice":
//   * 512-thread workgroup, one per CU; work item = (32 output channels, image, 8 x 32-pixel tile); wave = tile row
//     (dilation 2: 8 x 16-pixel tiles, one M tile per wave, so that two 12 x 20-pixel brick slots fit LDS).
//   * The haloed 10 x 34-pixel brick of one 32-channel chunk is split into bf16 pieces when it enters LDS
//     ([piece][8-channel chunk][pixel] x 16 B, pixel index rotated by 2*chunk: conflict-free fill and fragment reads);
//     two brick slots: while chunk step s computes its 9 taps, the brick of step s+1 is written (taps 0..2, from
//     registers loaded during step s-1) and the brick of step s+2 is fetched from HBM/L2 (taps 3..8).  Steps run across
//     chunk, tile and item boundaries without a pipeline drain.
//   * Weights: 8 KB records [3 pieces][2 n-tiles][64 lanes][8] per (group, chunk, tap), streamed L2 -> registers -> a
//     2-slot LDS buffer one tap ahead; fragments of tap t+1 are read during tap t; one LDS-only barrier per tap.
//   * A tap is one basic block with a prescribed issue order (one MFMA, then a few other instructions); the epilogue
//     of a finished item is deferred into the first taps of the following step.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>
#include <utility>

#include "estd_hip.h"
#include "estd_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((__vector_size__(16)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((__vector_size__(16)));
typedef unsigned int u32x2 __attribute__((__vector_size__(8)));

constexpr int TH = 8;
constexpr int WREC_BYTES = 512 * 16;           // weight record of one (group, chunk, tap); 6144 used
constexpr unsigned OOB_OFFSET = 0xFFFFFF00u;

// Geometry of one instance: dilation 1 -> 8 x 32-pixel tiles (two 16-pixel M tiles per wave), brick 10 x 34;
//                           dilation 2 -> 8 x 16-pixel tiles (one M tile per wave), brick 12 x 20 (two brick slots must fit LDS)
template <int DIL> struct Geo {
    static constexpr int MT = DIL == 1 ? 2 : 1;
    static constexpr int TW = 16 * MT;
    static constexpr int IN_H = TH + 2 * DIL, IN_W = TW + 2 * DIL;
    static constexpr int NPIX = IN_H * IN_W;                       // 340 / 240 pixels per brick (with halo)
    static constexpr int PLANE = (NPIX + 6 + 15) / 16 * 16;        // 352 / 256
    static constexpr int CHUNK_BYTES = PLANE * 16;
    static constexpr int PIECE_BYTES = 4 * CHUNK_BYTES;
    static constexpr int SLOT_BYTES = 3 * PIECE_BYTES;             // 67584 / 49152
    static constexpr int FILL_E = NPIX * 4;
    static constexpr int FIT = (FILL_E + 511) / 512;               // 3 / 2
    static constexpr int LAST_FULL = FILL_E - 512 * (FIT - 1);     // threads below this own a last item
    static constexpr int LDS_W = 2 * SLOT_BYTES;
    static constexpr int LDS_DUMP = LDS_W + 2 * WREC_BYTES;
    static constexpr int LDS_TOTAL = LDS_DUMP + (512 - LAST_FULL) * 16;
};

__device__ __forceinline__ float4 as_float4(u32x4 v) { float4 f; __builtin_memcpy(&f, &v, 16); return f; }
__device__ __forceinline__ float2 as_float2(u32x2 v) { float2 f; __builtin_memcpy(&f, &v, 8); return f; }

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

template <typename F, int... I>
__device__ __forceinline__ void for_each_tap(F&& f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}

template <int TAP, int MT>
__device__ __forceinline__ void tap_pipeline()
{
    constexpr int NM = 12 * MT, NR = 3 * MT + 6;
#pragma unroll
    for (int k = 0; k < NM; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // MFMA
        if (k < NR - NM / 2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // DS reads (more reads than early MFMA slots when MT = 1)
        else if (k < NM / 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if (TAP <= 2) __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);        // VALU: slice split + epilogue
        if (k >= NM / 2 && k < NM / 2 + 4) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // DS write
        if (k >= NM / 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
        if (TAP >= 1 && TAP <= 2 && k >= NM - 4) __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);   // VMEM write
    }
}

struct Step { int grp, n, th0, tw0, c; bool valid; };

template <int DIL, bool RES>
__global__ __launch_bounds__(512, 1) void conv2d_k3_split_kernel(const estd_conv2d_desc p, int tiles_w, int tiles_h, int total_items)
{
    typedef Geo<DIL> G_;
    constexpr int MT = G_::MT, TW = G_::TW, IN_W = G_::IN_W, CHUNK_BYTES = G_::CHUNK_BYTES, PIECE_BYTES = G_::PIECE_BYTES;
    constexpr int SLOT_BYTES = G_::SLOT_BYTES, FILL_E = G_::FILL_E, FIT = G_::FIT, LDS_W = G_::LDS_W, LDS_DUMP = G_::LDS_DUMP;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, i = lane & 15;
    const int H = p.H, W = p.W, Cin = p.cin, Cout = p.cout;
    const int nchunks = Cin >> 5;
    const int tiles_per_group = p.N * tiles_h * tiles_w;

    int u0, u_end;
    {
        const int G = gridDim.x, bid = blockIdx.x;
        const int r = ((G & 7) == 0) ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;
        u0 = (int)((long long)total_items * r / G);
        u_end = (int)((long long)total_items * (r + 1) / G);
    }
    if (u0 >= u_end) return;
    const int nsteps = (u_end - u0) * nchunks;

    const size_t in_bytes = (size_t)p.N * H * W * Cin * 4, out_bytes = (size_t)p.N * H * W * Cout * 4;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, in_bytes);
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(p.out, out_bytes);
    const __amdgpu_buffer_rsrc_t rs_res = make_rsrc(p.residual ? p.residual : p.out, p.residual ? out_bytes : 0);
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(p.w_split, (size_t)(Cout >> 5) * nchunks * 9 * WREC_BYTES);
    const float floor_before = p.relu_before_residual ? 0.0f : -__builtin_huge_valf();
    const float floor_after = p.relu_after_residual ? 0.0f : -__builtin_huge_valf();

    const int a_lane = g * CHUNK_BYTES + (i + 2 * g) * 16;
    const int b_lane = lane * 16;
    const int w_lane = tid * 16;
    const int dump16 = LDS_DUMP + (tid >= G_::LAST_FULL ? tid - G_::LAST_FULL : 0) * 16;

    // ---- the WG's sequence of chunk steps: step s = (item u0 + s / nchunks, chunk s % nchunks) ----
    auto decode_step = [&](int s) {
        Step st;
        st.valid = s < nsteps;
        const int sc = st.valid ? s : 0;
        const int item = u0 + sc / nchunks;
        st.c = sc % nchunks;
        st.grp = item / tiles_per_group;
        int t = item - st.grp * tiles_per_group;
        const int twi = t % tiles_w; t /= tiles_w;
        const int thi = t % tiles_h; st.n = t / tiles_h;
        st.th0 = thi * TH; st.tw0 = twi * TW;
        return st;
    };
    // per-thread byte offsets of the 3 fill items (pixel e/4, 8-channel group e%4) of a step's brick; OOB -> zeros
    int loff[FIT];
#pragma unroll
    for (int it = 0; it < FIT; ++it) {
        const int e = tid + it * 512;
        loff[it] = e < FILL_E ? (e & 3) * CHUNK_BYTES + ((e >> 2) + 2 * (e & 3)) * 16 : -1;
    }
    const bool last_item = loff[FIT - 1] >= 0;
    auto brick_voff = [&](const Step& st, unsigned (&v)[FIT]) {
#pragma unroll
        for (int it = 0; it < FIT; ++it) {
            const int e = tid + it * 512;
            const int vs = e >> 2, c4 = e & 3;
            const int zy = vs / IN_W, zx = vs - zy * IN_W;
            const int gy = st.th0 - DIL + zy, gx = st.tw0 - DIL + zx;
            const bool ok = st.valid && e < FILL_E && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            v[it] = ok ? (unsigned)(((st.n * H + gy) * W + gx) * Cin + st.c * 32 + c4 * 8) * 4u : OOB_OFFSET;
        }
    };
    auto fill = [&](int slot_bytes, int it, float4 a, float4 b) {
        const bool real = it < FIT - 1 || last_item;
        const int o0 = real ? slot_bytes + loff[it] : dump16;
        const int st = real ? PIECE_BYTES : 0;
        fill_item(smem, o0, o0 + st, o0 + 2 * st, a, b);
    };
    auto wrec = [&](const Step& st) { return ((st.grp * nchunks + st.c) * 9) * WREC_BYTES; };     // byte offset of the step's tap-0 record

    // ---- prologue: brick of step 0 -> slot 0, brick of step 1 -> registers, weights of taps 0/1 -> LDS, tap 2 -> registers ----
    Step cur = decode_step(0), nxt = decode_step(1);
    float4 pf[2 * FIT];
    {
        unsigned v[FIT];
        brick_voff(cur, v);
        float4 t0[FIT], t1[FIT];
#pragma unroll
        for (int it = 0; it < FIT; ++it) {
            t0[it] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, v[it], 0, 0));
            t1[it] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, v[it], 16, 0));
        }
#pragma unroll
        for (int it = 0; it < FIT; ++it) fill(0, it, t0[it], t1[it]);
        brick_voff(nxt, v);
#pragma unroll
        for (int it = 0; it < FIT; ++it) {
            pf[2 * it] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, v[it], 0, 0));
            pf[2 * it + 1] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, v[it], 16, 0));
        }
    }
    int wcur = wrec(cur), wnxt = nxt.valid ? wrec(nxt) : wcur;
    u32x4 wreg;
    {
        const u32x4 w0 = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, wcur, 0);
        const u32x4 w1 = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, wcur + WREC_BYTES, 0);
        wreg = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, wcur + 2 * WREC_BYTES, 0);
        *reinterpret_cast<u32x4*>(smem + LDS_W + w_lane) = w0;
        *reinterpret_cast<u32x4*>(smem + LDS_W + WREC_BYTES + w_lane) = w1;
    }
    lds_barrier();
    bf16x8 acur[MT][3], bcur[3][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
            acur[m][pc] = *reinterpret_cast<const bf16x8*>(smem + pc * PIECE_BYTES + a_lane + (wave * IN_W + 16 * m) * 16);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
            bcur[pc][nn] = *reinterpret_cast<const bf16x8*>(smem + LDS_W + (pc * 2 + nn) * 1024 + b_lane);
    lds_barrier();

    int wsel = 1;
    f32x4 acc[MT][2], pend[MT][2] = {};
    unsigned eoff_p[MT][4];             // output offsets of the pending (finished) item
    float psc0 = 0.f, psh0 = 0.f, psc1 = 0.f, psh1 = 0.f;
    bool pend_live = false;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) eoff_p[m][r] = OOB_OFFSET;
    struct EpiLoads { u32x2 r1[4]; } el;

    auto epi_load = [&](int m) {
        if (!RES) return;
#pragma unroll
        for (int r = 0; r < 4; ++r) el.r1[r] = __builtin_amdgcn_raw_buffer_load_b64(rs_res, pend_live ? eoff_p[m][r] : OOB_OFFSET, 0, 0);
    };
    auto epi_finish = [&](int m) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned eo = pend_live ? eoff_p[m][r] : OOB_OFFSET;
            float v0 = fmaxf(pend[m][0][r] * psc0 + psh0, floor_before);
            float v1 = fmaxf(pend[m][1][r] * psc1 + psh1, floor_before);
            if (RES) {
                const float2 q = as_float2(el.r1[r]);
                v0 += q.x; v1 += q.y;
            }
            v0 = fmaxf(v0, floor_after);
            v1 = fmaxf(v1, floor_after);
            const float2 ov = make_float2(v0, v1);
            u32x2 od; __builtin_memcpy(&od, &ov, 8);
            __builtin_amdgcn_raw_buffer_store_b64(od, rs_out, eo, 0, 0);
        }
    };

    for (int s = 0; s < nsteps; ++s) {
        const int sb = (s & 1) * SLOT_BYTES, sbn = ((s + 1) & 1) * SLOT_BYTES;
        if (cur.c == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) acc[m][nn] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const Step nn2 = decode_step(s + 2);
        unsigned v2[FIT];
        brick_voff(nn2, v2);

        for_each_tap([&](auto tap_c) __attribute__((always_inline)) {
            constexpr int tap = decltype(tap_c)::value;
            constexpr int nt = tap == 8 ? 0 : tap + 1;
            constexpr int nkh = nt / 3, nkw = nt % 3;
            const int nsb = tap == 8 ? sbn : sb;
            const int wrd = LDS_W + wsel * WREC_BYTES, wwr = LDS_W + (wsel ^ 1) * WREC_BYTES;

            bf16x8 anext[MT][3], bnext[3][2];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc)
                    anext[m][pc] = *reinterpret_cast<const bf16x8*>(smem + nsb + pc * PIECE_BYTES + a_lane + ((wave + nkh * DIL) * IN_W + nkw * DIL + 16 * m) * 16);
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn)
                    bnext[pc][nn] = *reinterpret_cast<const bf16x8*>(smem + wrd + (pc * 2 + nn) * 1024 + b_lane);
            // weights of tap+2 -> LDS, tap+3 -> registers (the stream continues into the next step)
            *reinterpret_cast<u32x4*>(smem + wwr + w_lane) = wreg;
            wreg = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_lane, tap + 3 < 9 ? wcur + (tap + 3) * WREC_BYTES : wnxt + (tap + 3 - 9) * WREC_BYTES, 0);
            // brick of step s+1: registers -> the other slot (taps 0..2); brick of step s+2: HBM/L2 -> registers (taps 3..8)
            if constexpr (tap < FIT) fill(sbn, tap, pf[2 * tap], pf[2 * tap + 1]);
            if constexpr (tap >= 3 && tap < 3 + 2 * FIT) {
                constexpr int k = tap - 3;
                pf[k] = as_float4(__builtin_amdgcn_raw_buffer_load_b128(rs_in, v2[k >> 1], (k & 1) * 16, 0));
            }
            // deferred epilogue of the item that finished with the previous step
            if constexpr (tap == 0) epi_load(0);
            if constexpr (tap == 1) { epi_finish(0); if (MT == 2) epi_load(MT - 1); }
            if constexpr (tap == 2) { if (MT == 2) epi_finish(MT - 1); }
            {
                constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int nn = 0; nn < 2; ++nn)
                            acc[m][nn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(acur[m][PA[t]], bcur[PB[t]][nn], acc[m][nn], 0, 0, 0);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) acur[m][pc] = anext[m][pc];
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) bcur[pc][nn] = bnext[pc][nn];
            wsel ^= 1;
            tap_pipeline<tap, MT>();
            __builtin_amdgcn_sched_barrier(0);
            lds_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }, std::make_integer_sequence<int, 9>{});

        pend_live = false;                       // the pending item (if any) was written during taps 0..2
        if (cur.c == nchunks - 1) {              // item finished: hand its accumulators to the deferred epilogue
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) pend[m][nn] = acc[m][nn];
            const int cb = cur.grp * 32 + 2 * i;
            psc0 = p.scale[cb]; psh0 = p.shift[cb]; psc1 = p.scale[cb + 1]; psh1 = p.shift[cb + 1];
            const int y = cur.th0 + wave;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int x = cur.tw0 + 16 * m + 4 * g + r;
                    eoff_p[m][r] = (y < H && x < W) ? (unsigned)(((cur.n * H + y) * W + x) * Cout + cb) * 4u : OOB_OFFSET;
                }
            pend_live = true;
        }
        cur = nxt; nxt = nn2;
        wcur = wnxt; wnxt = nxt.valid ? wrec(nxt) : wcur;
    }
    if (pend_live) {
        epi_load(0); epi_finish(0);
        if (MT == 2) { epi_load(MT - 1); epi_finish(MT - 1); }
    }
}

template <int DIL, bool RES>
int launch_split2d(const estd_conv2d_desc& d, hipStream_t stream)
{
    typedef Geo<DIL> G_;
    const int tiles_w = (d.W + G_::TW - 1) / G_::TW, tiles_h = (d.H + TH - 1) / TH;
    const long long total = (long long)(d.cout >> 5) * d.N * tiles_h * tiles_w;
    if (total > 0x7fffffffLL) return ESTD_ERR_ARG;
    const int slots = estd_persistent_wgs(1);
    int grid = total < slots ? (int)total : slots;
    if (grid >= 8) grid &= ~7;
    estd_allow_dynamic_lds<conv2d_k3_split_kernel<DIL, RES>>((int)G_::LDS_TOTAL);
    hipLaunchKernelGGL((conv2d_k3_split_kernel<DIL, RES>), dim3(grid), dim3(512), G_::LDS_TOTAL, stream, d, tiles_w, tiles_h, (int)total);
    return hipGetLastError() == hipSuccess ? ESTD_OK : ESTD_ERR_LAUNCH;
}

}  // namespace

extern "C" int estd_conv2d_k3_split(const estd_conv2d_desc* dp, estd_stream_t s)
{
    if (!dp) return ESTD_ERR_ARG;
    const estd_conv2d_desc& d = *dp;
    if (d.N <= 0 || d.H <= 0 || d.W <= 0 || !d.in || !d.w_split || !d.scale || !d.shift || !d.out) return ESTD_ERR_ARG;
    if (d.cin < 32 || (d.cin & 31) || d.cout < 32 || (d.cout & 31)) return ESTD_ERR_ARG;
    if (d.dilation != 1 && d.dilation != 2) return ESTD_ERR_UNSUPPORTED;
    const long long widest = (long long)d.N * d.H * d.W * (d.cin > d.cout ? d.cin : d.cout) * 4;
    if (widest >= 0x7fffff00LL) return ESTD_ERR_UNSUPPORTED;          // one descriptor spans the whole batch
    if ((long long)(d.cout >> 5) * (d.cin >> 5) * 9 * WREC_BYTES >= 0x7fffff00LL) return ESTD_ERR_UNSUPPORTED;
    hipStream_t stream = static_cast<hipStream_t>(s);
    if (d.residual) return d.dilation == 1 ? launch_split2d<1, true>(d, stream) : launch_split2d<2, true>(d, stream);
    return d.dilation == 1 ? launch_split2d<1, false>(d, stream) : launch_split2d<2, false>(d, stream);
}
