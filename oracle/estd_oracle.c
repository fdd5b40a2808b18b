/*
 * estd_oracle.c -- CPU ORACLE (test infrastructure, NOT the product path).
 *
 * Plain-C restatement of the arithmetic on ESTDepth's plane-sweep + EST-transformer
 * hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; estdepth_amd/ never does.
 *
 * Each function cites the reference (/root/reference) file:line it follows.  The
 * reference's arithmetic is a composition of PyTorch ATen ops (torch 2.10 CPU is the
 * oracle version named by BASELINE.json:north_star); the ATen semantics restated here
 * are: grid_sample bilinear/trilinear, padding zeros, align_corners=False
 * (ATen/native/GridSampler.h: unnormalize = ((c+1)*size-1)/2; out-of-bounds corners
 * contribute 0), nearest upsample src=floor(dst/scale), BatchNorm eval affine,
 * GroupNorm(1 group) population variance, max-subtracted softmax.
 *
 * Parity pin: tests/test_oracle_ops_golden.py + tests/test_oracle_model_golden.py checks every function against golden
 * vectors produced by tools/gen_golden.py, which imports the reference itself.
 *
 * All tensors fp32, contiguous, batch-less (the Python wrapper loops over batch).
 * Layouts are the reference's: [C][H][W] and [C][D][H][W].
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------- */
/* homo_warping: utils/homo_utils.py:458-504                                  */
/*   rot/trans come from proj = src_proj @ inverse(ref_proj) (:469-471), which */
/*   the wrapper computes with LAPACK in fp32 exactly like torch.inverse.      */
/* ------------------------------------------------------------------------- */
void orc_homo_warping(const float* src, const float* rot /*3x3*/, const float* trans /*3*/,
                      const float* depth_values /*D, or [D][H*W] when per_pixel (:462)*/, int per_pixel, int C, int H, int W, int D,
                      float* out /*[C][D][H][W]*/)
{
    const long HW = (long)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int d = 0; d < D; ++d) {
        for (int y = 0; y < H; ++y) {
            for (int x = 0; x < W; ++x) {
                const float fx = (float)x, fy = (float)y;
                /* rot_xyz = rot @ [x,y,1]   (:479): a torch.matmul, i.e. a GEMM kernel that accumulates k = 0,1,2 in
                 * order with fused multiply-adds (checked bit for bit in tests/test_oracle_ops_golden.py); every other
                 * step is an elementwise ATen op with its own rounding (the file is built with -ffp-contract=off). */
                float r0 = fmaf(rot[1], fy, rot[0] * fx) + rot[2];
                float r1 = fmaf(rot[4], fy, rot[3] * fx) + rot[5];
                float r2 = fmaf(rot[7], fy, rot[6] * fx) + rot[8];
                /* * depth + trans  (:480-482) */
                const float dv = per_pixel ? depth_values[(long)d * HW + (long)y * W + x] : depth_values[d];
                float p0 = r0 * dv + trans[0];
                float p1 = r1 * dv + trans[1];
                float p2 = r2 * dv + trans[2];
                /* proj_xy = xy / (z + 1e-8)  (:483)  -- no z>0 check in the reference */
                float px = p0 / (p2 + 1e-8f);
                float py = p1 / (p2 + 1e-8f);
                float xn = px / ((float)(W - 1) / 2.0f) - 1.0f;   /* :484 */
                float yn = py / ((float)(H - 1) / 2.0f) - 1.0f;   /* :485 */
                if (xn > 1.0f || xn < -1.0f) xn = 2.0f;           /* :488-489 */
                if (yn > 1.0f || yn < -1.0f) yn = 2.0f;           /* :490-491 */
                /* grid_sample(bilinear, zeros, align_corners=False)  (:499-501) */
                float ix = ((xn + 1.0f) * (float)W - 1.0f) / 2.0f;
                float iy = ((yn + 1.0f) * (float)H - 1.0f) / 2.0f;
                float fx0 = floorf(ix), fy0 = floorf(iy);
                int x0 = (int)fx0, y0 = (int)fy0;
                int x1 = x0 + 1, y1 = y0 + 1;
                float wx1 = ix - fx0, wx0 = 1.0f - wx1;
                float wy1 = iy - fy0, wy0 = 1.0f - wy1;
                float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
                int vx0 = (x0 >= 0 && x0 < W), vx1 = (x1 >= 0 && x1 < W);
                int vy0 = (y0 >= 0 && y0 < H), vy1 = (y1 >= 0 && y1 < H);
                /* NaN coordinates (possible when p2+1e-8 == 0): ATen treats the
                   comparisons as false -> all corners out of bounds -> 0. */
                if (!(ix == ix) || !(iy == iy)) { vx0 = vx1 = vy0 = vy1 = 0; x0 = y0 = x1 = y1 = 0; }
                for (int c = 0; c < C; ++c) {
                    const float* s = src + (long)c * HW;
                    float v = 0.0f;
                    if (vy0 && vx0) v += s[(long)y0 * W + x0] * w00;
                    if (vy0 && vx1) v += s[(long)y0 * W + x1] * w01;
                    if (vy1 && vx0) v += s[(long)y1 * W + x0] * w10;
                    if (vy1 && vx1) v += s[(long)y1 * W + x1] * w11;
                    out[((long)c * D + d) * HW + (long)y * W + x] = v;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* warp_volume: utils/homo_utils.py:240-279 with helpers                       */
/*   pixel2cam :40-62, cam2cam :26-37, cam2pixel_depth :107-134,               */
/*   normalize_pixel_coords_volume :170-205, 5-D grid_sample :276-277.         */
/*   kinv = inverse(cam_intr) (:51), m = inverse(pose) (:258): from wrapper.   */
/* ------------------------------------------------------------------------- */
void orc_warp_volume(const float* vol /*[C][D][H][W]*/, const float* depth /*[D][H*W]*/,
                     const float* kinv /*3x3*/, const float* m /*4x4*/, const float* kmat /*3x3*/,
                     float depth_min, float depth_interval,
                     int use_disp, float disp_min, float disp_interval,      /* :187-190 disparity planes */
                     int border, float padding_value,                        /* :271-274 padding_mode='border' on _set_vol_border(vol) (:305-319) */
                     int C, int D, int H, int W, float* out /*[C][D][H][W]*/)
{
    const long HW = (long)H * W;
    const long DHW = (long)D * HW;
#pragma omp parallel for collapse(2) schedule(static)
    for (int d = 0; d < D; ++d) {
        for (int y = 0; y < H; ++y) {
            for (int x = 0; x < W; ++x) {
                const float fx = (float)x, fy = (float)y;
                const float dep = depth[(long)d * HW + (long)y * W + x];
                /* The three matrix products are bmm calls: GEMM kernels accumulating k in order with fused
                 * multiply-adds (acc = a0*b0; acc = fma(a1,b1,acc); ...), see orc_homo_warping. */
                /* pixel2cam: (K^-1 @ [x,y,1]) * depth   (:51-54) */
                float c0 = (fmaf(kinv[1], fy, kinv[0] * fx) + kinv[2]) * dep;
                float c1 = (fmaf(kinv[4], fy, kinv[3] * fx) + kinv[5]) * dep;
                float c2 = (fmaf(kinv[7], fy, kinv[6] * fx) + kinv[8]) * dep;
                /* cam2cam: M @ [c;1]   (:33-36) */
                float s0 = fmaf(m[2], c2, fmaf(m[1], c1, m[0] * c0)) + m[3];
                float s1 = fmaf(m[6], c2, fmaf(m[5], c1, m[4] * c0)) + m[7];
                float s2 = fmaf(m[10], c2, fmaf(m[9], c1, m[8] * c0)) + m[11];
                /* cam2pixel_depth: K @ s[:3]; x/(z+1e-10), y/(z+1e-10), z   (:115-121) */
                float q0 = fmaf(kmat[2], s2, fmaf(kmat[1], s1, kmat[0] * s0));
                float q1 = fmaf(kmat[5], s2, fmaf(kmat[4], s1, kmat[3] * s0));
                float q2 = fmaf(kmat[8], s2, fmaf(kmat[7], s1, kmat[6] * s0));
                float X = q0 / (q2 + 1e-10f);
                float Y = q1 / (q2 + 1e-10f);
                float Z = q2;
                /* normalize_pixel_coords_volume (:183-198) */
                float xn = 2.0f * X / (float)(W - 1) - 1.0f;
                float yn = 2.0f * Y / (float)(H - 1) - 1.0f;
                float zn = use_disp ? 2.0f * ((1.0f / (Z + 1e-10f) - disp_min) / disp_interval) / (float)(D - 1) - 1.0f   /* :190 */
                                    : 2.0f * ((Z - depth_min) / depth_interval) / (float)(D - 1) - 1.0f;                  /* :187 */
                /* the masks are applied whatever warp_volume's padding_mode is: normalize_pixel_coords_volume is called with
                 * its own default padding_mode='zeros' (:262-269) */
                if (xn > 1.0f || xn < -1.0f) xn = 2.0f;
                if (yn > 1.0f || yn < -1.0f) yn = 2.0f;
                if (zn > 1.0f || zn < -1.0f) zn = 2.0f;
                /* 5-D grid_sample, 'bilinear' (= trilinear), align_corners=False; zeros, or border = ATen clip_coordinates
                 * min(size - 1, max(i, 0)) after the un-normalisation */
                float ix = ((xn + 1.0f) * (float)W - 1.0f) / 2.0f;
                float iy = ((yn + 1.0f) * (float)H - 1.0f) / 2.0f;
                float iz = ((zn + 1.0f) * (float)D - 1.0f) / 2.0f;
                if (border) {
                    ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));
                    iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
                    iz = fminf((float)(D - 1), fmaxf(iz, 0.0f));
                }
                float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
                int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
                float tx = ix - fx0, ty = iy - fy0, tz = iz - fz0;
                int nanc = !(ix == ix) || !(iy == iy) || !(iz == iz);
                float wgt[8];
                long off[8];
                int ok[8], edge[8];
                for (int k = 0; k < 8; ++k) {
                    int dx = k & 1, dy = (k >> 1) & 1, dz = (k >> 2) & 1;
                    int xx = x0 + dx, yy = y0 + dy, zz = z0 + dz;
                    ok[k] = !nanc && xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < D;
                    wgt[k] = (dx ? tx : 1.0f - tx) * (dy ? ty : 1.0f - ty) * (dz ? tz : 1.0f - tz);
                    off[k] = ok[k] ? ((long)zz * HW + (long)yy * W + xx) : 0;
                    /* _set_vol_border: the outermost voxel layer of the volume holds padding_value (:311-317) */
                    edge[k] = border && (xx == 0 || xx == W - 1 || yy == 0 || yy == H - 1 || zz == 0 || zz == D - 1);
                }
                for (int c = 0; c < C; ++c) {
                    const float* s = vol + (long)c * DHW;
                    float v = 0.0f;
                    for (int k = 0; k < 8; ++k)
                        if (ok[k]) v += (edge[k] ? padding_value : s[off[k]]) * wgt[k];
                    out[(long)c * DHW + (long)d * HW + (long)y * W + x] = v;
                }
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Conv3d, kernel k in {1,3}, stride 1, zero padding k/2, optional bias.       */
/*   networks/layers_op.py:16-39 (bias=False + BatchNorm3d [+ReLU/Tanh]) and   */
/*   transformer/epipolar_transformer.py:21,26 (bias=True), and the 1x1x1      */
/*   heads hybrid_depth_decoder.py:106,111.                                    */
/*   Internally channels-last so the inner loop over Cout vectorises.          */
/* ------------------------------------------------------------------------- */
void orc_conv3d(const float* in /*[Cin][D][H][W]*/, const float* w /*[Cout][Cin][k][k][k]*/,
                const float* bias /*Cout or NULL*/, int Cin, int Cout, int k,
                int D, int H, int W, float* out /*[Cout][D][H][W]*/)
{
    const int p = k / 2;
    const int Dp = D + 2 * p, Hp = H + 2 * p, Wp = W + 2 * p;
    const long HW = (long)H * W, DHW = (long)D * HW;
    const int T = k * k * k;
    float* xin = (float*)calloc((size_t)Dp * Hp * Wp * Cin, sizeof(float));
    float* wt = (float*)malloc((size_t)T * Cin * Cout * sizeof(float));
    /* repack input to padded [Dp][Hp][Wp][Cin] */
#pragma omp parallel for collapse(2) schedule(static)
    for (int d = 0; d < D; ++d)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float* dst = xin + (((long)(d + p) * Hp + (y + p)) * Wp + (x + p)) * Cin;
                for (int c = 0; c < Cin; ++c) dst[c] = in[(long)c * DHW + (long)d * HW + (long)y * W + x];
            }
    /* weights to [tap][Cin][Cout] */
    for (int o = 0; o < Cout; ++o)
        for (int c = 0; c < Cin; ++c)
            for (int t = 0; t < T; ++t)
                wt[((long)t * Cin + c) * Cout + o] = w[((long)o * Cin + c) * T + t];
#pragma omp parallel for collapse(2) schedule(static)
    for (int d = 0; d < D; ++d)
        for (int y = 0; y < H; ++y) {
            float acc[64];
            for (int x = 0; x < W; ++x) {
                for (int o = 0; o < Cout; ++o) acc[o] = bias ? bias[o] : 0.0f;
                for (int kd = 0; kd < k; ++kd)
                    for (int kh = 0; kh < k; ++kh)
                        for (int kw = 0; kw < k; ++kw) {
                            const float* src = xin + (((long)(d + kd) * Hp + (y + kh)) * Wp + (x + kw)) * Cin;
                            const float* wp = wt + (long)((kd * k + kh) * k + kw) * Cin * Cout;
                            for (int c = 0; c < Cin; ++c) {
                                const float a = src[c];
                                const float* wr = wp + (long)c * Cout;
                                for (int o = 0; o < Cout; ++o) acc[o] += a * wr[o];
                            }
                        }
                for (int o = 0; o < Cout; ++o)
                    out[(long)o * DHW + (long)d * HW + (long)y * W + x] = acc[o];
            }
        }
    free(xin);
    free(wt);
}

/* BatchNorm3d (eval) + activation, in place.  act: 0 none, 1 relu, 2 tanh.
 * y = x*(g*rsqrt(var+eps)) + (b - mean*g*rsqrt(var+eps))   (nn.BatchNorm3d eval;
 * networks/layers_op.py:19,32,38). */
void orc_bn_act(float* x /*[C][N]*/, const float* gamma, const float* beta, const float* mean,
                const float* var, float eps, int act, int C, long N)
{
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const float inv = 1.0f / sqrtf(var[c] + eps);
        const float sc = gamma[c] * inv;
        const float sh = beta[c] - mean[c] * sc;
        float* p = x + (long)c * N;
        for (long i = 0; i < N; ++i) {
            float v = p[i] * sc + sh;
            if (act == 1) v = v > 0.0f ? v : 0.0f;
            else if (act == 2) v = tanhf(v);
            p[i] = v;
        }
    }
}

/* GroupNorm(num_groups=1, C, eps, affine): transformer/epipolar_transformer.py:22-23,27.
 * Statistics over all C*N elements of the sample, population variance.  ATen uses a
 * cascaded fp32 Welford; double accumulation here is the exact-statistics restatement. */
void orc_groupnorm1(const float* x /*[C][N]*/, const float* gamma, const float* beta, float eps,
                    int C, long N, float* out)
{
    const long T = (long)C * N;
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (long i = 0; i < T; ++i) s += (double)x[i];
    const double mean = s / (double)T;
    double v = 0.0;
#pragma omp parallel for reduction(+ : v) schedule(static)
    for (long i = 0; i < T; ++i) { double dlt = (double)x[i] - mean; v += dlt * dlt; }
    const double var = v / (double)T;
    const float fmean = (float)mean;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c)
        for (long i = 0; i < N; ++i)
            out[(long)c * N + i] = (x[(long)c * N + i] - fmean) * rstd * gamma[c] + beta[c];
}

/* Epipolar attention: transformer/epipolar_transformer.py:62-73.
 *   corr_n = sum_c Kt*Kn (:65); softmax over n (:69); h = mean_n(Vn * a_n) (:73). */
void orc_epipolar_attention(const float* kt /*[C][N]*/, const float* wk /*[nv][C][N]*/,
                            const float* wv /*[nv][C][N]*/, int nv, int C, long N, float* h /*[C][N]*/)
{
#pragma omp parallel for schedule(static)
    for (long i = 0; i < N; ++i) {
        float corr[16];
        float mx = -INFINITY;
        for (int n = 0; n < nv; ++n) {
            float s = 0.0f;
            for (int c = 0; c < C; ++c) s += kt[(long)c * N + i] * wk[((long)n * C + c) * N + i];
            corr[n] = s;
            if (s > mx) mx = s;
        }
        float den = 0.0f;
        for (int n = 0; n < nv; ++n) { corr[n] = expf(corr[n] - mx); den += corr[n]; }
        for (int c = 0; c < C; ++c) {
            float acc = 0.0f;
            for (int n = 0; n < nv; ++n) acc += wv[((long)n * C + c) * N + i] * (corr[n] / den);
            h[(long)c * N + i] = acc / (float)nv;
        }
    }
}

/* depthlayer on nearest-upsampled logits: hybrid_depth_decoder.py:33-38 with
 * F.interpolate(scale_factor=s) (:202,:259,:359,:379): src index = floor(dst/s). */
void orc_depthlayer_up(const float* logits /*[D][H][W]*/, const float* depth_values /*D*/,
                       int D, int H, int W, int s, float* depth /*[sH][sW]*/, float* prob /*[sH][sW]*/)
{
    const long HW = (long)H * W;
    const int Wo = W * s, Ho = H * s;
#pragma omp parallel for schedule(static)
    for (int yo = 0; yo < Ho; ++yo) {
        for (int xo = 0; xo < Wo; ++xo) {
            const int y = yo / s, x = xo / s;
            const float* l = logits + (long)y * W + x;
            float mx = -INFINITY;
            for (int d = 0; d < D; ++d) { float v = l[(long)d * HW]; if (v > mx) mx = v; }
            float den = 0.0f;
            for (int d = 0; d < D; ++d) den += expf(l[(long)d * HW] - mx);
            float dep = 0.0f, pm = 0.0f;
            for (int d = 0; d < D; ++d) {
                float pr = expf(l[(long)d * HW] - mx) / den;
                dep += pr * depth_values[d];
                if (pr > pm) pm = pr;
            }
            depth[(long)yo * Wo + xo] = dep;
            prob[(long)yo * Wo + xo] = pm;
        }
    }
}
