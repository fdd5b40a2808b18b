/*
 * estd_hip.h -- C ABI of libestd_hip.so: the MI355X (gfx950) kernels of ESTDepth's
 * plane-sweep + EST-transformer hot path.
 *
 * The reference (xxlong0/ESTDepth) has no native layer at all: every op below replaces a
 * composition of ATen calls inside the Python functions cited per entry point.  The ABI is
 * therefore ours: plain device pointers + sizes + a hipStream_t, int status return (0 = OK,
 * negative = estd_status), no exceptions, no torch types.  All pointers are DEVICE pointers to
 * fp32 unless stated; every call only enqueues work on `stream` (no hidden synchronisation).
 *
 * Internal volume layouts (private to the library + its host wrapper):
 *   vol32  : [N][D][H][W][32]  channels-last cost / feature volumes
 *   kv     : [D][H][W][32]     value = channels 0..15, key = channels 16..31
 *   scalar : [N][D][H][W]      1-channel volumes (semantic plane scores, logits, 33rd channel)
 */
#ifndef ESTD_HIP_H
#define ESTD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* estd_stream_t; /* hipStream_t */

enum estd_status {
    ESTD_OK = 0,
    ESTD_ERR_ARG = -1,      /* null pointer / non-positive size / unsupported channel count */
    ESTD_ERR_LAUNCH = -2,   /* hipLaunch / hipGetLastError failure */
    ESTD_ERR_UNSUPPORTED = -3
};

enum estd_act { ESTD_ACT_NONE = 0, ESTD_ACT_RELU = 1, ESTD_ACT_TANH = 2 };

/* upper bound on the views + memory volumes one target attends to in estd_warp_attention (the reference loops over
 * any number, hybrid_depth_decoder.py:229-246; 16 covers Joint mode up to seq_len 17 / 15 targets + 2 memories) */
#define ESTD_MAX_ATTENTION_SOURCES 16

int estd_version(void);
/* launches the empty kernel `estd_mark_kernel` so a rocprofv3 kernel trace can be cut to a timed region */
int estd_profile_mark(int id, estd_stream_t stream);
const char* estd_status_string(int status);
/* Compute units the persistent convolution grids leave free (0..128, rounded up to a multiple of 8 = one per XCD; default 0, or
 * the environment variable ESTD_RESERVED_CUS read at load time).  The convolution kernels launch one or two resident
 * workgroups per CU with STATIC tile ranges: a concurrent kernel that holds even a few CUs (an RCCL collective overlapped
 * with the step) would push the workgroups that no longer fit behind all the others and double the launch's duration.
 * Multi-GPU hosts reserve 8 (bench.py does for N > 1).  Process-wide; takes effect for launches (and graph captures) made
 * afterwards.  Returns the value in effect. */
int estd_set_reserved_cus(int n);
int estd_get_reserved_cus(void);

/* ---- camera algebra on device (tiny fp64 kernels; keeps the forward free of host syncs) ------
 * Replaces the torch.inverse / matmul calls at hybrid_models/model_hybrid.py:74-88,
 * utils/homo_utils.py:469-471, hybrid_models/hybrid_depth_decoder.py:235 and
 * utils/homo_utils.py:51,:258.  */

/* proj12 = rows 0..2 of (src_proj @ inverse(ref_proj)) as rot[9] | trans[3]  (homo_utils.py:469-471) */
int estd_cam_pair_proj(const float* src_proj16, const float* ref_proj16, float* proj12, estd_stream_t stream);

/* Same, starting from camera-to-world poses and the 1/4-scale intrinsics, i.e. including
 * model_hybrid.py:74-88 (extrinsic = inverse(pose); proj[:3,:4] = K @ extrinsic[:3,:4]). */
int estd_cam_sweep_proj(const float* ref_pose16, const float* src_pose16, const float* intr9,
                        float* proj12, estd_stream_t stream);

/* mats30 = inverse(K)[9] | inverse(pose_j @ inverse(pose_i))[rows 0..2 = 12] | K[9]
 * (hybrid_depth_decoder.py:235 then homo_utils.py:51,:258).  If pose_i == NULL, pose_j is taken
 * as the already-formed relative pose (the level-1 warp_volume() call). */
int estd_cam_volume_mats(const float* pose_j16, const float* pose_i16, const float* intr9,
                         float* mats30, estd_stream_t stream);

/* ---- plane sweep ----------------------------------------------------------------------------- */

/* Level-1 operator utils/homo_utils.py:458-504 homo_warping(): src [C][H][W] -> out [C][D][H][W]. */
int estd_homo_warping(const float* src_chw, const float* proj12, const float* depth_values,
                      float* out_cdhw, int C, int D, int H, int W, estd_stream_t stream);

/* Same operator with PER-PIXEL depth hypotheses depth_dhw [D][H][W] (the reference's second accepted shape of depth_values,
 * utils/homo_utils.py:462 "[B, Ndepth] o [B, Ndepth, H, W]", :480-481). */
int estd_homo_warping_px(const float* src_chw, const float* proj12, const float* depth_dhw,
                         float* out_cdhw, int C, int D, int H, int W, estd_stream_t stream);

/* 1x1 channel mix of a 2D feature map, NCHW in -> HWC out: out[p][o] = sum_c w[o][c]*in[c][p] + b[o].
 * Used to push pre0 (model_hybrid.py:58,:93-94: 1x1x1 conv 64->32 + BN over cat[ref, warped]) in
 * front of the warp: pre0(cat[ref,warp(src)]) = mix_ref(ref)+shift + warp(mix_src(src)). Cin,Cout<=64 */
int estd_mix1x1_chw_to_hwc(const float* in_chw, const float* w, const float* bias, float* out_hwc,
                           int Cin, int Cout, int HW, estd_stream_t stream);

/* Fused homo_warping + pre0: out[d][y][x][:] = ref_mix[y][x][:] + bilinear(src_mix)(d,y,x)  (32 ch). */
int estd_homo_warp_costvol(const float* src_mix_hwc, const float* ref_mix_hwc, const float* proj12,
                           const float* depth_values, float* out_vol32, int D, int H, int W,
                           estd_stream_t stream);

/* ---- 3x3x3 convolution, implicit GEMM on fp32 MFMA (v_mfma_f32_16x16x4_f32) -------------------
 * Replaces networks/layers_op.py:16-39 (Conv3d bias=False + BatchNorm3d eval + ReLU/Tanh) as used at
 * model_hybrid.py:59-60,:95 and hybrid_depth_decoder.py:84-112,:190-200,:256,:377, and the two biased
 * Conv3d of transformer/epipolar_transformer.py:21,:26.  Weights are pre-packed by the host wrapper
 * (estdepth_amd/packing.py documents the fragment order). */
typedef struct estd_conv3d_desc {
    int N, D, H, W;
    int cin_main;             /* 16 or 32 channels read from in_main */
    int in_stride;            /* floats between consecutive voxels of in_main (>= cin_main) */
    int n_tiles;              /* output 16-channel tiles: 1 or 2; 3 = 32 channels on MFMA + a 33rd channel on the VALU */
    const float* in_main;     /* [N][D][H][W][in_stride] */
    const float* in_extra;    /* [N][D][H][W] scalar input channel, or NULL */
    const float* w_main;      /* packed, see packing.py */
    const float* w_extra;     /* packed extra-channel taps, or NULL */
    const float* w_xout;      /* n_tiles == 3 only: output channel 32's weights in the packing of the entry point called (packing.py:
                               * pack_xout / pack_conv3d_wino_xout / pack_conv3d_wino2_xout), else NULL */
    const float* scale;       /* [n_out] folded BN scale per output channel (n_out = 16, 32 or 33) */
    const float* shift;       /* [n_out] folded BN shift / conv bias */
    int act_a, act_b, act_split;  /* channels < act_split use act_a, others act_b */
    /* main output (channels-last, out_stride floats per voxel, may alias a sub-range of a wider tensor) */
    float* out_main;          /* NULL when only the head output is wanted */
    int out_stride;
    int out_channels;         /* 16 or 32 real channels written to out_main */
    const float* residual;    /* vol with the layout of out_main added after the activation, or NULL */
    const float* residual2;   /* a second such volume (sum over two source views before the linear pre2), or NULL */
    float out_scale;          /* applied after the residual add (1.0 = none) */
    int accumulate;           /* 1: out_main += result (running sum over source views) */
    float* out_extra;         /* scalar volume receiving output channel 32 (n_tiles == 3), or NULL */
    /* fused 1x1x1 head (stereo_head*.1, hybrid_depth_decoder.py:106,:111): logit = sum_c head_w[c]*y[c] + head_b */
    const float* head_w;      /* [16] or NULL (requires n_tiles == 1) */
    const float* head_b;      /* device pointer to 1 float */
    float* out_head;          /* scalar volume [N][D][H][W] */
    /* GroupNorm(1 group) statistics of the raw outputs, per 16-channel group: partial sums per block,
     * double[grid][2 groups][2] = {sum, sumsq}; finalised by estd_groupnorm_finalize. */
    double* stats_partials;   /* or NULL */
    /* estd_conv3d_k3_split only: the 32->32 weights split into three bf16 pieces,
     * uint16 [27 taps][3 pieces][2 n-tiles][64 lanes][8] (packing.py::pack_conv3d_split), else NULL */
    const void* w_split;
    /* estd_conv3d_k3_wino only: the 32->32 filters in depth-Winograd F(2,3) form, float32
     * [37 taps (4 x 9 + 1 pad)][2 channel halves][2 quads][64 lanes][4] (packing.py::pack_conv3d_wino), else NULL */
    const float* w_wino;
    /* estd_conv3d_k3_wino2 only: the 32->32 filters with depth AND row axis in Winograd F(2,3) form, float32
     * [48 taps = (3 sd + kw) * 4 + sh][2 channel halves][2 quads][64 lanes][4] (packing.py::pack_conv3d_wino2), else NULL */
    const float* w_wino2;
    /* estd_conv3d_k3_wino2, 32 -> 16 instance only (the ConvGRU's output convolution, transformer/epipolar_transformer.py:51-52): the reset
     * gate applied in the convolution's plane loads instead of in a pass of its own (estd_gru_reset_apply): input channels 16..31 (h) are
     * multiplied by sigmoid(GroupNorm(r)) with r = channels 0..15 of gate_r [N][D][H][W][32] (the gate convolution's raw output),
     * gate_stats = {mean_r, rstd_r, ..} as estd_groupnorm_finalize writes them, gate_gamma / gate_beta [16] the affine of
     * reset_gate_norm (:44,:46).  All four NULL = no gate. */
    const float* gate_r;
    const float* gate_stats;
    const float* gate_gamma;
    const float* gate_beta;
} estd_conv3d_desc;

int estd_conv3d_k3(const estd_conv3d_desc* desc, estd_stream_t stream);
#ifdef ESTD_BUILD_AB   /* superseded A/B kernel: built and exported only with ESTD_BUILD_AB=1 (estdepth_amd/build.py) */
/* Same operator for the plain 32->32 case (cin_main = 32, n_tiles = 2, no extra channel / head / 33rd output), with
 * every fp32 product evaluated as six bf16 MFMA products of exactly split operands (a = a1+a2+a3, b = b1+b2+b3,
 * fp32 accumulation; dropped terms <= 2^-26 |ab|): fp32-level error at 96 instead of 256 matrix-pipe cycles per
 * 16x16x32 block.  Reads w_split instead of w_main.  ESTD_ERR_UNSUPPORTED for any other shape. */
int estd_conv3d_k3_split(const estd_conv3d_desc* desc, estd_stream_t stream);
#endif
#ifdef ESTD_BUILD_AB   /* superseded A/B kernel: built and exported only with ESTD_BUILD_AB=1 (estdepth_amd/build.py) */
/* Same operator for the plain 32->32 instance (cin_main = 32, n_tiles = 2, no extra channel / head / 33rd output; BN, ReLU,
 * residuals, scale, accumulation and GroupNorm partials as estd_conv3d_k3) with the depth axis in Winograd F(2,3) form:
 * two output planes from four transformed input planes, 36 instead of 54 tap products, every product an fp32 MFMA with
 * fp32 accumulation (csrc/conv3d_wino.hip).  Reads w_wino instead of w_main.  ESTD_ERR_UNSUPPORTED for any other shape. */
int estd_conv3d_k3_wino(const estd_conv3d_desc* desc, estd_stream_t stream);
#endif
/* Same operator with the depth AND the image-row axis in Winograd form, F(2x2, 3x3): 2 x 2 outputs (two planes, two rows) from a
 * 4 x 4 transformed input patch, 48 tap products per 4 outputs = 0.444 of the direct kernel's MFMA work (csrc/conv3d_wino2.hip).
 * Instances: cin_main = 32 with n_tiles = 2 (32 -> 32; with in_extra + w_extra the 33 -> 32 key|value form; every epilogue feature of
 * estd_conv3d_k3_wino -- GroupNorm partials not together with in_extra) and n_tiles = 1 (32 -> 16, the ConvGRU output convolution;
 * no in_extra), and cin_main = 16 with n_tiles = 1, head_w / head_b / out_head set and out_main = NULL (16 -> 16 + the fused 1x1x1
 * head, only the logit volume is written: stereo_head0 / stereo_head1, hybrid_depth_decoder.py:96-112; csrc/conv3d_wino2_c16.hip,
 * weights float32 [48 taps][64 lanes][4], packing.py::pack_conv3d_wino2_c16; no residuals / statistics / tanh).
 * n_tiles = 3 with in_extra, w_extra, out_extra and w_xout (packing.py::pack_conv3d_wino2_xout): the 33 -> 33 instance (dres2,
 * hybrid_depth_decoder.py:106) -- output channel 32 on the VALU from the fragments the MFMAs consume; no read-back streams, no statistics.
 * Reads w_wino2 (packing.py::pack_conv3d_wino2).  ESTD_ERR_UNSUPPORTED for any other shape. */
int estd_conv3d_k3_wino2(const estd_conv3d_desc* desc, estd_stream_t stream);
#ifdef ESTD_BUILD_AB   /* superseded A/B kernel: built and exported only with ESTD_BUILD_AB=1 (estdepth_amd/build.py) */
/* The 32 -> 32 instance of estd_conv3d_k3_wino2 (cin_main = 32, n_tiles = 2, no in_extra / out_extra / head; BN, ReLU, residuals, scale,
 * running sum, GroupNorm partials -- the latter without read-back streams; no tanh) on the operand-reuse kernel csrc/conv3d_wino2x.hip:
 * same F(2x2, 3x3) arithmetic, one 512-register wave per SIMD on v_mfma_f32_32x32x2_f32, both transforms in front of the LDS, wave-private
 * operand blocks.  Reads desc->w_wino2, which must then hold the packing of packing.py::pack_conv3d_wino2x: float32
 * [4 sd][3 kw][2 chunks][2 q][4 sh][64 lanes][4].  ESTD_ERR_UNSUPPORTED for any other shape (callers fall back to estd_conv3d_k3_wino2). */
int estd_conv3d_k3_wino2x(const estd_conv3d_desc* desc, estd_stream_t stream);
#endif
/* The 32-output-channel instances of estd_conv3d_k3_wino2 (cin_main = 32, n_tiles = 2, no out_extra / head / gate) with ALL THREE axes in
 * Winograd F(2,3) form -- F(2x2x2, 3x3x3), 8/27 of the direct products (csrc/conv3d_wino3.hip; the default for these launches).  Instances:
 *   * 32 -> 32: BN, activation (ReLU / tanh / split), residuals, scale, running sum (the read-back epilogues);
 *   * 32 -> 32 + stats_partials (GroupNorm partial sums; the ConvGRU gate convolution) -- WITHOUT read-back streams;
 *   * 33 -> 32: in_extra + w_extra, the latter in packing.py::pack_conv3d_wino3_extra form (the key || value convolution,
 *     hybrid_depth_decoder.py:198-199) -- without read-back streams and without stats_partials.
 * Reads desc->w_wino2, which must then hold the packing of packing.py::pack_conv3d_wino3: float32
 * [64 blocks ((4 sd + sh) * 2 + cc) * 2 + hh][2 halves][2 tap pairs][64 lanes][4].  ESTD_ERR_UNSUPPORTED for any other shape (callers fall back to
 * estd_conv3d_k3_wino2). */
int estd_conv3d_k3_wino3(const estd_conv3d_desc* desc, estd_stream_t stream);
/* Output channel 32 ALONE of the 33 -> 33 instance (n_tiles = 3: dres2, hybrid_depth_decoder.py:93-95,:196): out_extra = act(conv(in_main[32] | in_extra
 * -> 1 channel) * scale[32] + shift[32]) -- the pass that lets the 32 main output channels of that layer run on estd_conv3d_k3_wino3's 33 -> 32
 * instance (csrc/conv3d_xout.hip: the 27 taps as the matrix core's rows, a shifted sum of scalars behind it).  Reads desc->w_xout in the packing of
 * packing.py::pack_conv3d_xout_taps (float32 [2][2][64][4] + [2][64]); cin_main = 32, in_extra, scale / shift with 33 entries and out_extra required;
 * out_main and every other output field are ignored.  ESTD_ERR_UNSUPPORTED for any other shape. */
int estd_conv3d_k3_xout(const estd_conv3d_desc* desc, estd_stream_t stream);
/* number of thread blocks estd_conv3d_k3 launches for a volume (size of stats_partials / 4 doubles) */
int estd_conv3d_k3_grid(int N, int D, int H, int W);

/* ---- 3x3 2D convolution on NHWC maps (PSMNet matching features) ------------------------------
 * Replaces Conv2d(3x3, stride 1, dilation 1|2, bias=False) + BatchNorm2d(eval) [+ReLU] [+residual add] of
 * networks/layers_op.py:10-27 as used by networks/psm_submodule.py:14-37,43-60,112-114.  Cin, Cout multiples of 32.
 * Weights packed by estdepth_amd/packing.py::pack_conv2d in groups of 16*group_tiles output channels. */
typedef struct estd_conv2d_desc {
    int N, H, W;
    int cin, cout;
    int dilation;             /* 1 or 2 (padding = dilation) */
    int group_tiles;          /* 2 or 4: output channels per work item = 16*group_tiles */
    const float* in;          /* [N][H][W][cin] */
    const float* w;           /* packed [cout/(16*group_tiles)][cin/32][10 taps (9 + 1 pad)][2*group_tiles][64][4] */
    const float* scale;       /* [cout] folded BN scale */
    const float* shift;       /* [cout] folded BN shift */
    int relu_before_residual; /* conv-bn-relu */
    int relu_after_residual;  /* relu(conv-bn + residual) */
    const float* residual;    /* [N][H][W][cout] or NULL */
    float* out;               /* [N][H][W][cout] */
    /* estd_conv2d_k3_split only: int16 [cout/32][cin/32][9 taps][4096] bf16 split weights (packing.py::pack_conv2d_split) */
    const void* w_split;
    /* estd_conv2d_k3_wino only: the filters with the ROW taps in Winograd F(2,3) form, float32
     * [cout/(16*group_tiles)][cin/32][13 taps (4 x 3 + 1 pad)][2*group_tiles][64][4] (packing.py::pack_conv2d_wino) */
    const float* w_wino;
} estd_conv2d_desc;

int estd_conv2d_k3(const estd_conv2d_desc* desc, estd_stream_t stream);
#ifdef ESTD_BUILD_AB   /* superseded A/B kernel: built and exported only with ESTD_BUILD_AB=1 (estdepth_amd/build.py) */
/* Same operator with the row axis in Winograd F(2,3) form: two output rows from four transformed input rows, 12 instead of
 * 18 tap products, every product an fp32 MFMA with fp32 accumulation (csrc/conv2d_wino.hip).  Reads w_wino instead of w. */
int estd_conv2d_k3_wino(const estd_conv2d_desc* desc, estd_stream_t stream);
#endif
/* Same operator (dilation 1 | 2; group_tiles ignored: 32 output channels per work item) with BOTH image axes in Winograd form,
 * F(2x2, 3x3): 2 x 2 output pixels from a 4 x 4 transformed input patch, 16 instead of 36 tap products = 0.444 of the direct kernel's
 * MFMA work (csrc/conv2d_wino2.hip).  Reads desc->w_wino, which must then hold the F(2x2, 3x3) packing: float32
 * [cout/32][cin/32][8 steps][4][2][64][4] (packing.py::pack_conv2d_wino2). */
int estd_conv2d_k3_wino2(const estd_conv2d_desc* desc, estd_stream_t stream);
#ifdef ESTD_BUILD_AB   /* superseded A/B kernel: built and exported only with ESTD_BUILD_AB=1 (estdepth_amd/build.py) */
/* Same operator (group_tiles ignored: 32 output channels per work item) with every fp32 product as six
 * bf16 MFMA products of exactly 3-way split operands, fp32 accumulation (see estd_conv3d_k3_split). */
int estd_conv2d_k3_split(const estd_conv2d_desc* desc, estd_stream_t stream);
#endif

/* mean/rstd from the partials: stats_out = {mean_g0, rstd_g0, mean_g1, rstd_g1}; count = 16*D*H*W per group
 * (transformer/epipolar_transformer.py:22-23,:27 GroupNorm(1, 16, eps=1e-5)). */
int estd_groupnorm_finalize(const double* partials, int n_blocks, double count, float eps, float* stats_out4,
                            estd_stream_t stream);

/* ---- soft-argmin ------------------------------------------------------------------------------
 * hybrid_depth_decoder.py:33-38 depthlayer() applied to F.interpolate(logits, scale_factor=s) (nearest;
 * :202-204,:259-260,:359-361,:379-381), computed at low resolution and replicated s x s.
 * logits [N][D][H][W] -> depth, prob [N][s*H][s*W]. */
int estd_softargmin_up(const float* logits, const float* depth_values, float* depth, float* prob,
                       int N, int D, int H, int W, int s, estd_stream_t stream);

/* ---- EST transformer --------------------------------------------------------------------------*/

/* Level-1 operator utils/homo_utils.py:240-279 warp_volume() (zeros padding, trilinear):
 * vol [C][D][H][W] -> out [C][D][H][W]; depth_values [D] are the plane depths (the reference passes
 * depth_values.repeat(H,W), hybrid_depth_decoder.py:237). */
int estd_warp_volume(const float* vol_cdhw, const float* mats30, const float* depth_values,
                     float depth_min, float depth_interval, float* out_cdhw,
                     int C, int D, int H, int W, estd_stream_t stream);

/* Every branch of the reference's warp_volume() signature (utils/homo_utils.py:240-279): depth per plane [D] or per VOXEL
 * [D][H*W] (:246,:253), depth or disparity planes for the z normalisation (:187-190), padding_mode 'zeros' or 'border' -- the
 * latter samples the volume whose outermost voxel layer is replaced by padding_value (:271-274, _set_vol_border :305-319). */
typedef struct estd_warp_volume_opts {
    int depth_per_voxel;      /* 0: depth[D], 1: depth[D][H*W] */
    int use_disp;             /* 0: depth planes (depth_min, depth_interval), 1: disparity planes (disp_min, disp_interval) */
    int border;               /* 0: padding_mode='zeros', 1: padding_mode='border' with padding_value */
    float depth_min, depth_interval;
    float disp_min, disp_interval;
    float padding_value;
} estd_warp_volume_opts;
int estd_warp_volume_ex(const float* vol_cdhw, const float* mats30, const float* depth, const estd_warp_volume_opts* opts,
                        float* out_cdhw, int C, int D, int H, int W, estd_stream_t stream);

/* Fused warp_volume(K_j), warp_volume(V_j) for all sources j + epipolar attention
 * (hybrid_depth_decoder.py:233-246 + transformer/epipolar_transformer.py:62-73):
 *   xh[vox][0:16] = V_t ; xh[vox][16:32] = h = mean_j( softmax_j(K_t . warp(K_j)) * warp(V_j) ).
 * kv_src: HOST array of n_src device pointers to kv volumes (copied into the launch arguments);
 * mats_dev: device [n_src][30] from estd_cam_volume_mats.  n_src in 1..ESTD_MAX_ATTENTION_SOURCES
 * (more: ESTD_ERR_UNSUPPORTED).  Global gather of the eight corner records per source (LDS staging of the source boxes
 * was measured slower and dropped, csrc/est_fusion.hip), 2x4x8 target bricks in XCD-contiguous order, 4 lanes per target
 * voxel; instances specialised for 1..4 sources and generic ones for up to 8 / 16 (per-source correlation held in registers,
 * max-subtracted softmax as the reference's). */
int estd_warp_attention(const float* kv_target, const float* const* kv_src, const float* mats_dev,
                        int n_src, const float* depth_values, float depth_min, float depth_interval,
                        float* xh_out, int D, int H, int W, estd_stream_t stream);

/* Attention over already-warped kv volumes (the level-1 EpipolarTransformer.forward signature,
 * transformer/epipolar_transformer.py:56-73): same output as estd_warp_attention without the gather; n_src in
 * 1..ESTD_MAX_ATTENTION_SOURCES like it (more: ESTD_ERR_UNSUPPORTED). */
int estd_attention_prewarped(const float* kv_target, const float* const* kv_src, int n_src,
                             float* xh_out, int64_t n_vox, estd_stream_t stream);

/* xrh[vox] = [ x , sigmoid(GN(r_raw)) * h ]  (epipolar_transformer.py:44,:46,:51):
 * xh = [x,h]; ru = raw gate conv output [r(0..15), u(16..31)];
 * stats4 from estd_groupnorm_finalize; gamma/beta = reset_gate_norm affine [16]. */
int estd_gru_reset_apply(const float* xh, const float* ru, const float* stats4, const float* gamma_r,
                         const float* beta_r, float* xrh, int64_t n_vox, estd_stream_t stream);

/* out_value[vox][0:16] (stride out_stride) = u*h + (1-u)*tanh(GN(o_raw)),  u = sigmoid(GN(u_raw))
 * (epipolar_transformer.py:45,:47,:53,:82-83). */
int estd_gru_blend(const float* xh, const float* ru, const float* o_raw, const float* stats_ru4,
                   const float* stats_o4, const float* gamma_u, const float* beta_u, const float* gamma_o,
                   const float* beta_o, float* out_value, int out_stride, int64_t n_vox, estd_stream_t stream);

/* ---- layout conversion at the API edge (reference tensors are NCDHW) -------------------------- */
/* [C][S] (channel planes, S = D*H*W) -> [S][dst_stride] at channel offset dst_off */
int estd_cdhw_to_vol(const float* src_cdhw, float* dst, int C, int64_t S, int dst_stride, int dst_off,
                     estd_stream_t stream);
int estd_vol_to_cdhw(const float* src, float* dst_cdhw, int C, int64_t S, int src_stride, int src_off,
                     estd_stream_t stream);

/* ---- fused inference BatchNorm2d (+ residual add) (+ ReLU) on NHWC maps, in place --------------------------------
 * x[p][c] = act(x[p][c] * scale[c] + shift[c] + residual[p][c]); replaces the BatchNorm2d -> (add) -> ReLU launches that
 * follow the library convolutions of the 2D backbones (resnet_encoder.py:43-49, psm_submodule.py:14-37,
 * hybrid_depth_decoder.py:17-30).  C multiple of 4; residual may be NULL. */
int estd_bn_act_nhwc(float* x, const float* scale, const float* shift, const float* residual, int relu,
                     int64_t n_pix, int C, estd_stream_t stream);

/* ---- PSMNet SPP tail (networks/psm_submodule.py:100-116): out[n][y][x] = cat(raw, skip, up(b[0]), .. up(b[nb-1])) ----------
 * raw [N][H][W][c_raw], skip [N][H][W][c_skip], b[k] [N][bh[k]][bw[k]][c_b] (NHWC), up = bilinear resize to HxW with
 * align_corners = False (F.upsample in the reference's torch version = F.interpolate(..., align_corners=False)).
 * One pass instead of nb upsample kernels + a 123 MB concatenation.  Channel counts multiples of 4, nb <= 4. */
int estd_spp_upsample_cat(const float* raw, int c_raw, const float* skip, int c_skip, const float* const* branches,
                          const int* bh, const int* bw, int nb, int c_b, float* out, int N, int H, int W,
                          estd_stream_t stream);

/* ---- 2D refinement tail of the decoder (hybrid_models/hybrid_depth_decoder.py:267-290 / :392-415), glue around its convolutions --
 * estd_planes_cat_nhwc:    torch.cat([a, relu?(b)], 1) of two NCHW stacks a [N][Ca][HW], b [N][Cb][HW] (:268
 *                          cat([semantic_vs, relu(all_fused_logits)])) written as the NHWC map [N][HW][Ca+Cb]; Ca+Cb <= 496.
 * estd_upsample2_cat_nhwc: torch.cat([upsample(x), skip], 1) (:269-272, :280-281): x [N][H/2][W/2][Cx] nearest x2 beside
 *                          skip [N][H][W][Cs] -> out [N][H][W][Cx+Cs] (NHWC; channel counts multiples of 4, H and W even).
 * estd_disp_head_nhwc:     depth_max * sigmoid(Conv2d(C, 1, 3, stride 1, padding 1, bias)(in)) (:274 dispconv_1, :279 dispconv_0):
 *                          in [N][H][W][C] NHWC, w [1][C][3][3], bias [1] (device), C = 16 | 32; out [N][1][upscale*H][upscale*W],
 *                          upscale = 1, or 2 = the F.interpolate(scale_factor=2) (nearest) of :274 fused in. */
/* 3x3 / stride 1 / padding 1 convolution to 16 channels + folded BatchNorm2d + ReLU on NHWC maps, the full-resolution ConvBlocks
 * of the decoder (hybrid_models/hybrid_depth_decoder.py:17-30 ConvBlock; :276 upconv_0_0, :277-278 upconv_0_1(upsample(x))):
 * in [N][Hin][Win][cin], cin = 16 | 32; upsample = 1: the convolution reads the nearest-x2 upsampled map (Hin = H/2, Win = W/2,
 * :11-14) without materialising it; out [N][H][W][16].  w_packed: packing.pack_conv2d_to16 ([9 taps][cin/16][64 lanes][4]). */
/* The small convolutions of the PSM extractor outside the tiled 3x3 / stride-1 kernels (networks/psm_submodule.py): 3x3 stride 2
 * (:52 layer2[0].conv1), 1x1 stride 1 | 2 (:78-83 downsample, :100-110 SPP branches, :72-74 lastconv's 1x1) + folded BatchNorm2d
 * (scale 1 / shift 0 where the reference has none) [+ ReLU] on NHWC maps: in [N][Hin][Win][cin] -> out [N][Ho][Wo][cout], padding
 * ksize / 2.  Instances: (cin, ksize, stride) = (32,3,2), (32,1,2), (32,1,1), (64,1,1), (128,1,1); cout a multiple of 16;
 * w_packed: packing.pack_conv2d_small ([cout/16][ksize^2 taps][cin/16][64 lanes][4]).  Other shapes: ESTD_ERR_UNSUPPORTED. */
int estd_conv2d_small_nhwc(const float* in, const float* w_packed, const float* scale, const float* shift, float* out, int N, int Hin,
                           int Win, int cin, int cout, int ksize, int stride, int relu, estd_stream_t stream);
/* 1x1 convolution (stride 1 | 2, no padding) of an NHWC map + folded BatchNorm2d(eval) [+ residual] [+ ReLU] in ONE launch: the
 * bottleneck convolutions of the semantic branch's ResNet (hybrid_models/resnet_encoder.py:40-51 over torchvision's Bottleneck:
 * conv1 + bn1 + relu, conv3 + bn3 + shortcut + relu, downsample[0] + downsample[1]).  out = max(in . w^T * scale + shift
 * (+ residual), relu ? 0 : -inf).  cin a multiple of 16, cout a multiple of 32; other shapes: ESTD_ERR_UNSUPPORTED
 * (csrc/conv1x1.hip). */
typedef struct estd_conv1x1_desc {
    int N, H, W;              /* input map */
    int cin, cout;
    int stride;               /* 1 or 2: output pixel (y, x) reads input pixel (stride*y, stride*x); Ho = (H-1)/stride+1 */
    int relu;                 /* 1: ReLU after the (residual) add */
    const float* in;          /* [N][H][W][cin] */
    const float* w;           /* [cout][cin]: the Conv2d weight [cout, cin, 1, 1] as it lies in memory */
    const float* scale;       /* [cout] folded BN scale, or NULL (= 1) */
    const float* shift;       /* [cout] folded BN shift / bias, or NULL (= 0) */
    const float* residual;    /* [N][Ho][Wo][cout] added before the ReLU, or NULL */
    float* out;               /* [N][Ho][Wo][cout] */
} estd_conv1x1_desc;
int estd_conv1x1_nhwc(const estd_conv1x1_desc* desc, estd_stream_t stream);
/* k x k convolution (k = 1 | 3 | 5, stride 1 | 2, zero padding pad) of an NHWC map + folded BatchNorm2d(eval) [+ residual]
 * [+ ReLU] in ONE launch: the stride-2 3x3 convolutions of the semantic ResNet's layer2..4 (hybrid_models/resnet_encoder.py:40-51
 * over torchvision's Bottleneck.conv2 / BasicBlock.conv1 + bn + relu) and the 3x3 convolutions on maps too small for the tiled
 * Winograd kernels (hybrid_models/hybrid_depth_decoder.py:17-30 ConvBlock on the 1/32 map).  cin a multiple of 16, cout a multiple
 * of 32; Ho = (H + 2 pad - ksize) / stride + 1.  w: packing.pack_conv2d_taps = [ksize*ksize taps][cout][cin].  Other shapes:
 * ESTD_ERR_UNSUPPORTED (csrc/conv2d_taps.hip). */
typedef struct estd_conv2d_taps_desc {
    int N, H, W;              /* input map */
    int cin, cout;
    int ksize, stride, pad;
    int relu;                 /* 1: ReLU after the (residual) add */
    const float* in;          /* [N][H][W][cin] */
    const float* w;           /* [ksize*ksize][cout][cin] */
    const float* scale;       /* [cout] folded BN scale, or NULL (= 1) */
    const float* shift;       /* [cout] folded BN shift / bias, or NULL (= 0) */
    const float* residual;    /* [N][Ho][Wo][cout] added before the ReLU, or NULL */
    float* out;               /* [N][Ho][Wo][cout] */
} estd_conv2d_taps_desc;
int estd_conv2d_taps_nhwc(const estd_conv2d_taps_desc* desc, estd_stream_t stream);
/* first layer of the semantic ResNet (torchvision conv1 = Conv2d(3, 64, 7, stride 2, padding 3) + bn1 + relu,
 * hybrid_models/resnet_encoder.py:42-44): in [N][H][W][3] NHWC -> out [N][Ho][Wo][64] NHWC, Ho = (H-1)/2 + 1, Wo = (W-1)/2 + 1.
 * w_packed: packing.pack_stem7x7 ([7 rows][6 k-steps][4 channel tiles][64 lanes]); scale / shift [64] = folded BatchNorm2d. */
int estd_stem7x7s2_nhwc(const float* in, const float* w_packed, const float* scale, const float* shift, float* out, int N, int H,
                        int W, estd_stream_t stream);
/* MaxPool2d(3, stride 2, padding 1) of an NHWC map (torchvision ResNet.maxpool, resnet_encoder.py:45): in [N][H][W][C] ->
 * out [N][(H-1)/2+1][(W-1)/2+1][C]; C a multiple of 4; a NaN in a window is the window's result (ATen). */
int estd_maxpool3x3s2_nhwc(const float* in, float* out, int N, int H, int W, int C, estd_stream_t stream);
/* AvgPool2d(k, k) of an NHWC map (networks/psm_submodule.py:56-70, the SPP branches): in [N][H][W][C] -> out [N][H/k][W/k][C];
 * C a multiple of 4; window sum in row-major order, then one division by k*k (ATen's order). */
int estd_avgpool_nhwc(const float* in, float* out, int N, int H, int W, int C, int k, estd_stream_t stream);
int estd_conv2d_k3_to16_nhwc(const float* in, const float* w_packed, const float* scale, const float* shift, float* out, int N,
                             int H, int W, int cin, int upsample, estd_stream_t stream);
/* image normalisation of DepthNetHybrid.forward (hybrid_models/model_hybrid.py:119: imgs = 2 * (imgs / 255.) - 1.):
 * in [N][3][HW] planes (0..255) -> out [N][HW][3] NHWC records; the same three fp32 roundings as the reference's three ops. */
int estd_normalise_nhwc(const float* in, float* out, int N, int64_t HW, estd_stream_t stream);
/* first layer of the PSM matching-feature extractor (networks/psm_submodule.py:47 convbn(3, 32, 3, 2, 1, 1) + ReLU, :14-22):
 * in [N][H][W][3] NHWC, w [32][3][3][3] (Conv2d layout), scale/shift [32] = folded BatchNorm2d -> out [N][Ho][Wo][32] NHWC,
 * Ho = (H-1)/2 + 1, Wo = (W-1)/2 + 1 (kernel 3, stride 2, zero padding 1). */
int estd_stem3x3s2_nhwc(const float* in, const float* w, const float* scale, const float* shift, float* out, int N, int H, int W,
                        estd_stream_t stream);
int estd_planes_cat_nhwc(const float* a, int Ca, const float* b, int Cb, int relu_b, float* out, int N, int64_t HW,
                         estd_stream_t stream);
/* [N][HW][C] NHWC records -> [N][C][HW] planes: the 2D decoder's plane scores (hybrid_depth_decoder.py:162-184, the last ConvBlock's
 * D-channel NHWC map) as the scalar volumes [T][D][H][W] the 3D path reads (dres2's 33rd input channel, :268's concatenation). */
int estd_nhwc_to_planes(const float* in, int C, float* out, int N, int64_t HW, estd_stream_t stream);
int estd_upsample2_cat_nhwc(const float* x, int Cx, const float* skip, int Cs, float* out, int N, int H, int W,
                            estd_stream_t stream);
int estd_disp_head_nhwc(const float* in, const float* w, const float* bias, float depth_max, float* out, int N, int H, int W,
                        int C, int upscale, estd_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ESTD_HIP_H */
