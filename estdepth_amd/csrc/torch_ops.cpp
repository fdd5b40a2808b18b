// torch_ops.cpp -- PyTorch-ROCm custom-operator registration of the ESTDepth hot path (namespace `estdepth_hip`).
//
// Thin, allocation-and-validation-only wrappers over the torch-free C ABI of libestd_hip.so (include/estd_hip.h):
//   * TORCH_CHECK on device / dtype / contiguity / shapes  -> Python RuntimeError, the way ATen shape errors surface
//     from the reference today (SURVEY §8b "Errors");
//   * outputs allocated with at::empty on the input device (caching allocator), never retained;
//   * every kernel is enqueued on at::hip::getCurrentHIPStream() -- asynchronous, graph-capturable;
//   * a negative estd_status becomes TORCH_CHECK(false, estd_status_string(status)).
// The operators replace these reference call sites: utils/homo_utils.py:458 (homo_warping), :240 (warp_volume),
// hybrid_models/model_hybrid.py:62-102 (get_costvolume front), networks/layers_op.py:16-39 (convbn*_3d),
// hybrid_models/hybrid_depth_decoder.py:33 (depthlayer), :229-260 (temporal fusion),
// transformer/epipolar_transformer.py:31-83 (attention + ConvGRU).
// Built by estdepth_amd/build.py into estdepth_amd/lib/libestd_torch_ops.so; loaded with torch.ops.load_library().
#include <ATen/ATen.h>
#include <ATen/hip/HIPContext.h>
#include <ATen/DeviceGuard.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include <vector>

#include "estd_hip.h"

namespace {

using at::Tensor;
using OptTensor = std::optional<Tensor>;

inline estd_stream_t cur_stream() { return static_cast<estd_stream_t>(c10::hip::getCurrentHIPStream().stream()); }

// Every operator opens an OpScope on its first tensor argument: a device guard (the kernels are enqueued on the current stream
// of THAT tensor's device, not of whatever device happens to be current), and every further tensor checked through fptr() must
// live on the same device.
struct OpScope {
    static inline thread_local const c10::Device* cur = nullptr;
    c10::OptionalDeviceGuard guard;
    c10::Device dev;
    const c10::Device* prev;
    explicit OpScope(const Tensor& t) : guard(at::device_of(t)), dev(t.defined() ? t.device() : c10::Device(c10::kCPU)), prev(cur) { cur = &dev; }
    ~OpScope() { cur = prev; }
    OpScope(const OpScope&) = delete;
    OpScope& operator=(const OpScope&) = delete;
};

inline void check_status(int st, const char* what)
{
    TORCH_CHECK(st == ESTD_OK, what, " failed: ", estd_status_string(st), " (estd_status ", st, ")");
}

inline const float* fptr(const Tensor& t, const char* name, bool need_contiguous = true)
{
    TORCH_CHECK(t.defined(), name, " must be a tensor");
    TORCH_CHECK(t.is_cuda(), name, " must live on a ROCm device (estdepth_hip has no CPU path); got ", t.device());
    if (OpScope::cur) TORCH_CHECK(t.device() == *OpScope::cur, name, " is on ", t.device(), " but the operator runs on ", *OpScope::cur);
    TORCH_CHECK(t.scalar_type() == at::kFloat, name, " must be float32, got ", t.scalar_type());
    if (need_contiguous) TORCH_CHECK(t.is_contiguous(), name, " must be contiguous");
    return t.data_ptr<float>();
}
inline float* fptr_mut(const Tensor& t, const char* name, bool need_contiguous = true) { return const_cast<float*>(fptr(t, name, need_contiguous)); }
inline const float* opt_fptr(const OptTensor& t, const char* name, bool need_contiguous = true)
{
    return (t.has_value() && t->defined()) ? fptr(*t, name, need_contiguous) : nullptr;
}
inline Tensor new_f32(at::IntArrayRef shape, const Tensor& like) { return at::empty(shape, like.options().dtype(at::kFloat)); }

// ------------------------------------------------------------------------------------------------ camera algebra
Tensor cam_pair_proj(const Tensor& src_proj, const Tensor& ref_proj)
{
    const OpScope scope(src_proj);
    TORCH_CHECK(src_proj.numel() == 16 && ref_proj.numel() == 16, "cam_pair_proj: 4x4 projection matrices expected");
    Tensor out = new_f32({12}, src_proj);
    check_status(estd_cam_pair_proj(fptr(src_proj, "src_proj"), fptr(ref_proj, "ref_proj"), out.data_ptr<float>(), cur_stream()),
                 "estd_cam_pair_proj");
    return out;
}

Tensor cam_sweep_proj(const Tensor& ref_pose, const Tensor& src_pose, const Tensor& intr)
{
    const OpScope scope(ref_pose);
    TORCH_CHECK(ref_pose.numel() == 16 && src_pose.numel() == 16 && intr.numel() == 9, "cam_sweep_proj: 4x4 poses and 3x3 intrinsics expected");
    Tensor out = new_f32({12}, ref_pose);
    check_status(estd_cam_sweep_proj(fptr(ref_pose, "ref_pose"), fptr(src_pose, "src_pose"), fptr(intr, "cam_intr"),
                                     out.data_ptr<float>(), cur_stream()), "estd_cam_sweep_proj");
    return out;
}

void cam_volume_mats(const Tensor& pose_j, const OptTensor& pose_i, const Tensor& intr, Tensor out)
{
    const OpScope scope(pose_j);
    TORCH_CHECK(pose_j.numel() == 16 && intr.numel() == 9, "cam_volume_mats: 4x4 pose and 3x3 intrinsics expected");
    TORCH_CHECK(out.numel() == 30, "cam_volume_mats: out must hold 30 floats");
    check_status(estd_cam_volume_mats(fptr(pose_j, "pose_j"), opt_fptr(pose_i, "pose_i"), fptr(intr, "cam_intr"),
                                      fptr_mut(out, "mats30"), cur_stream()), "estd_cam_volume_mats");
}

// ------------------------------------------------------------------------------------------------ plane sweep
Tensor homo_warping(const Tensor& src_chw, const Tensor& proj12, const Tensor& depth_values, int64_t D)
{
    const OpScope scope(src_chw);
    TORCH_CHECK(src_chw.dim() == 3, "homo_warping: src_fea must be [C,H,W]");
    TORCH_CHECK(depth_values.numel() >= D && proj12.numel() == 12, "homo_warping: depth_values / proj12 size");
    const int64_t C = src_chw.size(0), H = src_chw.size(1), W = src_chw.size(2);
    Tensor out = new_f32({C, D, H, W}, src_chw);
    check_status(estd_homo_warping(fptr(src_chw, "src_fea"), fptr(proj12, "proj12"), fptr(depth_values, "depth_values"),
                                   out.data_ptr<float>(), (int)C, (int)D, (int)H, (int)W, cur_stream()), "estd_homo_warping");
    return out;
}

Tensor homo_warping_px(const Tensor& src_chw, const Tensor& proj12, const Tensor& depth_dhw)
{
    const OpScope scope(src_chw);
    TORCH_CHECK(src_chw.dim() == 3 && depth_dhw.dim() == 3, "homo_warping_px: src_fea [C,H,W], depth [D,H,W]");
    const int64_t C = src_chw.size(0), H = src_chw.size(1), W = src_chw.size(2), D = depth_dhw.size(0);
    TORCH_CHECK(depth_dhw.size(1) == H && depth_dhw.size(2) == W && proj12.numel() == 12, "homo_warping_px: depth / proj12 size");
    Tensor out = new_f32({C, D, H, W}, src_chw);
    check_status(estd_homo_warping_px(fptr(src_chw, "src_fea"), fptr(proj12, "proj12"), fptr(depth_dhw, "depth_values"),
                                      out.data_ptr<float>(), (int)C, (int)D, (int)H, (int)W, cur_stream()), "estd_homo_warping_px");
    return out;
}

Tensor mix1x1(const Tensor& in_chw, const Tensor& w, const OptTensor& bias)
{
    const OpScope scope(in_chw);
    TORCH_CHECK(in_chw.dim() == 3 && w.dim() == 2 && w.size(1) == in_chw.size(0), "mix1x1: in [Cin,H,W], w [Cout,Cin]");
    const int64_t Cin = in_chw.size(0), H = in_chw.size(1), W = in_chw.size(2), Cout = w.size(0);
    Tensor out = new_f32({H, W, Cout}, in_chw);
    check_status(estd_mix1x1_chw_to_hwc(fptr(in_chw, "feature"), fptr(w, "mix weight"), opt_fptr(bias, "mix bias"),
                                        out.data_ptr<float>(), (int)Cin, (int)Cout, (int)(H * W), cur_stream()), "estd_mix1x1_chw_to_hwc");
    return out;
}

void homo_warp_costvol(const Tensor& src_mix, const Tensor& ref_mix, const Tensor& proj12, const Tensor& depth_values,
                       int64_t D, Tensor out)
{
    const OpScope scope(src_mix);
    TORCH_CHECK(src_mix.dim() == 3 && src_mix.size(2) == 32 && ref_mix.sizes() == src_mix.sizes(),
                "homo_warp_costvol: src_mix / ref_mix must be [H,W,32] of one shape");
    TORCH_CHECK(depth_values.numel() >= D && D >= 1 && proj12.numel() == 12, "homo_warp_costvol: depth_values / proj12 size");
    const int64_t H = src_mix.size(0), W = src_mix.size(1);
    TORCH_CHECK(out.numel() == D * H * W * 32, "homo_warp_costvol: out must be [D,H,W,32]");
    check_status(estd_homo_warp_costvol(fptr(src_mix, "src_mix"), fptr(ref_mix, "ref_mix"), fptr(proj12, "proj12"),
                                        fptr(depth_values, "depth_values"), fptr_mut(out, "out"), (int)D, (int)H, (int)W, cur_stream()),
                 "estd_homo_warp_costvol");
}

// ------------------------------------------------------------------------------------------------ conv3d / conv2d
// Volumes are addressed as base pointers + strides (views into wider channels-last records are allowed), so only device
// and dtype are checked for them; the C ABI validates the stride / channel combinations.
void conv3d_k3(const Tensor& x, const OptTensor& x_extra, const OptTensor& w_main, const OptTensor& w_extra, const OptTensor& w_xout,
               const OptTensor& w_alt, const Tensor& scale, const Tensor& shift, at::IntArrayRef dims, int64_t cin_main,
               int64_t in_stride, int64_t n_tiles, int64_t act_a, int64_t act_b, int64_t act_split, const OptTensor& out,
               int64_t out_stride, int64_t out_channels, const OptTensor& residual, const OptTensor& residual2, double out_scale,
               bool accumulate, const OptTensor& out_extra, const OptTensor& head_w, const OptTensor& head_b, const OptTensor& out_head,
               const OptTensor& stats_partials, int64_t variant, const OptTensor& gate_r, const OptTensor& gate_stats,
               const OptTensor& gate_gamma, const OptTensor& gate_beta)
{
    const OpScope scope(x);
    TORCH_CHECK(dims.size() == 4, "conv3d_k3: dims = (N, D, H, W)");
    estd_conv3d_desc d{};
    d.N = (int)dims[0]; d.D = (int)dims[1]; d.H = (int)dims[2]; d.W = (int)dims[3];
    d.cin_main = (int)cin_main; d.in_stride = (int)in_stride; d.n_tiles = (int)n_tiles;
    const int64_t vox = dims[0] * dims[1] * dims[2] * dims[3];
    d.in_main = fptr(x, "conv3d input", false);
    d.in_extra = opt_fptr(x_extra, "conv3d extra input channel");
    if (d.in_extra) TORCH_CHECK(x_extra->numel() >= vox, "conv3d_k3: extra input channel smaller than N*D*H*W");
    d.w_main = opt_fptr(w_main, "packed weights");      // the direct kernel reads it (variant 0); the Winograd variants read w_alt
    TORCH_CHECK(variant != 0 || d.w_main, "conv3d_k3: the direct kernel (variant 0) needs w_main");
    d.w_extra = opt_fptr(w_extra, "packed extra-channel weights");
    d.w_xout = opt_fptr(w_xout, "packed 33rd-output weights");
    TORCH_CHECK((d.in_extra == nullptr) == (d.w_extra == nullptr), "conv3d plan/extra-channel mismatch");
    d.scale = fptr(scale, "scale"); d.shift = fptr(shift, "shift");
    d.act_a = (int)act_a; d.act_b = (int)act_b; d.act_split = (int)act_split;
    d.out_main = const_cast<float*>(opt_fptr(out, "conv3d output", false));
    d.out_stride = (int)out_stride; d.out_channels = (int)out_channels;
    d.residual = opt_fptr(residual, "residual", false);
    d.residual2 = opt_fptr(residual2, "residual2", false);
    d.out_scale = (float)out_scale; d.accumulate = accumulate ? 1 : 0;
    d.out_extra = const_cast<float*>(opt_fptr(out_extra, "33rd output channel"));
    d.head_w = opt_fptr(head_w, "head weight"); d.head_b = opt_fptr(head_b, "head bias");
    d.out_head = const_cast<float*>(opt_fptr(out_head, "head output"));
    if (!d.out_head) { d.head_w = nullptr; d.head_b = nullptr; }
    if (d.out_head) TORCH_CHECK(out_head->numel() >= vox, "conv3d_k3: head output smaller than N*D*H*W");
    d.stats_partials = nullptr;
    if (stats_partials.has_value() && stats_partials->defined()) {
        TORCH_CHECK(stats_partials->is_cuda() && stats_partials->scalar_type() == at::kDouble && stats_partials->is_contiguous(),
                    "conv3d_k3: stats_partials must be a contiguous float64 ROCm tensor");
        const int blocks = estd_conv3d_k3_grid(d.N, d.D, d.H, d.W);
        TORCH_CHECK(blocks > 0 && stats_partials->numel() >= (int64_t)blocks * 4, "conv3d_k3: stats_partials needs 4 doubles per tile");
        d.stats_partials = stats_partials->data_ptr<double>();
    }
    // reset gate folded into the loads of the 32 -> 16 two-axis Winograd instance (all four tensors or none)
    d.gate_r = opt_fptr(gate_r, "gate r volume", false);
    d.gate_stats = opt_fptr(gate_stats, "gate statistics");
    d.gate_gamma = opt_fptr(gate_gamma, "gate gamma");
    d.gate_beta = opt_fptr(gate_beta, "gate beta");
    if (d.gate_r) {
        TORCH_CHECK(d.gate_stats && d.gate_gamma && d.gate_beta, "conv3d_k3: the reset gate needs r, statistics, gamma and beta");
        TORCH_CHECK(gate_r->numel() >= vox * 32 && gate_stats->numel() >= 2 && gate_gamma->numel() >= 16 && gate_beta->numel() >= 16,
                    "conv3d_k3: reset-gate tensors too small");
        TORCH_CHECK(variant == 3, "conv3d_k3: the reset gate exists in the two-axis Winograd kernel's 32 -> 16 instance only");
    }
    // variant: 0 = direct fp32 MFMA; 1 = exact 3 x bf16 operand split (w_alt = split weights); 2 = fp32 MFMA with the depth
    // axis in Winograd F(2,3) form (w_alt = transformed filters); 3 = depth and row axis in Winograd form (w_extra / w_xout in that kernel's packing)
    if (variant != 0) TORCH_CHECK(w_alt.has_value() && w_alt->defined() && w_alt->is_cuda(), "conv3d_k3: this variant needs its packed weights");
    if (variant == 1 || variant == 2) {
#ifdef ESTD_BUILD_AB
        if (variant == 1) {
            d.w_split = w_alt->data_ptr();
            check_status(estd_conv3d_k3_split(&d, cur_stream()), "estd_conv3d_k3_split");
        } else {
            d.w_wino = fptr(*w_alt, "Winograd-packed weights");
            check_status(estd_conv3d_k3_wino(&d, cur_stream()), "estd_conv3d_k3_wino");
        }
#else
        TORCH_CHECK(false, "conv3d_k3: variant ", variant, " (bf16 operand split / depth-only Winograd) needs a library built with ESTD_BUILD_AB=1");
#endif
    } else if (variant == 3) {
        d.w_wino2 = fptr(*w_alt, "2-axis Winograd-packed weights");
        check_status(estd_conv3d_k3_wino2(&d, cur_stream()), "estd_conv3d_k3_wino2");
    } else if (variant == 4) {          // the 32 -> 32 instance on the operand-reuse kernel (w_alt = packing.pack_conv3d_wino2x)
#ifdef ESTD_BUILD_AB
        d.w_wino2 = fptr(*w_alt, "2-axis Winograd-packed weights (reuse form)");
        check_status(estd_conv3d_k3_wino2x(&d, cur_stream()), "estd_conv3d_k3_wino2x");
#else
        TORCH_CHECK(false, "conv3d_k3: variant 4 (operand-reuse two-axis Winograd kernel) needs a library built with ESTD_BUILD_AB=1");
#endif
    } else if (variant == 5) {          // the 32 -> 32 instance with all three axes in Winograd form (w_alt = packing.pack_conv3d_wino3)
        d.w_wino2 = fptr(*w_alt, "3-axis Winograd-packed weights");
        check_status(estd_conv3d_k3_wino3(&d, cur_stream()), "estd_conv3d_k3_wino3");
    } else if (variant == 6) {          // output channel 32 of the 33 -> 33 instance alone (w_alt = packing.pack_conv3d_xout_taps)
        d.w_xout = fptr(*w_alt, "tap-major weights of output channel 32");
        check_status(estd_conv3d_k3_xout(&d, cur_stream()), "estd_conv3d_k3_xout");
    } else {
        TORCH_CHECK(variant == 0, "conv3d_k3: unknown variant ", variant);
        check_status(estd_conv3d_k3(&d, cur_stream()), "estd_conv3d_k3");
    }
}

Tensor conv2d_k3(const Tensor& x_nhwc, const OptTensor& w, const OptTensor& w_alt, const Tensor& scale, const Tensor& shift,
                 int64_t cout, int64_t dilation, int64_t group_tiles, bool relu_before_residual, bool relu_after_residual,
                 const OptTensor& residual, int64_t variant)
{
    const OpScope scope(x_nhwc);
    TORCH_CHECK(x_nhwc.dim() == 4, "conv2d_k3: input must be [N,H,W,Cin] (NHWC, contiguous)");
    estd_conv2d_desc d{};
    d.N = (int)x_nhwc.size(0); d.H = (int)x_nhwc.size(1); d.W = (int)x_nhwc.size(2); d.cin = (int)x_nhwc.size(3);
    d.cout = (int)cout; d.dilation = (int)dilation; d.group_tiles = (int)group_tiles;
    d.in = fptr(x_nhwc, "conv2d input");
    d.w = opt_fptr(w, "packed conv2d weights");         // the direct kernel reads it (variant 0); the other variants read w_alt
    TORCH_CHECK(variant != 0 || d.w, "conv2d_k3: the direct kernel (variant 0) needs w");
    d.scale = fptr(scale, "scale"); d.shift = fptr(shift, "shift");
    TORCH_CHECK(scale.numel() == cout && shift.numel() == cout, "conv2d_k3: scale/shift must have Cout entries");
    d.relu_before_residual = relu_before_residual; d.relu_after_residual = relu_after_residual;
    Tensor out = new_f32({x_nhwc.size(0), x_nhwc.size(1), x_nhwc.size(2), cout}, x_nhwc);
    d.residual = opt_fptr(residual, "conv2d residual");
    if (d.residual) TORCH_CHECK(residual->sizes() == out.sizes(), "conv2d_k3: residual must be contiguous NHWC of the output shape");
    d.out = out.data_ptr<float>();
    // variant: 0 = direct fp32 MFMA; 1 = exact 3 x bf16 operand split; 2 = fp32 MFMA with the row axis in Winograd F(2,3) form
    if (variant != 0) TORCH_CHECK(w_alt.has_value() && w_alt->defined() && w_alt->is_cuda(), "conv2d_k3: this variant needs its packed weights");
    if (variant == 1 || variant == 2) {
#ifdef ESTD_BUILD_AB
        if (variant == 1) {
            d.w_split = w_alt->data_ptr();
            check_status(estd_conv2d_k3_split(&d, cur_stream()), "estd_conv2d_k3_split");
        } else {
            d.w_wino = fptr(*w_alt, "Winograd-packed weights");
            check_status(estd_conv2d_k3_wino(&d, cur_stream()), "estd_conv2d_k3_wino");
        }
#else
        TORCH_CHECK(false, "conv2d_k3: variant ", variant, " (bf16 operand split / row-only Winograd) needs a library built with ESTD_BUILD_AB=1");
#endif
    } else if (variant == 3) {          // both image axes in Winograd form (w_alt = packing.pack_conv2d_wino2)
        d.w_wino = fptr(*w_alt, "2-axis Winograd-packed weights");
        check_status(estd_conv2d_k3_wino2(&d, cur_stream()), "estd_conv2d_k3_wino2");
    } else {
        TORCH_CHECK(variant == 0, "conv2d_k3: unknown variant ", variant);
        check_status(estd_conv2d_k3(&d, cur_stream()), "estd_conv2d_k3");
    }
    return out;
}

Tensor groupnorm_finalize(const Tensor& partials, int64_t n_blocks, double count, double eps)
{
    const OpScope scope(partials);
    TORCH_CHECK(partials.is_cuda() && partials.scalar_type() == at::kDouble && partials.is_contiguous() && partials.numel() >= n_blocks * 4,
                "groupnorm_finalize: partials must be a contiguous float64 ROCm tensor with 4 doubles per block");
    Tensor out = at::empty({4}, partials.options().dtype(at::kFloat));
    check_status(estd_groupnorm_finalize(partials.data_ptr<double>(), (int)n_blocks, count, (float)eps, out.data_ptr<float>(), cur_stream()),
                 "estd_groupnorm_finalize");
    return out;
}

// ------------------------------------------------------------------------------------------------ soft-argmin
std::tuple<Tensor, Tensor> softargmin_up(const Tensor& logits, const Tensor& depth_values, int64_t scale)
{
    const OpScope scope(logits);
    TORCH_CHECK(logits.dim() == 4, "softargmin_up: logits must be [N,D,H,W]");
    const int64_t N = logits.size(0), D = logits.size(1), H = logits.size(2), W = logits.size(3);
    TORCH_CHECK(depth_values.numel() >= D, "softargmin_up: one depth value per plane expected");
    Tensor depth = new_f32({N, 1, H * scale, W * scale}, logits);
    Tensor prob = at::empty_like(depth);
    check_status(estd_softargmin_up(fptr(logits, "logits"), fptr(depth_values, "depth_values"), depth.data_ptr<float>(),
                                    prob.data_ptr<float>(), (int)N, (int)D, (int)H, (int)W, (int)scale, cur_stream()), "estd_softargmin_up");
    return {depth, prob};
}

// ------------------------------------------------------------------------------------------------ EST fusion
Tensor warp_volume(const Tensor& vol, const Tensor& mats30, const Tensor& depth_values, double depth_min, double depth_interval)
{
    const OpScope scope(vol);
    TORCH_CHECK(vol.dim() == 4 && mats30.numel() == 30, "warp_volume: vol [C,D,H,W], mats30 [30]");
    TORCH_CHECK(depth_values.numel() >= vol.size(1), "warp_volume: one depth value per plane expected");
    Tensor out = at::empty_like(vol);
    check_status(estd_warp_volume(fptr(vol, "feat_volume"), fptr(mats30, "mats30"), fptr(depth_values, "depth"), (float)depth_min,
                                  (float)depth_interval, out.data_ptr<float>(), (int)vol.size(0), (int)vol.size(1), (int)vol.size(2),
                                  (int)vol.size(3), cur_stream()), "estd_warp_volume");
    return out;
}

Tensor warp_volume_ex(const Tensor& vol, const Tensor& mats30, const Tensor& depth, bool depth_per_voxel, double depth_min,
                      double depth_interval, bool use_disp, double disp_min, double disp_interval, bool border, double padding_value)
{
    const OpScope scope(vol);
    TORCH_CHECK(vol.dim() == 4 && mats30.numel() == 30, "warp_volume: vol [C,D,H,W], mats30 [30]");
    const int64_t need = depth_per_voxel ? vol.size(1) * vol.size(2) * vol.size(3) : vol.size(1);
    TORCH_CHECK(depth.numel() >= need, "warp_volume: depth must hold ", need, " values");
    estd_warp_volume_opts o{};
    o.depth_per_voxel = depth_per_voxel; o.use_disp = use_disp; o.border = border;
    o.depth_min = (float)depth_min; o.depth_interval = (float)depth_interval;
    o.disp_min = (float)disp_min; o.disp_interval = (float)disp_interval; o.padding_value = (float)padding_value;
    Tensor out = at::empty_like(vol);
    check_status(estd_warp_volume_ex(fptr(vol, "feat_volume"), fptr(mats30, "mats30"), fptr(depth, "depth"), &o, out.data_ptr<float>(),
                                     (int)vol.size(0), (int)vol.size(1), (int)vol.size(2), (int)vol.size(3), cur_stream()),
                 "estd_warp_volume_ex");
    return out;
}

Tensor warp_attention(const Tensor& kv_target, at::TensorList kv_sources, const Tensor& mats, const Tensor& depth_values,
                      double depth_min, double depth_interval)
{
    const OpScope scope(kv_target);
    TORCH_CHECK(kv_target.dim() == 4 && kv_target.size(3) == 32, "warp_attention: kv volumes must be [D,H,W,32]");
    const int64_t D = kv_target.size(0), H = kv_target.size(1), W = kv_target.size(2);
    const int n = (int)kv_sources.size();
    TORCH_CHECK(n >= 1, "warp_attention: at least one source volume");
    TORCH_CHECK(n <= ESTD_MAX_ATTENTION_SOURCES, "warp_attention: at most ", ESTD_MAX_ATTENTION_SOURCES, " source volumes, got ", n);
    TORCH_CHECK(mats.numel() == (int64_t)n * 30, "warp_attention: mats must be [n_src,30]");
    TORCH_CHECK(depth_values.numel() >= D, "warp_attention: one depth value per plane expected");
    std::vector<const float*> ptrs(n);
    for (int j = 0; j < n; ++j) {
        TORCH_CHECK(kv_sources[j].sizes() == kv_target.sizes(), "warp_attention: source ", j, " has another shape than the target");
        ptrs[j] = fptr(kv_sources[j], "kv source");
    }
    Tensor xh = at::empty_like(kv_target);
    check_status(estd_warp_attention(fptr(kv_target, "kv target"), ptrs.data(), fptr(mats, "mats"), n, fptr(depth_values, "depth_values"),
                                     (float)depth_min, (float)depth_interval, xh.data_ptr<float>(), (int)D, (int)H, (int)W, cur_stream()),
                 "estd_warp_attention");
    return xh;
}

Tensor attention_prewarped(const Tensor& kv_target, at::TensorList kv_sources)
{
    const OpScope scope(kv_target);
    const int n = (int)kv_sources.size();
    TORCH_CHECK(n >= 1 && n <= ESTD_MAX_ATTENTION_SOURCES, "attention_prewarped: 1..", ESTD_MAX_ATTENTION_SOURCES, " pre-warped source volumes");
    TORCH_CHECK(kv_target.numel() % 32 == 0, "attention_prewarped: kv volumes hold 32 floats per voxel");
    std::vector<const float*> ptrs(n);
    for (int j = 0; j < n; ++j) {
        TORCH_CHECK(kv_sources[j].numel() == kv_target.numel(), "attention_prewarped: source ", j, " has another size than the target");
        ptrs[j] = fptr(kv_sources[j], "kv source");
    }
    Tensor xh = at::empty_like(kv_target);
    check_status(estd_attention_prewarped(fptr(kv_target, "kv target"), ptrs.data(), n, xh.data_ptr<float>(), kv_target.numel() / 32,
                                          cur_stream()), "estd_attention_prewarped");
    return xh;
}

Tensor gru_reset_apply(const Tensor& xh, const Tensor& ru, const Tensor& stats4, const Tensor& gamma_r, const Tensor& beta_r)
{
    const OpScope scope(xh);
    TORCH_CHECK(xh.numel() == ru.numel() && xh.numel() % 32 == 0, "gru_reset_apply: xh and ru are [D,H,W,32] volumes");
    TORCH_CHECK(stats4.numel() == 4 && gamma_r.numel() == 16 && beta_r.numel() == 16, "gru_reset_apply: stats [4], affine [16]");
    Tensor xrh = at::empty_like(xh);
    check_status(estd_gru_reset_apply(fptr(xh, "xh"), fptr(ru, "ru"), fptr(stats4, "stats"), fptr(gamma_r, "gamma"), fptr(beta_r, "beta"),
                                      xrh.data_ptr<float>(), xh.numel() / 32, cur_stream()), "estd_gru_reset_apply");
    return xrh;
}

void gru_blend(const Tensor& xh, const Tensor& ru, const Tensor& o_raw, const Tensor& stats_ru, const Tensor& stats_o,
               const Tensor& gamma_u, const Tensor& beta_u, const Tensor& gamma_o, const Tensor& beta_o, Tensor out_value, int64_t out_stride)
{
    const OpScope scope(xh);
    const int64_t n_vox = xh.numel() / 32;
    TORCH_CHECK(xh.numel() == ru.numel() && o_raw.numel() == n_vox * 16, "gru_blend: xh, ru [D,H,W,32]; o_raw [D,H,W,16]");
    TORCH_CHECK(stats_ru.numel() == 4 && stats_o.numel() == 4 && gamma_u.numel() == 16 && beta_u.numel() == 16 &&
                gamma_o.numel() == 16 && beta_o.numel() == 16, "gru_blend: stats [4], affine [16]");
    check_status(estd_gru_blend(fptr(xh, "xh"), fptr(ru, "ru"), fptr(o_raw, "o_raw"), fptr(stats_ru, "stats_ru"), fptr(stats_o, "stats_o"),
                                fptr(gamma_u, "gamma_u"), fptr(beta_u, "beta_u"), fptr(gamma_o, "gamma_o"), fptr(beta_o, "beta_o"),
                                fptr_mut(out_value, "out_value", false), (int)out_stride, n_vox, cur_stream()), "estd_gru_blend");
}

// ------------------------------------------------------------------------------------------------ 2D backbone epilogues, layouts
Tensor bn_act_nhwc_(Tensor x, const Tensor& scale, const Tensor& shift, bool relu, const OptTensor& residual)
{
    const OpScope scope(x);
    TORCH_CHECK(x.dim() == 4 && x.is_cuda() && x.scalar_type() == at::kFloat && x.is_contiguous(at::MemoryFormat::ChannelsLast),
                "bn_act_nhwc_: expected a float32 ROCm tensor in channels_last memory (no CPU path)");
    const int64_t n = x.size(0), c = x.size(1), h = x.size(2), w = x.size(3);
    const float* res = nullptr;
    if (residual.has_value() && residual->defined()) {
        TORCH_CHECK(residual->sizes() == x.sizes() && residual->is_contiguous(at::MemoryFormat::ChannelsLast) &&
                    residual->scalar_type() == at::kFloat && residual->is_cuda(), "bn_act_nhwc_: residual must match x (shape, channels_last)");
        res = residual->data_ptr<float>();
    }
    TORCH_CHECK(scale.numel() == c && shift.numel() == c, "bn_act_nhwc_: scale/shift must have C entries");
    check_status(estd_bn_act_nhwc(x.data_ptr<float>(), fptr(scale, "scale"), fptr(shift, "shift"), res, relu ? 1 : 0, n * h * w, (int)c,
                                  cur_stream()), "estd_bn_act_nhwc");
    return x;
}

Tensor spp_upsample_cat(const Tensor& raw, const Tensor& skip, at::TensorList branches)
{
    const OpScope scope(raw);
    const int nb = (int)branches.size();
    TORCH_CHECK(raw.dim() == 4 && skip.dim() == 4 && nb >= 1 && nb <= 4, "spp_upsample_cat: NHWC raw/skip and 1..4 branches");
    const int64_t n = raw.size(0), h = raw.size(1), w = raw.size(2), cr = raw.size(3), cs = skip.size(3), cb = branches[0].size(3);
    std::vector<const float*> ptrs(nb);
    std::vector<int> bh(nb), bw(nb);
    for (int k = 0; k < nb; ++k) {
        TORCH_CHECK(branches[k].dim() == 4 && branches[k].size(0) == n && branches[k].size(3) == cb, "spp_upsample_cat: branch shape");
        ptrs[k] = fptr(branches[k], "spp branch"); bh[k] = (int)branches[k].size(1); bw[k] = (int)branches[k].size(2);
    }
    Tensor out = new_f32({n, h, w, cr + cs + nb * cb}, raw);
    check_status(estd_spp_upsample_cat(fptr(raw, "raw"), (int)cr, fptr(skip, "skip"), (int)cs, ptrs.data(), bh.data(), bw.data(), nb, (int)cb,
                                       out.data_ptr<float>(), (int)n, (int)h, (int)w, cur_stream()), "estd_spp_upsample_cat");
    return out;
}

Tensor conv1x1_nhwc(const Tensor& x, const Tensor& w2, const OptTensor& scale, const OptTensor& shift, int64_t stride, bool relu,
                    const OptTensor& residual)
{
    const OpScope scope(x);
    TORCH_CHECK(x.dim() == 4 && w2.dim() == 2 && w2.size(1) == x.size(3), "conv1x1_nhwc: NHWC x [N,H,W,cin] and w [cout,cin] expected");
    TORCH_CHECK(stride == 1 || stride == 2, "conv1x1_nhwc: stride 1 or 2");
    const int64_t n = x.size(0), h = x.size(1), w = x.size(2), cin = x.size(3), cout = w2.size(0);
    const int64_t ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
    estd_conv1x1_desc d{};
    d.N = (int)n; d.H = (int)h; d.W = (int)w; d.cin = (int)cin; d.cout = (int)cout; d.stride = (int)stride; d.relu = relu ? 1 : 0;
    d.in = fptr(x, "x"); d.w = fptr(w2, "w");
    d.scale = opt_fptr(scale, "scale"); d.shift = opt_fptr(shift, "shift");
    if (d.scale) TORCH_CHECK(scale->numel() == cout, "conv1x1_nhwc: scale [cout] expected");
    if (d.shift) TORCH_CHECK(shift->numel() == cout, "conv1x1_nhwc: shift [cout] expected");
    d.residual = opt_fptr(residual, "residual");
    if (d.residual) TORCH_CHECK(residual->numel() == n * ho * wo * cout, "conv1x1_nhwc: residual must be NHWC [N,Ho,Wo,cout]");
    Tensor out = new_f32({n, ho, wo, cout}, x);
    d.out = out.data_ptr<float>();
    check_status(estd_conv1x1_nhwc(&d, cur_stream()), "estd_conv1x1_nhwc");
    return out;
}

Tensor conv2d_taps_nhwc(const Tensor& x, const Tensor& w_taps, const OptTensor& scale, const OptTensor& shift, int64_t ksize, int64_t stride,
                        int64_t pad, bool relu, const OptTensor& residual)
{
    const OpScope scope(x);
    TORCH_CHECK(x.dim() == 4 && w_taps.dim() == 3 && w_taps.size(0) == ksize * ksize && w_taps.size(2) == x.size(3),
                "conv2d_taps_nhwc: NHWC x [N,H,W,cin] and w [k*k,cout,cin] expected");
    TORCH_CHECK((stride == 1 || stride == 2) && pad >= 0, "conv2d_taps_nhwc: stride 1 or 2, pad >= 0");
    const int64_t n = x.size(0), h = x.size(1), w = x.size(2), cin = x.size(3), cout = w_taps.size(1);
    TORCH_CHECK(h + 2 * pad >= ksize && w + 2 * pad >= ksize, "conv2d_taps_nhwc: map smaller than the kernel");
    const int64_t ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
    estd_conv2d_taps_desc d{};
    d.N = (int)n; d.H = (int)h; d.W = (int)w; d.cin = (int)cin; d.cout = (int)cout;
    d.ksize = (int)ksize; d.stride = (int)stride; d.pad = (int)pad; d.relu = relu ? 1 : 0;
    d.in = fptr(x, "x"); d.w = fptr(w_taps, "w");
    d.scale = opt_fptr(scale, "scale"); d.shift = opt_fptr(shift, "shift");
    if (d.scale) TORCH_CHECK(scale->numel() == cout, "conv2d_taps_nhwc: scale [cout] expected");
    if (d.shift) TORCH_CHECK(shift->numel() == cout, "conv2d_taps_nhwc: shift [cout] expected");
    d.residual = opt_fptr(residual, "residual");
    if (d.residual) TORCH_CHECK(residual->numel() == n * ho * wo * cout, "conv2d_taps_nhwc: residual must be NHWC [N,Ho,Wo,cout]");
    Tensor out = new_f32({n, ho, wo, cout}, x);
    d.out = out.data_ptr<float>();
    check_status(estd_conv2d_taps_nhwc(&d, cur_stream()), "estd_conv2d_taps_nhwc");
    return out;
}

Tensor stem7x7s2_nhwc(const Tensor& x, const Tensor& w_packed, const Tensor& scale, const Tensor& shift)
{
    const OpScope scope(x);
    TORCH_CHECK(x.dim() == 4 && x.size(3) == 3 && w_packed.numel() == 7 * 6 * 4 * 64 && scale.numel() == 64 && shift.numel() == 64,
                "stem7x7s2_nhwc: NHWC x [N,H,W,3], packed weights [7,6,4,64], scale/shift [64] expected");
    const int64_t n = x.size(0), h = x.size(1), w = x.size(2);
    Tensor out = new_f32({n, (h - 1) / 2 + 1, (w - 1) / 2 + 1, 64}, x);
    check_status(estd_stem7x7s2_nhwc(fptr(x, "x"), fptr(w_packed, "packed weights"), fptr(scale, "scale"), fptr(shift, "shift"), out.data_ptr<float>(),
                                     (int)n, (int)h, (int)w, cur_stream()), "estd_stem7x7s2_nhwc");
    return out;
}

Tensor maxpool3x3s2_nhwc(const Tensor& x)
{
    const OpScope scope(x);
    TORCH_CHECK(x.dim() == 4 && x.size(3) % 4 == 0, "maxpool3x3s2_nhwc: NHWC x [N,H,W,C], C a multiple of 4, expected");
    const int64_t n = x.size(0), h = x.size(1), w = x.size(2), c = x.size(3);
    Tensor out = new_f32({n, (h - 1) / 2 + 1, (w - 1) / 2 + 1, c}, x);
    check_status(estd_maxpool3x3s2_nhwc(fptr(x, "x"), out.data_ptr<float>(), (int)n, (int)h, (int)w, (int)c, cur_stream()), "estd_maxpool3x3s2_nhwc");
    return out;
}

Tensor avgpool_nhwc(const Tensor& x, int64_t k)
{
    const OpScope scope(x);
    TORCH_CHECK(x.dim() == 4 && x.size(3) % 4 == 0 && k >= 1 && k <= x.size(1) && k <= x.size(2),
                "avgpool_nhwc: NHWC x [N,H,W,C], C a multiple of 4, 1 <= k <= min(H, W) expected");
    const int64_t n = x.size(0), h = x.size(1), w = x.size(2), c = x.size(3);
    Tensor out = new_f32({n, h / k, w / k, c}, x);
    check_status(estd_avgpool_nhwc(fptr(x, "x"), out.data_ptr<float>(), (int)n, (int)h, (int)w, (int)c, (int)k, cur_stream()), "estd_avgpool_nhwc");
    return out;
}

Tensor conv2d_small_nhwc(const Tensor& x, const Tensor& w_packed, const Tensor& scale, const Tensor& shift, int64_t cout, int64_t ksize,
                         int64_t stride, bool relu)
{
    const OpScope scope(x);
    TORCH_CHECK(x.dim() == 4, "conv2d_small_nhwc: NHWC x expected");
    const int64_t n = x.size(0), h = x.size(1), w = x.size(2), cin = x.size(3);
    TORCH_CHECK((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2) && cout > 0 && cout % 16 == 0 && cin % 16 == 0,
                "conv2d_small_nhwc: ksize 1|3, stride 1|2, channel counts multiples of 16");
    TORCH_CHECK(w_packed.numel() == cout / 16 * ksize * ksize * (cin / 16) * 256 && scale.numel() == cout && shift.numel() == cout,
                "conv2d_small_nhwc: packed weights [cout/16][taps][cin/16][64][4] and scale/shift [cout] expected");
    const int64_t pad = ksize / 2, ho = (h + 2 * pad - ksize) / stride + 1, wo = (w + 2 * pad - ksize) / stride + 1;
    Tensor out = new_f32({n, ho, wo, cout}, x);
    check_status(estd_conv2d_small_nhwc(fptr(x, "x"), fptr(w_packed, "packed weights"), fptr(scale, "scale"), fptr(shift, "shift"),
                                        out.data_ptr<float>(), (int)n, (int)h, (int)w, (int)cin, (int)cout, (int)ksize, (int)stride, relu ? 1 : 0,
                                        cur_stream()), "estd_conv2d_small_nhwc");
    return out;
}

Tensor conv2d_k3_to16_nhwc(const Tensor& x, const Tensor& w_packed, const Tensor& scale, const Tensor& shift, bool upsample)
{
    const OpScope scope(x);
    TORCH_CHECK(x.dim() == 4 && (x.size(3) == 16 || x.size(3) == 32), "conv2d_k3_to16_nhwc: NHWC x with 16 or 32 channels expected");
    const int64_t n = x.size(0), c = x.size(3), h = (upsample ? 2 : 1) * x.size(1), w = (upsample ? 2 : 1) * x.size(2);
    TORCH_CHECK(w_packed.numel() == 9 * (c / 16) * 64 * 4 && scale.numel() == 16 && shift.numel() == 16,
                "conv2d_k3_to16_nhwc: packed weights [9][cin/16][64][4] and scale/shift [16] expected");
    Tensor out = new_f32({n, h, w, 16}, x);
    check_status(estd_conv2d_k3_to16_nhwc(fptr(x, "x"), fptr(w_packed, "packed weights"), fptr(scale, "scale"), fptr(shift, "shift"),
                                          out.data_ptr<float>(), (int)n, (int)h, (int)w, (int)c, upsample ? 1 : 0, cur_stream()),
                 "estd_conv2d_k3_to16_nhwc");
    return out;
}

Tensor normalise_nhwc(const Tensor& imgs)
{
    const OpScope scope(imgs);
    TORCH_CHECK(imgs.dim() == 4 && imgs.size(1) == 3, "normalise_nhwc: [N,3,H,W] images expected");
    const int64_t n = imgs.size(0), h = imgs.size(2), w = imgs.size(3);
    Tensor out = new_f32({n, h, w, 3}, imgs);
    check_status(estd_normalise_nhwc(fptr(imgs, "imgs"), out.data_ptr<float>(), (int)n, h * w, cur_stream()), "estd_normalise_nhwc");
    return out;
}

Tensor stem3x3s2_nhwc(const Tensor& x, const Tensor& weight, const Tensor& scale, const Tensor& shift)
{
    const OpScope scope(x);
    TORCH_CHECK(x.dim() == 4 && x.size(3) == 3 && weight.dim() == 4 && weight.size(0) == 32 && weight.size(1) == 3 && weight.size(2) == 3 &&
                weight.size(3) == 3 && scale.numel() == 32 && shift.numel() == 32,
                "stem3x3s2_nhwc: NHWC x [N,H,W,3], weight [32,3,3,3], scale/shift [32] expected");
    const int64_t n = x.size(0), h = x.size(1), w = x.size(2);
    Tensor out = new_f32({n, (h - 1) / 2 + 1, (w - 1) / 2 + 1, 32}, x);
    check_status(estd_stem3x3s2_nhwc(fptr(x, "x"), fptr(weight, "weight"), fptr(scale, "scale"), fptr(shift, "shift"), out.data_ptr<float>(), (int)n,
                                     (int)h, (int)w, cur_stream()), "estd_stem3x3s2_nhwc");
    return out;
}

Tensor nhwc_to_planes(const Tensor& x)
{
    const OpScope scope(x);
    TORCH_CHECK(x.dim() == 4, "nhwc_to_planes: NHWC x [N,H,W,C] expected");
    const int64_t n = x.size(0), h = x.size(1), w = x.size(2), c = x.size(3);
    Tensor out = new_f32({n, c, h, w}, x);
    check_status(estd_nhwc_to_planes(fptr(x, "x"), (int)c, out.data_ptr<float>(), (int)n, h * w, cur_stream()), "estd_nhwc_to_planes");
    return out;
}

Tensor planes_cat_nhwc(const Tensor& a, const Tensor& b, bool relu_b)
{
    const OpScope scope(a);
    TORCH_CHECK(a.dim() == 4 && b.dim() == 4 && a.size(0) == b.size(0) && a.size(2) == b.size(2) && a.size(3) == b.size(3),
                "planes_cat_nhwc: two NCHW stacks of the same N, H, W expected");
    const int64_t n = a.size(0), ca = a.size(1), cb = b.size(1), h = a.size(2), w = a.size(3);
    Tensor out = new_f32({n, h, w, ca + cb}, a);
    check_status(estd_planes_cat_nhwc(fptr(a, "a"), (int)ca, fptr(b, "b"), (int)cb, relu_b ? 1 : 0, out.data_ptr<float>(), (int)n, h * w,
                                      cur_stream()), "estd_planes_cat_nhwc");
    return out;
}

Tensor upsample2_cat_nhwc(const Tensor& x, const Tensor& skip)
{
    const OpScope scope(x);
    TORCH_CHECK(x.dim() == 4 && skip.dim() == 4 && x.size(0) == skip.size(0) && 2 * x.size(1) == skip.size(1) && 2 * x.size(2) == skip.size(2),
                "upsample2_cat_nhwc: NHWC x [N,H/2,W/2,Cx] and skip [N,H,W,Cs] expected");
    const int64_t n = skip.size(0), h = skip.size(1), w = skip.size(2), cx = x.size(3), cs = skip.size(3);
    Tensor out = new_f32({n, h, w, cx + cs}, x);
    check_status(estd_upsample2_cat_nhwc(fptr(x, "x"), (int)cx, fptr(skip, "skip"), (int)cs, out.data_ptr<float>(), (int)n, (int)h, (int)w,
                                         cur_stream()), "estd_upsample2_cat_nhwc");
    return out;
}

Tensor disp_head_nhwc(const Tensor& x, const Tensor& weight, const Tensor& bias, double depth_max, int64_t upscale)
{
    const OpScope scope(x);
    TORCH_CHECK(x.dim() == 4 && weight.dim() == 4 && weight.size(0) == 1 && weight.size(1) == x.size(3) && weight.size(2) == 3 &&
                weight.size(3) == 3 && bias.numel() == 1, "disp_head_nhwc: NHWC x [N,H,W,C], weight [1,C,3,3], bias [1] expected");
    TORCH_CHECK(upscale == 1 || upscale == 2, "disp_head_nhwc: upscale must be 1 or 2");
    const int64_t n = x.size(0), h = x.size(1), w = x.size(2), c = x.size(3);
    Tensor out = new_f32({n, 1, upscale * h, upscale * w}, x);
    check_status(estd_disp_head_nhwc(fptr(x, "x"), fptr(weight, "weight"), fptr(bias, "bias"), (float)depth_max, out.data_ptr<float>(), (int)n,
                                     (int)h, (int)w, (int)c, (int)upscale, cur_stream()), "estd_disp_head_nhwc");
    return out;
}

void cdhw_to_vol(const Tensor& src, Tensor dst, int64_t dst_stride, int64_t dst_off)
{
    const OpScope scope(src);
    TORCH_CHECK(src.dim() >= 2, "cdhw_to_vol: src must be [C, ...]");
    const int64_t C = src.size(0), S = src.numel() / C;
    TORCH_CHECK(dst.numel() >= S * dst_stride, "cdhw_to_vol: destination too small");
    check_status(estd_cdhw_to_vol(fptr(src, "volume"), fptr_mut(dst, "destination", false), (int)C, S, (int)dst_stride, (int)dst_off, cur_stream()),
                 "estd_cdhw_to_vol");
}

Tensor vol_to_cdhw(const Tensor& src, int64_t C, at::IntArrayRef dims, int64_t src_stride, int64_t src_off)
{
    const OpScope scope(src);
    TORCH_CHECK(dims.size() == 3, "vol_to_cdhw: dims = (D, H, W)");
    const int64_t S = dims[0] * dims[1] * dims[2];
    TORCH_CHECK(src.numel() >= S * src_stride, "vol_to_cdhw: source too small");
    Tensor out = new_f32({C, dims[0], dims[1], dims[2]}, src);
    check_status(estd_vol_to_cdhw(fptr(src, "source", false), out.data_ptr<float>(), (int)C, S, (int)src_stride, (int)src_off, cur_stream()),
                 "estd_vol_to_cdhw");
    return out;
}

// ------------------------------------------------------------------------------------------------ camera algebra on the HOST
// The reference composes its camera matrices with torch.inverse / torch.matmul on [B,4,4] CPU tensors (model_hybrid.py:74-88,
// homo_utils.py:469-471, hybrid_depth_decoder.py:235, homo_utils.py:51,:258).  These matrices decide which samples fall across
// the discontinuous |norm| > 1 masks, so the product evaluates them with the SAME ATen CPU kernels, shapes and association
// order -- bit-identical matrices (estdepth_amd/camera.py is the line-by-line Python statement of this function; a CPU test
// asserts equality).  One call per forward from C++ instead of ~40 Python-dispatched ATen calls (0.4 ms -> 0.06 ms of host
// time during which the GPU would idle behind the D2H copy of the poses).
std::tuple<Tensor, Tensor> camera_matrices_host(const Tensor& cam_poses, const Tensor& cam_intr_q, at::TensorList pre_poses, bool with_volume)
{
    TORCH_CHECK(cam_poses.device().is_cpu() && cam_intr_q.device().is_cpu(), "camera_matrices_host: CPU tensors expected (copy the poses once)");
    TORCH_CHECK(cam_poses.dim() == 4 && cam_poses.size(0) == 1 && cam_poses.size(2) == 4 && cam_poses.size(3) == 4 && cam_poses.size(1) >= 3,
                "camera_matrices_host: cam_poses must be [1,V,4,4] with V >= 3");
    TORCH_CHECK(cam_intr_q.sizes() == at::IntArrayRef({1, 3, 3}), "camera_matrices_host: cam_intr must be [1,3,3]");
    using namespace at::indexing;
    const Tensor poses = cam_poses.to(at::kFloat), K = cam_intr_q.to(at::kFloat);
    const int64_t V = poses.size(1), T = V - 2;
    auto view_proj = [&](int64_t v) {
        Tensor extrinsic = at::inverse(poses.index({Slice(), v, Slice(), Slice()}));                   // model_hybrid.py:74,:83
        Tensor proj = extrinsic.clone();
        proj.index_put_({Slice(), Slice(None, 3), Slice(None, 4)},
                        at::matmul(K, extrinsic.index({Slice(), Slice(None, 3), Slice(None, 4)})));    // :87-88
        return proj;
    };
    Tensor sweep = at::empty({T, 2, 12}, poses.options());
    for (int64_t t = 0; t < T; ++t) {
        const Tensor ref_inv = at::inverse(view_proj(t + 1));
        const int64_t srcs[2] = {t, t + 2};
        for (int k = 0; k < 2; ++k) {
            const Tensor pr = at::matmul(view_proj(srcs[k]), ref_inv);                                 // homo_utils.py:469
            sweep.index_put_({t, k, Slice(None, 9)}, pr.index({0, Slice(None, 3), Slice(None, 3)}).reshape({9}));   // :470
            sweep.index_put_({t, k, Slice(9, None)}, pr.index({0, Slice(None, 3), 3}));                              // :471
        }
    }
    Tensor vol;
    if (with_volume) {
        std::vector<Tensor> P;
        for (int64_t t = 0; t < T; ++t) P.push_back(poses.index({Slice(), t + 1}).reshape({1, 4, 4}));
        for (const Tensor& p : pre_poses) {
            TORCH_CHECK(p.device().is_cpu() && p.numel() == 16, "camera_matrices_host: memory poses must be CPU [1,4,4] tensors");
            P.push_back(p.to(at::kFloat).reshape({1, 4, 4}));
        }
        const int64_t n = (int64_t)P.size();
        TORCH_CHECK(n >= 2, "camera_matrices_host: the EST fusion needs at least one other view");
        const Tensor kinv = at::inverse(K);                                                            // homo_utils.py:51
        vol = at::empty({T, n - 1, 30}, poses.options());
        for (int64_t i = 0; i < T; ++i) {
            const Tensor inv_i = at::inverse(P[i]);
            int64_t r = 0;
            for (int64_t j = 0; j < n; ++j) {
                if (j == i) continue;
                const Tensor rel = at::matmul(P[j], inv_i);                                            // hybrid_depth_decoder.py:235 (Q8)
                const Tensor m = at::inverse(rel);                                                     // homo_utils.py:258
                vol.index_put_({i, r, Slice(None, 9)}, kinv[0].reshape({9}));
                vol.index_put_({i, r, Slice(9, 21)}, m.index({0, Slice(None, 3), Slice()}).reshape({12}));
                vol.index_put_({i, r, Slice(21, None)}, K[0].reshape({9}));
                ++r;
            }
        }
    } else {
        vol = at::empty({0}, poses.options());
    }
    return {sweep, vol};
}

int64_t set_reserved_cus(int64_t n) { return estd_set_reserved_cus((int)n); }
void profile_mark(int64_t id) { check_status(estd_profile_mark((int)id, cur_stream()), "estd_profile_mark"); }
int64_t conv3d_grid(int64_t N, int64_t D, int64_t H, int64_t W)
{
    const int g = estd_conv3d_k3_grid((int)N, (int)D, (int)H, (int)W);
    check_status(g < 0 ? g : ESTD_OK, "estd_conv3d_k3_grid");
    return g;
}

}  // namespace

TORCH_LIBRARY(estdepth_hip, m)
{
    m.def("cam_pair_proj(Tensor src_proj, Tensor ref_proj) -> Tensor");
    m.def("cam_sweep_proj(Tensor ref_pose, Tensor src_pose, Tensor cam_intr) -> Tensor");
    m.def("cam_volume_mats(Tensor pose_j, Tensor? pose_i, Tensor cam_intr, Tensor(a!) out) -> ()");
    m.def("homo_warping(Tensor src_fea, Tensor proj12, Tensor depth_values, int D) -> Tensor");
    m.def("homo_warping_px(Tensor src_fea, Tensor proj12, Tensor depth_dhw) -> Tensor");
    m.def("mix1x1(Tensor feature, Tensor weight, Tensor? bias) -> Tensor");
    m.def("homo_warp_costvol(Tensor src_mix, Tensor ref_mix, Tensor proj12, Tensor depth_values, int D, Tensor(a!) out) -> ()");
    m.def("conv3d_k3(Tensor x, Tensor? x_extra, Tensor? w_main, Tensor? w_extra, Tensor? w_xout, Tensor? w_alt, Tensor scale, Tensor shift, "
          "int[] dims, int cin_main, int in_stride, int n_tiles, int act_a, int act_b, int act_split, Tensor(a!)? out, int out_stride, "
          "int out_channels, Tensor? residual, Tensor? residual2, float out_scale, bool accumulate, Tensor(b!)? out_extra, Tensor? head_w, "
          "Tensor? head_b, Tensor(c!)? out_head, Tensor(d!)? stats_partials, int variant, Tensor? gate_r=None, Tensor? gate_stats=None, "
          "Tensor? gate_gamma=None, Tensor? gate_beta=None) -> ()");
    m.def("conv2d_k3(Tensor x_nhwc, Tensor? w, Tensor? w_alt, Tensor scale, Tensor shift, int cout, int dilation, int group_tiles, "
          "bool relu_before_residual, bool relu_after_residual, Tensor? residual, int variant) -> Tensor");
    m.def("groupnorm_finalize(Tensor partials, int n_blocks, float count, float eps) -> Tensor");
    m.def("softargmin_up(Tensor logits, Tensor depth_values, int scale) -> (Tensor, Tensor)");
    m.def("warp_volume(Tensor feat_volume, Tensor mats30, Tensor depth_values, float depth_min, float depth_interval) -> Tensor");
    m.def("warp_volume_ex(Tensor feat_volume, Tensor mats30, Tensor depth, bool depth_per_voxel, float depth_min, float depth_interval, "
          "bool use_disp, float disp_min, float disp_interval, bool border, float padding_value) -> Tensor");
    m.def("warp_attention(Tensor kv_target, Tensor[] kv_sources, Tensor mats, Tensor depth_values, float depth_min, float depth_interval) -> Tensor");
    m.def("attention_prewarped(Tensor kv_target, Tensor[] kv_sources) -> Tensor");
    m.def("gru_reset_apply(Tensor xh, Tensor ru, Tensor stats4, Tensor gamma_r, Tensor beta_r) -> Tensor");
    m.def("gru_blend(Tensor xh, Tensor ru, Tensor o_raw, Tensor stats_ru, Tensor stats_o, Tensor gamma_u, Tensor beta_u, Tensor gamma_o, "
          "Tensor beta_o, Tensor(a!) out_value, int out_stride) -> ()");
    m.def("bn_act_nhwc_(Tensor(a!) x, Tensor scale, Tensor shift, bool relu, Tensor? residual) -> Tensor(a!)");
    m.def("spp_upsample_cat(Tensor raw, Tensor skip, Tensor[] branches) -> Tensor");
    m.def("conv1x1_nhwc(Tensor x, Tensor w, Tensor? scale, Tensor? shift, int stride, bool relu, Tensor? residual) -> Tensor");
    m.def("conv2d_taps_nhwc(Tensor x, Tensor w, Tensor? scale, Tensor? shift, int ksize, int stride, int pad, bool relu, Tensor? residual) -> Tensor");
    m.def("stem7x7s2_nhwc(Tensor x, Tensor w_packed, Tensor scale, Tensor shift) -> Tensor");
    m.def("maxpool3x3s2_nhwc(Tensor x) -> Tensor");
    m.def("avgpool_nhwc(Tensor x, int k) -> Tensor");
    m.def("conv2d_small_nhwc(Tensor x, Tensor w_packed, Tensor scale, Tensor shift, int cout, int ksize, int stride, bool relu) -> Tensor");
    m.def("conv2d_k3_to16_nhwc(Tensor x, Tensor w_packed, Tensor scale, Tensor shift, bool upsample) -> Tensor");
    m.def("normalise_nhwc(Tensor imgs) -> Tensor");
    m.def("stem3x3s2_nhwc(Tensor x, Tensor weight, Tensor scale, Tensor shift) -> Tensor");
    m.def("nhwc_to_planes(Tensor x) -> Tensor");
    m.def("planes_cat_nhwc(Tensor a, Tensor b, bool relu_b) -> Tensor");
    m.def("upsample2_cat_nhwc(Tensor x, Tensor skip) -> Tensor");
    m.def("disp_head_nhwc(Tensor x, Tensor weight, Tensor bias, float depth_max, int upscale) -> Tensor");
    m.def("cdhw_to_vol(Tensor src, Tensor(a!) dst, int dst_stride, int dst_off) -> ()");
    m.def("vol_to_cdhw(Tensor src, int C, int[] dims, int src_stride, int src_off) -> Tensor");
    m.def("camera_matrices_host(Tensor cam_poses, Tensor cam_intr, Tensor[] pre_poses, bool with_volume) -> (Tensor, Tensor)");
    m.def("profile_mark(int id) -> ()");
    m.def("set_reserved_cus(int n) -> int");
    m.def("conv3d_grid(int N, int D, int H, int W) -> int");
}

// Every operator is ROCm-only: registered for the CUDA dispatch key (= HIP in PyTorch-ROCm).  CPU tensors find no kernel and
// raise the dispatcher's NotImplementedError -- there is no CPU fallback by design.  The two argument-free helpers are
// registered as CompositeExplicitAutograd (no tensor to dispatch on).
TORCH_LIBRARY_IMPL(estdepth_hip, CUDA, m)
{
    m.impl("cam_pair_proj", cam_pair_proj);
    m.impl("cam_sweep_proj", cam_sweep_proj);
    m.impl("cam_volume_mats", cam_volume_mats);
    m.impl("homo_warping", homo_warping);
    m.impl("homo_warping_px", homo_warping_px);
    m.impl("mix1x1", mix1x1);
    m.impl("homo_warp_costvol", homo_warp_costvol);
    m.impl("conv3d_k3", conv3d_k3);
    m.impl("conv2d_k3", conv2d_k3);
    m.impl("groupnorm_finalize", groupnorm_finalize);
    m.impl("softargmin_up", softargmin_up);
    m.impl("warp_volume", warp_volume);
    m.impl("warp_volume_ex", warp_volume_ex);
    m.impl("warp_attention", warp_attention);
    m.impl("attention_prewarped", attention_prewarped);
    m.impl("gru_reset_apply", gru_reset_apply);
    m.impl("gru_blend", gru_blend);
    m.impl("bn_act_nhwc_", bn_act_nhwc_);
    m.impl("spp_upsample_cat", spp_upsample_cat);
    m.impl("conv1x1_nhwc", conv1x1_nhwc);
    m.impl("conv2d_taps_nhwc", conv2d_taps_nhwc);
    m.impl("stem7x7s2_nhwc", stem7x7s2_nhwc);
    m.impl("maxpool3x3s2_nhwc", maxpool3x3s2_nhwc);
    m.impl("avgpool_nhwc", avgpool_nhwc);
    m.impl("conv2d_small_nhwc", conv2d_small_nhwc);
    m.impl("conv2d_k3_to16_nhwc", conv2d_k3_to16_nhwc);
    m.impl("normalise_nhwc", normalise_nhwc);
    m.impl("stem3x3s2_nhwc", stem3x3s2_nhwc);
    m.impl("nhwc_to_planes", nhwc_to_planes);
    m.impl("planes_cat_nhwc", planes_cat_nhwc);
    m.impl("upsample2_cat_nhwc", upsample2_cat_nhwc);
    m.impl("disp_head_nhwc", disp_head_nhwc);
    m.impl("cdhw_to_vol", cdhw_to_vol);
    m.impl("vol_to_cdhw", vol_to_cdhw);
}

TORCH_LIBRARY_IMPL(estdepth_hip, CPU, m)
{
    m.impl("camera_matrices_host", camera_matrices_host);      // host-side algebra on CPU tensors (not a compute fallback)
}

TORCH_LIBRARY_IMPL(estdepth_hip, CompositeExplicitAutograd, m)
{
    m.impl("profile_mark", profile_mark);
    m.impl("set_reserved_cus", set_reserved_cus);
    m.impl("conv3d_grid", conv3d_grid);
}
