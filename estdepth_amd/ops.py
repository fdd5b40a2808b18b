"""Torch-tensor front-end of the C ABI (device memory, streams and allocation are PyTorch-ROCm's;
the arithmetic is libestd_hip.so's).  Every function enqueues on the current HIP stream and
returns tensors owned by the caching allocator.  CUDA(ROCm)-only: CPU tensors raise RuntimeError.
"""
import ctypes
import os

import torch

from . import _native as N
from . import packing

# Binding of the hot-path operators:
#   "torch"  (default) -- PyTorch custom operators torch.ops.estdepth_hip.* registered with TORCH_LIBRARY by
#             csrc/torch_ops.cpp (libestd_torch_ops.so): dispatcher, TORCH_CHECK validation, at::empty outputs, current
#             HIP stream taken in C++;
#   "ctypes" -- the torch-free C ABI of libestd_hip.so called directly with raw device pointers (what a non-PyTorch host
#             would bind; kept as the second test path: ESTD_BINDING=ctypes).
# Both end in the same extern "C" entry points; there is no CPU / eager fallback under either.
BINDING = os.environ.get("ESTD_BINDING", "torch")
_TORCH_OPS_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libestd_torch_ops.so")
_torch_ops = None


def T():
    """torch.ops.estdepth_hip (loads libestd_torch_ops.so once; RuntimeError when it has not been built)."""
    global _torch_ops
    if _torch_ops is None:
        if BINDING not in ("torch", "ctypes"):
            raise RuntimeError("ESTD_BINDING must be 'torch' or 'ctypes', got %r" % (BINDING,))
        if not os.path.exists(_TORCH_OPS_LIB):
            raise RuntimeError("libestd_torch_ops.so not found at %s -- build it with `python -m estdepth_amd.build` "
                               "(there is no CPU/eager fallback)" % _TORCH_OPS_LIB)
        N.lib()                                   # libestd_hip.so first: the operator library links against it
        torch.ops.load_library(_TORCH_OPS_LIB)
        _torch_ops = torch.ops.estdepth_hip
    return _torch_ops


def _use_torch():
    return BINDING == "torch"

ACT = {"none": 0, "relu": 1, "tanh": 2}
MAX_ATTENTION_SOURCES = 16      # ESTD_MAX_ATTENTION_SOURCES (include/estd_hip.h)

# bench.py sets this to a list to collect (group, amount, start_event, end_event) around every launch of the hot-path
# kernels on the stream they are launched on: amount = algorithmic FLOPs (groups "conv3d:<Cin>-><Cout>[+x]") or
# algorithmic bytes (SURVEY §8d figures; groups "homo_warp_costvol", "warp_attention", "gru_elementwise", "softargmin").
PROFILE = None


class _Prof:
    """HIP-event pair around one launch (events are recorded on torch's CURRENT stream = the launch stream)."""
    __slots__ = ("group", "amount", "e0")

    def __init__(self, group, amount):
        self.group, self.amount, self.e0 = group, amount, None

    def __enter__(self):
        if PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record(torch.cuda.current_stream())
        return self

    def __exit__(self, *exc):
        if self.e0 is not None and PROFILE is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(torch.cuda.current_stream())
            PROFILE.append((self.group, float(self.amount), self.e0, e1))
        return False

# Arithmetic of the plain 32->32 3x3x3 convolutions: "f32" = v_mfma_f32_16x16x4_f32 (csrc/conv3d_mfma.hip),
# "bf16x3" = exact 3-way bf16 operand split, six bf16 MFMAs per product block (csrc/conv3d_split_bf16.hip).
CONV3D_ARITH = os.environ.get("ESTD_CONV3D_ARITH", "f32")
CONV2D_ARITH = os.environ.get("ESTD_CONV2D_ARITH", "f32")     # same choice for the 3x3 / dilation-1 NHWC convolutions
# Algorithm of the plain 32->32 3x3x3 convolutions under CONV3D_ARITH == "f32" (every product an fp32 MFMA either way):
# "wino2" = depth AND row axis in Winograd F(2,3) form for the plain 32 -> 32 instance, 0.444 of the products
# (csrc/conv3d_wino2.hip, every 3x3x3 instance of the step); "wino" = depth axis only, 2/3 of the products
# (csrc/conv3d_wino.hip); "direct" = 27 taps (csrc/conv3d_mfma.hip)
CONV3D_ALGO = os.environ.get("ESTD_CONV3D_ALGO", "wino2")
# A/B: "0" sends the 33 -> 33 convolution (dres2) to the depth-only Winograd kernel as in round 3
W2_XOUT = os.environ.get("ESTD_W2_XOUT", "1") != "0"
# the plain 32 -> 32 instances of the two-axis Winograd kernel on the operand-reuse form (csrc/conv3d_wino2x.hip: one wave per SIMD,
# 32x32x2 MFMAs, transforms in front of the LDS); opt-in ("1"): at parity with the 8-wave kernel of csrc/conv3d_wino2.hip, not faster (profiles/r5_wino2x_table.txt)
W2X = os.environ.get("ESTD_W2X", "0") != "0"
# the 32 -> 32 instances without a scalar channel (BN / activation / residuals / running sum / GroupNorm partials) with ALL THREE axes in Winograd form
# (csrc/conv3d_wino3.hip: F(2x2x2, 3x3x3), 8/27 of the direct products; default since round 5: 0.65 vs 0.81 ms for 3 volumes, Joint step 16.9 -> 15.8 ms; "0": two-axis kernel)
W3 = os.environ.get("ESTD_W3", "1") != "0"
# the key || value convolution (33 -> 32) on the three-axis kernel's scalar-channel instance as well (0.73 vs 0.85 ms for 3 volumes; "0": two-axis kernel)
W3_EXTRA = os.environ.get("ESTD_W3_EXTRA", "1") != "0"
# dres2 (33 -> 33): the 32 main output channels on the three-axis kernel's scalar-channel instance + output channel 32 as a pass of its own
# (csrc/conv3d_xout.hip: taps as matrix rows) instead of the two-axis kernel's 33 -> 33 instance ("0")
W3_XOUT = os.environ.get("ESTD_W3_XOUT", "1") != "0"
# same choice for the 3x3 / dilation-1 NHWC convolutions: row axis in Winograd F(2,3) form (csrc/conv2d_wino.hip) or direct
CONV2D_ALGO = os.environ.get("ESTD_CONV2D_ALGO", "wino2")
CONV2D_NT = os.environ.get("ESTD_CONV2D_NT", "auto")
C2W2_DIL2 = os.environ.get("ESTD_C2W2_DIL2", "1") == "1"      # A/B switch: dilation-2 convolutions on the F(2x2,3x3) kernel too (0: row-only kernel)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError("%s must be a tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must live on a ROCm device (estdepth_amd has no CPU path); got %s" % (name, t.device))
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    return t


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def set_reserved_cus(n):
    """Compute units the persistent convolution grids leave free for a concurrent collective (include/estd_hip.h)."""
    if _use_torch():
        return int(T().set_reserved_cus(int(n)))
    return int(N.lib().estd_set_reserved_cus(int(n)))


def get_reserved_cus():
    return int(N.lib().estd_get_reserved_cus())


def profile_mark(idx):
    if _use_torch():
        return T().profile_mark(int(idx))
    N.check(N.lib().estd_profile_mark(int(idx), _stream()), "estd_profile_mark")


# ---------------------------------------------------------------------------------- camera algebra
def cam_pair_proj(src_proj, ref_proj):
    """rot|trans of src_proj @ inverse(ref_proj) for one batch element -> [12]."""
    if _use_torch():
        return T().cam_pair_proj(src_proj, ref_proj)
    out = torch.empty(12, device=src_proj.device, dtype=torch.float32)
    N.check(N.lib().estd_cam_pair_proj(_p(_chk(src_proj, "src_proj")), _p(_chk(ref_proj, "ref_proj")), _p(out), _stream()),
            "estd_cam_pair_proj")
    return out


def cam_sweep_proj(ref_pose, src_pose, intr):
    if _use_torch():
        return T().cam_sweep_proj(ref_pose, src_pose, intr)
    out = torch.empty(12, device=ref_pose.device, dtype=torch.float32)
    N.check(N.lib().estd_cam_sweep_proj(_p(_chk(ref_pose, "ref_pose")), _p(_chk(src_pose, "src_pose")),
                                        _p(_chk(intr, "cam_intr")), _p(out), _stream()), "estd_cam_sweep_proj")
    return out


def cam_volume_mats(pose_j, pose_i, intr, out=None):
    if out is None:
        out = torch.empty(30, device=pose_j.device, dtype=torch.float32)
    if _use_torch():
        T().cam_volume_mats(pose_j, pose_i, intr, out)
        return out
    N.check(N.lib().estd_cam_volume_mats(_p(_chk(pose_j, "pose_j")), _p(_chk(pose_i, "pose_i")) if pose_i is not None else None,
                                         _p(_chk(intr, "cam_intr")), _p(out), _stream()), "estd_cam_volume_mats")
    return out


# ---------------------------------------------------------------------------------- plane sweep
def homo_warping_chw(src_chw, proj12, depth_values, D):
    if _use_torch():
        return T().homo_warping(src_chw, proj12, depth_values, D)
    C, H, W = src_chw.shape
    out = torch.empty((C, D, H, W), device=src_chw.device, dtype=torch.float32)
    N.check(N.lib().estd_homo_warping(_p(_chk(src_chw, "src_fea")), _p(proj12), _p(_chk(depth_values, "depth_values")),
                                      _p(out), C, D, H, W, _stream()), "estd_homo_warping")
    return out


def homo_warping_px_chw(src_chw, proj12, depth_dhw):
    """per-pixel depth hypotheses [D,H,W] (homo_utils.py:462)."""
    if _use_torch():
        return T().homo_warping_px(src_chw, proj12, depth_dhw)
    C, H, W = src_chw.shape
    D = depth_dhw.shape[0]
    out = torch.empty((C, D, H, W), device=src_chw.device, dtype=torch.float32)
    N.check(N.lib().estd_homo_warping_px(_p(_chk(src_chw, "src_fea")), _p(proj12), _p(_chk(depth_dhw, "depth_values")),
                                         _p(out), C, D, H, W, _stream()), "estd_homo_warping_px")
    return out


def mix1x1(in_chw, w, bias):
    """[Cin,H,W] -> [H,W,Cout] channel mix."""
    if _use_torch():
        return T().mix1x1(in_chw, w, bias)
    Cin, H, W = in_chw.shape
    Cout = w.shape[0]
    out = torch.empty((H, W, Cout), device=in_chw.device, dtype=torch.float32)
    N.check(N.lib().estd_mix1x1_chw_to_hwc(_p(_chk(in_chw, "feature")), _p(_chk(w, "mix weight")),
                                           _p(bias) if bias is not None else None, _p(out), Cin, Cout, H * W, _stream()),
            "estd_mix1x1_chw_to_hwc")
    return out


def homo_warp_costvol(src_mix, ref_mix, proj12, depth_values, D, out=None):
    H, W, _ = src_mix.shape
    if out is None:
        out = torch.empty((D, H, W, 32), device=src_mix.device, dtype=torch.float32)
    with _Prof("homo_warp_costvol", 4.0 * 32 * H * W * (2 + D)):          # src map + ref map + one volume
        if _use_torch():
            T().homo_warp_costvol(src_mix, ref_mix, proj12, depth_values, D, out)
            return out
        N.check(N.lib().estd_homo_warp_costvol(_p(_chk(src_mix, "src_mix")), _p(_chk(ref_mix, "ref_mix")), _p(proj12),
                                               _p(_chk(depth_values, "depth_values")), _p(out), D, H, W, _stream()),
                "estd_homo_warp_costvol")
    return out


# ---------------------------------------------------------------------------------- conv3d
class _Packs:
    """Weight forms of a plan, packed and uploaded on first use (``define`` registers how, ``has`` says whether the plan's shape admits the
    form, ``get`` packs once).  Shared by the copies ``with_shift_scaled`` makes."""

    def __init__(self, device):
        self.device, self._fns, self._vals = device, {}, {}

    def define(self, name, fn):
        self._fns[name] = fn

    def has(self, name):
        return name in self._fns

    def get(self, name):
        if name not in self._vals:
            fn = self._fns.get(name)
            self._vals[name] = fn().to(self.device) if fn is not None else None
        return self._vals[name]

    def packed(self):
        """names of the forms that have been packed so far"""
        return sorted(k for k, v in self._vals.items() if v is not None)


class Conv3dPlan:
    """Packed weights + epilogue constants of one 3x3x3 convolution, resident on a device."""

    def __getattr__(self, name):                     # self.w_<form>: packed on first use (None when the plan's shape has no such form)
        if name.startswith("w_") and "_packs" in self.__dict__:
            return self._packs.get(name)
        raise AttributeError(name)

    def __init__(self, weight, main_idx, extra_idx, out_idx, n_tiles, scale, shift, act_a="none", act_b=None,
                 act_split=0, head_w=None, head_b=None, device="cuda"):
        self.cin_main = len(main_idx)
        self.n_tiles = n_tiles
        self.n_out = len(out_idx)
        self.has_extra = extra_idx is not None
        weight = weight.detach()
        # Weight forms are packed ON FIRST USE (``self.w_<form>``, ``_Packs``): a plan carries only what the kernels it is actually run on
        # read -- under the default switches one Winograd form per plan, the direct kernel's ``w_main`` only where a launch falls back to it.
        # (Every launch of a captured forward has run eagerly first: GraphedForward warms up before it captures.)
        P = self._packs = _Packs(device)
        P.define("w_main", lambda: packing.pack_conv3d(weight, main_idx, extra_idx, out_idx, n_tiles)[0])
        if extra_idx is not None:
            P.define("w_extra", lambda: packing.pack_conv3d(weight, main_idx, extra_idx, out_idx, n_tiles)[1])
        if n_tiles == 3:
            P.define("w_xout", lambda: packing.pack_xout(weight, main_idx, extra_idx, out_idx[32]))
        splittable = len(main_idx) == 32 and head_w is None and \
            (n_tiles == 2 or (n_tiles == 3 and extra_idx is not None) or (n_tiles == 1 and extra_idx is None))
        # the superseded A/B kernels (bf16 operand split, depth-only Winograd, operand-reuse wino2x): only in a library built with ESTD_BUILD_AB=1
        ab = N.has_ab()
        if splittable and ab:
            P.define("w_split", lambda: packing.pack_conv3d_split(weight, main_idx, out_idx, extra_idx, n_tiles))
        wino_ok = len(main_idx) == 32 and head_w is None and \
            ((n_tiles == 2 and len(out_idx) == 32) or (n_tiles == 3 and len(out_idx) == 33 and extra_idx is not None))
        self.wino_ok = wino_ok
        if wino_ok and ab:
            P.define("w_wino", lambda: packing.pack_conv3d_wino(weight, main_idx, out_idx[:32]))
            if extra_idx is not None:
                P.define("w_wino_extra", lambda: packing.pack_conv3d_wino_extra(weight, extra_idx, out_idx[:32]))
            if n_tiles == 3:
                P.define("w_wino_xout", lambda: packing.pack_conv3d_wino_xout(weight, main_idx, extra_idx, out_idx[32]))
            if n_tiles == 2 and extra_idx is None:
                P.define("w_wino2x", lambda: packing.pack_conv3d_wino2x(weight, main_idx, out_idx[:32]))
        if wino_ok:
            P.define("w_wino2", lambda: packing.pack_conv3d_wino2(weight, main_idx, out_idx[:32]))
            if n_tiles == 2 or (n_tiles == 3 and extra_idx is not None):
                P.define("w_wino3", lambda: packing.pack_conv3d_wino3(weight, main_idx, out_idx[:32]))
                if extra_idx is not None:
                    P.define("w_wino3_extra", lambda: packing.pack_conv3d_wino3_extra(weight, extra_idx, out_idx[:32]))
            if n_tiles == 3 and extra_idx is not None:      # ... + output channel 32 as a pass of its own (csrc/conv3d_xout.hip)
                P.define("w_xout_taps", lambda: packing.pack_conv3d_xout_taps(weight, main_idx, extra_idx, out_idx[32]))
            if extra_idx is not None:
                P.define("w_wino2_extra", lambda: packing.pack_conv3d_wino2_extra(weight, extra_idx, out_idx[:32]))
            if n_tiles == 3:       # 33 -> 33 (dres2): the 33rd output channel of the wino2 kernel's XOUT instance
                P.define("w_wino2_xout", lambda: packing.pack_conv3d_wino2_xout(weight, main_idx, extra_idx, out_idx[32]))
        # 32 -> 16 (the GRU output convolution): the wino2 kernel's 16-output-channel instance
        if len(main_idx) == 32 and n_tiles == 1 and len(out_idx) == 16 and extra_idx is None and head_w is None:
            P.define("w_wino2_o16", lambda: packing.pack_conv3d_wino2(weight, main_idx, out_idx[:16]))
        # 16 -> 16 + 1x1x1 head (the stereo heads): csrc/conv3d_wino2_c16.hip
        if len(main_idx) == 16 and n_tiles == 1 and len(out_idx) == 16 and extra_idx is None and head_w is not None:
            P.define("w_wino2_c16", lambda: packing.pack_conv3d_wino2_c16(weight, main_idx, out_idx[:16]))
        self.scale = scale.float().contiguous().to(device)
        self.shift = shift.float().contiguous().to(device)
        self.act_a = ACT[act_a]
        self.act_b = ACT[act_b if act_b is not None else act_a]
        self.act_split = act_split if act_b is not None else 0
        self.head_w = head_w.float().contiguous().to(device) if head_w is not None else None
        self.head_b = head_b.float().contiguous().to(device) if head_b is not None else None

    def _run_split33(self, x, dims, in_stride, in_extra, out, out_stride, out_extra):
        """the 33 -> 33 instance as two launches: estd_conv3d_k3_wino3 (33 -> 32, the main output channels) + estd_conv3d_k3_xout (33 -> 1)"""
        Nn, D, H, W = dims
        cin = self.cin_main + 1
        vox = float(Nn) * D * H * W
        if _use_torch():
            # (two profile groups: the launches belong to two kernel families of the replay trace)
            with _Prof("conv3d:%d->32" % cin, 2.0 * 27 * cin * 32 * vox):
                T().conv3d_k3(x, in_extra, None, self.w_wino3_extra, None, self.w_wino3, self.scale, self.shift, (Nn, D, H, W), self.cin_main, in_stride, 2,
                              self.act_a, self.act_b, self.act_split, out, out_stride, 32, None, None, 1.0, False, None, None, None, None, None, 5,
                              None, None, None, None)
            with _Prof("conv3d:%d->1" % cin, 2.0 * 27 * cin * vox):
                T().conv3d_k3(x, in_extra, None, self.w_wino3_extra, None, self.w_xout_taps, self.scale, self.shift, (Nn, D, H, W), self.cin_main, in_stride, 3,
                              self.act_a, self.act_b, self.act_split, None, out_stride, 32, None, None, 1.0, False, out_extra, None, None, None, None, 6,
                              None, None, None, None)
            return
        d = N.Conv3dDesc()
        d.N, d.D, d.H, d.W = Nn, D, H, W
        d.cin_main, d.in_stride, d.n_tiles = self.cin_main, in_stride, 2
        d.in_main, d.in_extra = x.data_ptr(), in_extra.data_ptr()
        d.scale, d.shift = self.scale.data_ptr(), self.shift.data_ptr()
        d.act_a, d.act_b, d.act_split = self.act_a, self.act_b, self.act_split
        d.out_main, d.out_stride, d.out_channels = out.data_ptr(), out_stride, 32
        d.out_scale = 1.0
        d.w_wino2, d.w_extra = self.w_wino3.data_ptr(), self.w_wino3_extra.data_ptr()
        with _Prof("conv3d:%d->32" % cin, 2.0 * 27 * cin * 32 * vox):
            N.check(N.lib().estd_conv3d_k3_wino3(ctypes.byref(d), _stream()), "estd_conv3d_k3_wino3")
        d.n_tiles, d.out_main, d.w_wino2, d.w_extra = 3, None, None, None
        d.w_xout, d.out_extra = self.w_xout_taps.data_ptr(), out_extra.data_ptr()
        with _Prof("conv3d:%d->1" % cin, 2.0 * 27 * cin * vox):
            N.check(N.lib().estd_conv3d_k3_xout(ctypes.byref(d), _stream()), "estd_conv3d_k3_xout")

    def with_shift_scaled(self, k):
        """same packed weights, BN shift multiplied by k: sum of k conv+BN results of a LINEAR layer computed as ONE
        convolution of the summed inputs (conv is linear, the shift is counted k times)."""
        import copy
        other = copy.copy(self)
        other.shift = (self.shift * float(k)).contiguous()
        return other

    def run(self, x, dims, in_stride=None, in_extra=None, out=None, out_stride=None, out_channels=None,
            residual=None, residual2=None, out_scale=1.0, accumulate=False, out_extra=None, out_head=None, stats_partials=None, gate=None):
        """x: channels-last volume(s) [N,D,H,W,in_stride] (or a base view of it); dims = (N,D,H,W).
        ``gate`` = (ru [N,D,H,W,32], statistics [4], gamma [16], beta [16]): the ConvGRU's reset gate applied to input channels 16..31 in the
        convolution's own loads (32 -> 16 instance of the two-axis Winograd kernel only; include/estd_hip.h ``gate_r``)."""
        Nn, D, H, W = dims
        if (in_extra is None) == self.has_extra:
            raise RuntimeError("conv3d plan/extra-channel mismatch")
        has = self._packs.has
        if CONV3D_ARITH not in ("f32", "bf16x3"):
            raise RuntimeError("ESTD_CONV3D_ARITH must be f32 or bf16x3, got %r" % (CONV3D_ARITH,))
        in_stride = in_stride if in_stride is not None else self.cin_main
        out_stride = out_stride if out_stride is not None else (16 * min(self.n_tiles, 2))
        out_channels = out_channels if out_channels is not None else 16 * min(self.n_tiles, 2)
        head_w = self.head_w if out_head is not None else None
        head_b = self.head_b if out_head is not None else None
        # instances of the split kernel (csrc/conv3d_split_bf16.hip dispatch): plain [+stats], extra input [tanh|relu], 33 -> 33
        tanh = ACT["tanh"] in ((self.act_a if self.act_split > 0 else self.act_b), self.act_b)
        if self.n_tiles == 3:
            inst = not tanh and stats_partials is None
        elif self.n_tiles == 1:
            inst = not tanh and not self.has_extra
        elif self.has_extra:
            inst = stats_partials is None
        else:
            inst = not tanh
        if CONV3D_ARITH == "bf16x3":
            N.require_ab("ESTD_CONV3D_ARITH=bf16x3 (csrc/conv3d_split_bf16.hip)")
        split = CONV3D_ARITH == "bf16x3" and has("w_split") and out is not None and inst
        if CONV3D_ALGO not in ("wino2", "wino", "direct"):
            raise RuntimeError("ESTD_CONV3D_ALGO must be wino2, wino or direct, got %r" % (CONV3D_ALGO,))
        if CONV3D_ALGO == "wino":
            N.require_ab("ESTD_CONV3D_ALGO=wino (csrc/conv3d_wino.hip)")
        wino_shape = (not split) and CONV3D_ALGO in ("wino", "wino2") and self.wino_ok and out is not None \
            and (out_extra is not None) == (self.n_tiles == 3) and out_head is None and out_channels == 32 \
            and (stats_partials is None or not self.has_extra)
        wino2 = wino_shape and CONV3D_ALGO == "wino2" and has("w_wino2") and (in_extra is None or stats_partials is None)
        if self.n_tiles == 3:                             # the XOUT instance has no read-back streams / statistics (dres2 needs none)
            wino2 = wino2 and W2_XOUT and residual is None and residual2 is None and not accumulate and float(out_scale) == 1.0 and stats_partials is None
        # the depth-only kernel: ESTD_CONV3D_ALGO=wino, or dres2 with ESTD_W2_XOUT=0 -- where the library carries it (else the direct kernel)
        wino = wino_shape and not wino2 and has("w_wino")
        o16 = (not split) and CONV3D_ALGO == "wino2" and has("w_wino2_o16") and out is not None and out_head is None \
            and in_extra is None and out_channels == 16 and out_extra is None
        # the stereo heads: only the head's logit volume leaves the kernel, no tanh
        c16 = (not split) and CONV3D_ALGO == "wino2" and has("w_wino2_c16") and out is None and out_head is not None \
            and in_extra is None and out_extra is None and residual is None and residual2 is None and not accumulate \
            and stats_partials is None and float(out_scale) == 1.0 and not tanh
        # 32 -> 32 without a scalar channel and without tanh: the operand-reuse kernel (GroupNorm partials only without read-back streams)
        wino2x = wino2 and W2X and has("w_wino2x") and in_extra is None and self.n_tiles == 2 and not tanh \
            and (stats_partials is None or (residual is None and residual2 is None and not accumulate and float(out_scale) == 1.0))
        plain_epi = residual is None and residual2 is None and not accumulate and float(out_scale) == 1.0
        wino3 = wino2 and W3 and has("w_wino3") and self.n_tiles == 2 and (stats_partials is None or plain_epi) \
            and (in_extra is None or (W3_EXTRA and has("w_wino3_extra") and plain_epi and stats_partials is None))
        wino2x = wino2x and not wino3
        # 33 -> 33 (dres2): 32 outputs on the three-axis kernel's 33 -> 32 instance, then output channel 32 alone
        split33 = wino2 and self.n_tiles == 3 and W3 and W3_EXTRA and W3_XOUT and has("w_wino3") and has("w_wino3_extra") and has("w_xout_taps") \
            and in_extra is not None and out_extra is not None
        if split33:
            return self._run_split33(x, dims, in_stride, in_extra, out, out_stride, out_extra)
        variant, alt = (1, "w_split") if split else (3, "w_wino2_c16") if c16 else (3, "w_wino2_o16") if o16 else (5, "w_wino3") if wino3 \
            else (4, "w_wino2x") if wino2x \
            else (3, "w_wino2") if wino2 else (2, "w_wino") if wino else (0, None)
        if W2X:
            N.require_ab("ESTD_W2X=1 (csrc/conv3d_wino2x.hip)")
        w_alt = self._packs.get(alt) if alt is not None else None
        if gate is not None and not o16:
            raise RuntimeError("the reset gate is folded into the 32 -> 16 instance of the two-axis Winograd kernel only")
        g_r, g_st, g_ga, g_be = gate if gate is not None else (None, None, None, None)
        cin = self.cin_main + (1 if self.has_extra else 0)
        with _Prof("conv3d:%d->%d" % (cin, self.n_out), 2.0 * 27 * cin * self.n_out * Nn * D * H * W):
            if _use_torch():
                direct = variant in (0, 1)          # (the operand-split kernel reads the direct form's scalar-channel / 33rd-output weights)
                T().conv3d_k3(x, in_extra, self.w_main if direct else None,
                              self.w_wino3_extra if wino3 else self.w_wino2_extra if wino2 else self.w_wino_extra if wino else self.w_extra,
                              self.w_wino2_xout if wino2 else self.w_wino_xout if wino else self.w_xout, w_alt, self.scale, self.shift,
                              (Nn, D, H, W), self.cin_main, in_stride, self.n_tiles, self.act_a, self.act_b, self.act_split, out,
                              out_stride, out_channels, residual, residual2, float(out_scale), bool(accumulate), out_extra, head_w, head_b,
                              out_head, stats_partials, variant, g_r, g_st, g_ga, g_be)
                return
            d = N.Conv3dDesc()
            d.N, d.D, d.H, d.W = Nn, D, H, W
            d.cin_main, d.in_stride, d.n_tiles = self.cin_main, in_stride, self.n_tiles
            d.in_main = x.data_ptr()
            d.in_extra = in_extra.data_ptr() if in_extra is not None else None
            if variant in (0, 1):
                d.w_main = self.w_main.data_ptr()
                d.w_extra = self.w_extra.data_ptr() if self.has_extra else None
                d.w_xout = self.w_xout.data_ptr() if self.n_tiles == 3 else None
            d.scale, d.shift = self.scale.data_ptr(), self.shift.data_ptr()
            d.act_a, d.act_b, d.act_split = self.act_a, self.act_b, self.act_split
            d.out_main = out.data_ptr() if out is not None else None
            d.out_stride, d.out_channels = out_stride, out_channels
            d.residual = residual.data_ptr() if residual is not None else None
            d.residual2 = residual2.data_ptr() if residual2 is not None else None
            d.out_scale = float(out_scale)
            d.accumulate = 1 if accumulate else 0
            d.out_extra = out_extra.data_ptr() if out_extra is not None else None
            d.head_w = head_w.data_ptr() if head_w is not None else None
            d.head_b = head_b.data_ptr() if head_b is not None else None
            d.out_head = out_head.data_ptr() if out_head is not None else None
            d.stats_partials = stats_partials.data_ptr() if stats_partials is not None else None
            if gate is not None:
                d.gate_r, d.gate_stats, d.gate_gamma, d.gate_beta = g_r.data_ptr(), g_st.data_ptr(), g_ga.data_ptr(), g_be.data_ptr()
            if split:
                d.w_split = self.w_split.data_ptr()
                N.check(N.lib().estd_conv3d_k3_split(ctypes.byref(d), _stream()), "estd_conv3d_k3_split")
            elif c16:
                d.w_wino2 = self.w_wino2_c16.data_ptr()
                N.check(N.lib().estd_conv3d_k3_wino2(ctypes.byref(d), _stream()), "estd_conv3d_k3_wino2")
            elif o16:
                d.w_wino2 = self.w_wino2_o16.data_ptr()
                N.check(N.lib().estd_conv3d_k3_wino2(ctypes.byref(d), _stream()), "estd_conv3d_k3_wino2")
            elif wino3:
                d.w_wino2 = self.w_wino3.data_ptr()
                d.w_extra = self.w_wino3_extra.data_ptr() if (in_extra is not None) else None
                N.check(N.lib().estd_conv3d_k3_wino3(ctypes.byref(d), _stream()), "estd_conv3d_k3_wino3")
            elif wino2x:
                d.w_wino2 = self.w_wino2x.data_ptr()
                N.check(N.lib().estd_conv3d_k3_wino2x(ctypes.byref(d), _stream()), "estd_conv3d_k3_wino2x")
            elif wino2:
                d.w_wino2 = self.w_wino2.data_ptr()
                d.w_extra = self.w_wino2_extra.data_ptr() if self.has_extra else None
                d.w_xout = self.w_wino2_xout.data_ptr() if self.n_tiles == 3 else None
                N.check(N.lib().estd_conv3d_k3_wino2(ctypes.byref(d), _stream()), "estd_conv3d_k3_wino2")
            elif wino:
                d.w_wino = self.w_wino.data_ptr()
                d.w_extra = self.w_wino_extra.data_ptr() if self.has_extra else None
                d.w_xout = self.w_wino_xout.data_ptr() if self.n_tiles == 3 else None
                N.check(N.lib().estd_conv3d_k3_wino(ctypes.byref(d), _stream()), "estd_conv3d_k3_wino")
            else:
                N.check(N.lib().estd_conv3d_k3(ctypes.byref(d), _stream()), "estd_conv3d_k3")


class Conv2dPlan:
    """3x3 Conv2d (stride 1, dilation 1|2) + folded BatchNorm2d [+ReLU] [+residual] on NHWC tensors
    (csrc/conv2d_mfma.hip).  ``conv`` / ``bn`` are the torch modules holding the parameters."""

    def __init__(self, conv, bn, relu_before=False, relu_after=False):
        if conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.groups != 1 or conv.bias is not None:
            raise RuntimeError("Conv2dPlan: 3x3 / stride 1 / bias-free convolutions only")
        if conv.dilation not in ((1, 1), (2, 2)) or conv.padding != conv.dilation:
            raise RuntimeError("Conv2dPlan: dilation 1 or 2 with padding = dilation only")
        if conv.in_channels % 32 or conv.out_channels % 32:
            raise RuntimeError("Conv2dPlan: channel counts must be multiples of 32")
        dev = conv.weight.device
        self.cin, self.cout, self.dil = conv.in_channels, conv.out_channels, conv.dilation[0]
        # output channels per work item: 64 (NT = 4: the input brick feeds twice the MFMAs) unless that leaves the 512 resident
        # workgroups badly balanced (e.g. 64 -> 64 on 5 x 120x160: 750 items = 1.46 rounds, 87 TFLOP/s; NT = 2: 1500 items, 103)
        # weight forms packed on first use (see Conv3dPlan): the default path reads w_wino2 only
        w = conv.weight.detach()
        P = self._packs = _Packs(dev)
        self.nts = (2, 4) if self.cout % 64 == 0 else (2,)
        for nt in self.nts:
            P.define("w_nt%d" % nt, lambda nt=nt: packing.pack_conv2d(w, nt))
        if N.has_ab():       # row-only Winograd / bf16 operand split: only in a library built with ESTD_BUILD_AB=1
            P.define("w_split", lambda: packing.pack_conv2d_split(w))
            for nt in self.nts:
                P.define("w_wino_nt%d" % nt, lambda nt=nt: packing.pack_conv2d_wino(w, nt))
        P.define("w_wino2", lambda: packing.pack_conv2d_wino2(w))      # F(2x2, 3x3): csrc/conv2d_wino2.hip (dilation 1 and 2)
        sc, sh = packing.fold_bn_fp32(bn, list(range(self.cout)))
        self.scale, self.shift = sc.to(dev), sh.to(dev)
        self.relu_before, self.relu_after = int(relu_before), int(relu_after)

    def __getattr__(self, name):                     # self.w_<form>: packed on first use (None when the plan has no such form)
        if name.startswith("w_") and "_packs" in self.__dict__:
            return self._packs.get(name)
        raise AttributeError(name)

    def _pick_nt(self, n, h, w):
        if 4 not in self.nts:
            return 2
        if CONV2D_NT in ("2", "4"):               # A/B switch (ESTD_CONV2D_NT): force the work-item width
            return int(CONV2D_NT)
        tiles = n * ((h + 7) // 8) * ((w + 15) // 16)
        def balance(items):                       # fraction of the persistent grid's rounds that does useful work
            return items / (512.0 * ((items + 511) // 512))
        return 2 if 0.93 * balance(tiles * (self.cout // 32)) > balance(tiles * (self.cout // 64)) else 4

    def run(self, x_nhwc, residual=None):
        """x_nhwc [N,H,W,Cin] contiguous -> [N,H,W,Cout]."""
        Nn, H, W, C = x_nhwc.shape
        if C != self.cin or not x_nhwc.is_contiguous():
            raise RuntimeError("Conv2dPlan.run: expected contiguous NHWC input with %d channels" % self.cin)
        if CONV2D_ARITH not in ("f32", "bf16x3"):
            raise RuntimeError("ESTD_CONV2D_ARITH must be f32 or bf16x3, got %r" % (CONV2D_ARITH,))
        if residual is not None and (tuple(residual.shape) != (Nn, H, W, self.cout) or not residual.is_contiguous()):
            raise RuntimeError("Conv2dPlan.run: residual must be contiguous NHWC of the output shape")
        nt = self._pick_nt(Nn, H, W)
        if CONV2D_ARITH == "bf16x3":
            N.require_ab("ESTD_CONV2D_ARITH=bf16x3 (csrc/conv2d_split_bf16.hip)")
        if CONV2D_ALGO == "wino":
            N.require_ab("ESTD_CONV2D_ALGO=wino (csrc/conv2d_wino.hip)")
        has = self._packs.has
        split = CONV2D_ARITH == "bf16x3" and has("w_split")
        if self.dil == 2 and not split and CONV2D_ALGO in ("wino", "wino2"):
            nt = 2        # the 64-channel work item of the dilated Winograd kernel spills registers into its MFMA loop (5x slower)
        if CONV2D_ALGO not in ("wino2", "wino", "direct"):
            raise RuntimeError("ESTD_CONV2D_ALGO must be wino2, wino or direct, got %r" % (CONV2D_ALGO,))
        wino2 = (not split) and CONV2D_ALGO == "wino2" and (self.dil == 1 or C2W2_DIL2)
        wino = (not split) and not wino2 and CONV2D_ALGO in ("wino", "wino2") and has("w_wino_nt%d" % nt)
        variant, alt = (1, "w_split") if split else (3, "w_wino2") if wino2 else (2, "w_wino_nt%d" % nt) if wino else (0, None)
        w_alt = self._packs.get(alt) if alt is not None else None
        w_direct = self._packs.get("w_nt%d" % nt) if variant in (0, 1) else None      # (the operand-split kernel validates d.w as well)
        if _use_torch():
            return T().conv2d_k3(x_nhwc, w_direct, w_alt, self.scale, self.shift, self.cout, self.dil, nt,
                                 bool(self.relu_before), bool(self.relu_after), residual, variant)
        out = torch.empty((Nn, H, W, self.cout), device=x_nhwc.device, dtype=torch.float32)
        d = N.Conv2dDesc()
        d.N, d.H, d.W, d.cin, d.cout, d.dilation, d.group_tiles = Nn, H, W, self.cin, self.cout, self.dil, nt
        d.in_ = _chk(x_nhwc, "conv2d input").data_ptr()
        d.w, d.scale, d.shift = (w_direct.data_ptr() if w_direct is not None else None), self.scale.data_ptr(), self.shift.data_ptr()
        d.relu_before_residual, d.relu_after_residual = self.relu_before, self.relu_after
        d.residual = residual.data_ptr() if residual is not None else None
        d.out = out.data_ptr()
        if split:
            d.w_split = w_alt.data_ptr()
            N.check(N.lib().estd_conv2d_k3_split(ctypes.byref(d), _stream()), "estd_conv2d_k3_split")
        elif wino2:
            d.w_wino = w_alt.data_ptr()
            N.check(N.lib().estd_conv2d_k3_wino2(ctypes.byref(d), _stream()), "estd_conv2d_k3_wino2")
        elif wino:
            d.w_wino = w_alt.data_ptr()
            N.check(N.lib().estd_conv2d_k3_wino(ctypes.byref(d), _stream()), "estd_conv2d_k3_wino")
        else:
            N.check(N.lib().estd_conv2d_k3(ctypes.byref(d), _stream()), "estd_conv2d_k3")
        return out


def conv3d_grid(Nn, D, H, W):
    g = N.lib().estd_conv3d_k3_grid(Nn, D, H, W)
    if g < 0:
        N.check(g, "estd_conv3d_k3_grid")
    return g


def groupnorm_finalize(partials, n_blocks, count, eps=1e-5):
    if _use_torch():
        return T().groupnorm_finalize(partials, n_blocks, float(count), float(eps))
    out = torch.empty(4, device=partials.device, dtype=torch.float32)
    N.check(N.lib().estd_groupnorm_finalize(_p(partials), n_blocks, float(count), float(eps), _p(out), _stream()),
            "estd_groupnorm_finalize")
    return out


# ---------------------------------------------------------------------------------- soft-argmin
def softargmin_up(logits, depth_values, scale=4):
    """logits [N,D,H,W] -> (depth, prob) each [N,1,scale*H,scale*W]."""
    Nn, D, H, W = logits.shape
    with _Prof("softargmin", 4.0 * Nn * H * W * (D + 2 * scale * scale)):       # logits in, depth + prob maps out
        if _use_torch():
            return T().softargmin_up(logits, depth_values, scale)
        depth = torch.empty((Nn, 1, H * scale, W * scale), device=logits.device, dtype=torch.float32)
        prob = torch.empty_like(depth)
        N.check(N.lib().estd_softargmin_up(_p(_chk(logits, "logits")), _p(_chk(depth_values, "depth_values")), _p(depth), _p(prob),
                                           Nn, D, H, W, scale, _stream()), "estd_softargmin_up")
    return depth, prob


# ---------------------------------------------------------------------------------- EST fusion
def warp_volume_cdhw(vol, mats30, depth_values, depth_min, depth_interval):
    if _use_torch():
        return T().warp_volume(vol, mats30, depth_values, float(depth_min), float(depth_interval))
    C, D, H, W = vol.shape
    out = torch.empty_like(vol)
    N.check(N.lib().estd_warp_volume(_p(_chk(vol, "feat_volume")), _p(mats30), _p(_chk(depth_values, "depth")),
                                     float(depth_min), float(depth_interval), _p(out), C, D, H, W, _stream()),
            "estd_warp_volume")
    return out


def warp_volume_ex_cdhw(vol, mats30, depth, depth_per_voxel, depth_min, depth_interval, disp_min=None, disp_interval=None,
                        border=False, padding_value=0.0):
    """every branch of the reference's warp_volume() signature (include/estd_hip.h::estd_warp_volume_ex)."""
    use_disp = disp_min is not None
    # values are passed through as given: a zero interval reaches the native check (ESTD_ERR_ARG / RuntimeError) instead of being
    # replaced silently -- the reference divides by it (homo_utils.py:187-190).  Placeholders only when disparity planes are off.
    dmin_ = float(disp_min) if use_disp else 0.0
    dint_ = float(disp_interval) if (use_disp and disp_interval is not None) else (0.0 if use_disp else 1.0)
    if _use_torch():
        return T().warp_volume_ex(vol, mats30, depth, bool(depth_per_voxel), float(depth_min), float(depth_interval), use_disp,
                                  dmin_, dint_, bool(border), float(padding_value))
    C, D, H, W = vol.shape
    out = torch.empty_like(vol)
    o = N.WarpVolumeOpts(int(bool(depth_per_voxel)), int(use_disp), int(bool(border)), float(depth_min), float(depth_interval),
                         dmin_, dint_, float(padding_value))
    N.check(N.lib().estd_warp_volume_ex(_p(_chk(vol, "feat_volume")), _p(mats30), _p(_chk(depth, "depth")), ctypes.byref(o),
                                        _p(out), C, D, H, W, _stream()), "estd_warp_volume_ex")
    return out


def warp_attention(kv_target, kv_sources, mats, depth_values, depth_min, depth_interval):
    """kv_target [D,H,W,32]; kv_sources list of the same; mats [n,30] -> xh [D,H,W,32] = [V_t | h]."""
    D, H, W, _ = kv_target.shape
    n = len(kv_sources)
    with _Prof("warp_attention", 4.0 * 16 * D * H * W * (2 + 2 * n)):          # K_t, h out, K_j and V_j of every source
        if _use_torch():
            return T().warp_attention(kv_target, list(kv_sources), mats, depth_values, float(depth_min), float(depth_interval))
        arr = (ctypes.c_void_p * n)(*[_chk(k, "kv source").data_ptr() for k in kv_sources])
        xh = torch.empty((D, H, W, 32), device=kv_target.device, dtype=torch.float32)
        N.check(N.lib().estd_warp_attention(_p(_chk(kv_target, "kv target")), arr, _p(_chk(mats, "mats")), n,
                                            _p(_chk(depth_values, "depth_values")), float(depth_min), float(depth_interval),
                                            _p(xh), D, H, W, _stream()), "estd_warp_attention")
    return xh


def attention_prewarped(kv_target, kv_sources):
    if _use_torch():
        return T().attention_prewarped(kv_target, list(kv_sources))
    n = len(kv_sources)
    arr = (ctypes.c_void_p * n)(*[_chk(k, "kv source").data_ptr() for k in kv_sources])
    xh = torch.empty_like(kv_target)
    N.check(N.lib().estd_attention_prewarped(_p(_chk(kv_target, "kv target")), arr, n, _p(xh), kv_target.numel() // 32,
                                             _stream()), "estd_attention_prewarped")
    return xh


def gru_reset_apply(xh, ru, stats4, gamma_r, beta_r):
    n_vox = xh.numel() // 32
    with _Prof("gru_elementwise", 4.0 * 16 * n_vox * 2):       # SURVEY §8d K11+K13 = 5 x 16 channels per voxel: r, h here
        if _use_torch():
            return T().gru_reset_apply(xh, ru, stats4, gamma_r, beta_r)
        xrh = torch.empty_like(xh)
        N.check(N.lib().estd_gru_reset_apply(_p(xh), _p(ru), _p(stats4), _p(gamma_r), _p(beta_r), _p(xrh), n_vox, _stream()),
                "estd_gru_reset_apply")
    return xrh


def gru_blend(xh, ru, o_raw, stats_ru, stats_o, gamma_u, beta_u, gamma_o, beta_o, out_value, out_stride):
    n_vox = xh.numel() // 32
    with _Prof("gru_elementwise", 4.0 * 16 * n_vox * 3):       # ... u, o_raw, out here (h counted once, in the reset pass)
        if _use_torch():
            return T().gru_blend(xh, ru, o_raw, stats_ru, stats_o, gamma_u, beta_u, gamma_o, beta_o, out_value, out_stride)
        N.check(N.lib().estd_gru_blend(_p(xh), _p(ru), _p(o_raw), _p(stats_ru), _p(stats_o), _p(gamma_u), _p(beta_u),
                                       _p(gamma_o), _p(beta_o), _p(out_value), out_stride, n_vox, _stream()), "estd_gru_blend")


# ---------------------------------------------------------------------------------- 2D backbone epilogues
def bn_act_nhwc_(x, scale, shift, relu, residual=None):
    """in place on an NCHW-shaped tensor in channels_last memory: x = act(x*scale[c] + shift[c] (+ residual))."""
    if _use_torch():
        return T().bn_act_nhwc_(x, scale, shift, bool(relu), residual)
    if not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous(memory_format=torch.channels_last):
        raise RuntimeError("bn_act_nhwc_: expected a float32 CUDA tensor in channels_last memory (no CPU path)")
    n, c, h, w = x.shape
    if residual is not None and (residual.shape != x.shape or not residual.is_contiguous(memory_format=torch.channels_last)):
        raise RuntimeError("bn_act_nhwc_: residual must match x (shape, channels_last)")
    N.check(N.lib().estd_bn_act_nhwc(_p(x), _p(scale), _p(shift), _p(residual) if residual is not None else None,
                                     1 if relu else 0, n * h * w, c, _stream()), "estd_bn_act_nhwc")
    return x


def _need_f32_cuda(name, *ts):
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise RuntimeError("%s: contiguous float32 ROCm tensors expected (no CPU path)" % name)


def conv2d_k3_to16_nhwc(x, w_packed, scale, shift, upsample=False):
    """3x3 conv (cin 16|32 -> 16) + folded BN + ReLU on an NHWC map, optionally on its nearest-x2 upsampling (never materialised)."""
    if _use_torch():
        return T().conv2d_k3_to16_nhwc(x, w_packed, scale, shift, bool(upsample))
    _need_f32_cuda("conv2d_k3_to16_nhwc", x, w_packed, scale, shift)
    n, h, w, c = x.shape
    if c not in (16, 32) or w_packed.numel() != 9 * (c // 16) * 256 or scale.numel() != 16 or shift.numel() != 16:
        raise RuntimeError("conv2d_k3_to16_nhwc: NHWC x with 16|32 channels, packed weights [9][cin/16][64][4], scale/shift [16] expected")
    u = 2 if upsample else 1
    out = torch.empty((n, u * h, u * w, 16), device=x.device, dtype=torch.float32)
    N.check(N.lib().estd_conv2d_k3_to16_nhwc(_p(x), _p(w_packed), _p(scale), _p(shift), _p(out), n, u * h, u * w, c, int(bool(upsample)), _stream()),
            "estd_conv2d_k3_to16_nhwc")
    return out


def conv1x1_nhwc(x, w2, scale, shift, stride=1, relu=False, residual=None):
    """1x1 convolution (stride 1|2) + folded BN [+ residual] [+ ReLU] of an NHWC map in one launch (csrc/conv1x1.hip).
    x [N,H,W,cin]; w2 [cout,cin] (the Conv2d weight as it lies); scale / shift [cout] or None -> NHWC [N,Ho,Wo,cout]."""
    if _use_torch():
        return T().conv1x1_nhwc(x, w2, scale, shift, int(stride), bool(relu), residual)
    _need_f32_cuda("conv1x1_nhwc", x, w2)
    n, h, w, c = x.shape
    cout = w2.shape[0]
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    out = torch.empty((n, ho, wo, cout), device=x.device, dtype=torch.float32)
    d = N.Conv1x1Desc()
    d.N, d.H, d.W, d.cin, d.cout, d.stride, d.relu = n, h, w, c, cout, int(stride), int(bool(relu))
    d.in_, d.w = _chk(x, "x").data_ptr(), _chk(w2, "w").data_ptr()
    d.scale = scale.data_ptr() if scale is not None else None
    d.shift = shift.data_ptr() if shift is not None else None
    if residual is not None:
        if tuple(residual.shape) != (n, ho, wo, cout):
            raise RuntimeError("conv1x1_nhwc: residual must be NHWC [%d,%d,%d,%d]" % (n, ho, wo, cout))
        d.residual = _chk(residual, "residual").data_ptr()
    d.out = out.data_ptr()
    N.check(N.lib().estd_conv1x1_nhwc(ctypes.byref(d), _stream()), "estd_conv1x1_nhwc")
    return out


def conv2d_taps_nhwc(x, w_taps, scale, shift, ksize, stride=1, pad=None, relu=False, residual=None):
    """k x k convolution (k = 1|3|5, stride 1|2, zero padding) + folded BN [+ residual] [+ ReLU] of an NHWC map in one launch
    (csrc/conv2d_taps.hip).  x [N,H,W,cin]; w_taps [k*k,cout,cin] (packing.pack_conv2d_taps) -> NHWC [N,Ho,Wo,cout]."""
    pad = ksize // 2 if pad is None else pad
    if _use_torch():
        return T().conv2d_taps_nhwc(x, w_taps, scale, shift, int(ksize), int(stride), int(pad), bool(relu), residual)
    _need_f32_cuda("conv2d_taps_nhwc", x, w_taps)
    n, h, w, c = x.shape
    if w_taps.dim() != 3 or w_taps.shape[0] != ksize * ksize or w_taps.shape[2] != c:
        raise RuntimeError("conv2d_taps_nhwc: NHWC x [N,H,W,cin] and w [k*k,cout,cin] expected")
    cout = w_taps.shape[1]
    ho, wo = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    out = torch.empty((n, ho, wo, cout), device=x.device, dtype=torch.float32)
    d = N.Conv2dTapsDesc()
    d.N, d.H, d.W, d.cin, d.cout = n, h, w, c, cout
    d.ksize, d.stride, d.pad, d.relu = int(ksize), int(stride), int(pad), int(bool(relu))
    d.in_, d.w = _chk(x, "x").data_ptr(), _chk(w_taps, "w").data_ptr()
    d.scale = scale.data_ptr() if scale is not None else None
    d.shift = shift.data_ptr() if shift is not None else None
    if residual is not None:
        if tuple(residual.shape) != (n, ho, wo, cout):
            raise RuntimeError("conv2d_taps_nhwc: residual must be NHWC [%d,%d,%d,%d]" % (n, ho, wo, cout))
        d.residual = _chk(residual, "residual").data_ptr()
    d.out = out.data_ptr()
    N.check(N.lib().estd_conv2d_taps_nhwc(ctypes.byref(d), _stream()), "estd_conv2d_taps_nhwc")
    return out


def stem7x7s2_nhwc(x, w_packed, scale, shift):
    """Conv2d(3, 64, 7, stride 2, padding 3) + folded BatchNorm2d + ReLU on an NHWC image batch [N,H,W,3] -> [N,Ho,Wo,64]
    (torchvision ResNet conv1 / bn1 / relu, hybrid_models/resnet_encoder.py:42-44); w_packed: packing.pack_stem7x7."""
    if _use_torch():
        return T().stem7x7s2_nhwc(x, w_packed, scale, shift)
    _need_f32_cuda("stem7x7s2_nhwc", x, w_packed, scale, shift)
    n, h, w, c = x.shape
    if c != 3 or w_packed.numel() != 7 * 6 * 4 * 64 or scale.numel() != 64 or shift.numel() != 64:
        raise RuntimeError("stem7x7s2_nhwc: NHWC x [N,H,W,3], packed weights [7,6,4,64], scale/shift [64] expected")
    out = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, 64), device=x.device, dtype=torch.float32)
    N.check(N.lib().estd_stem7x7s2_nhwc(_p(x), _p(w_packed), _p(scale), _p(shift), _p(out), n, h, w, _stream()), "estd_stem7x7s2_nhwc")
    return out


def maxpool3x3s2_nhwc(x):
    """MaxPool2d(3, 2, 1) of an NHWC map [N,H,W,C] (C % 4 == 0) -> [N,(H-1)//2+1,(W-1)//2+1,C]."""
    if _use_torch():
        return T().maxpool3x3s2_nhwc(x)
    _need_f32_cuda("maxpool3x3s2_nhwc", x)
    n, h, w, c = x.shape
    out = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c), device=x.device, dtype=torch.float32)
    N.check(N.lib().estd_maxpool3x3s2_nhwc(_p(x), _p(out), n, h, w, c, _stream()), "estd_maxpool3x3s2_nhwc")
    return out


def avgpool_nhwc(x, k):
    """AvgPool2d(k, k) of an NHWC map [N,H,W,C] (C % 4 == 0) -> [N,H//k,W//k,C]."""
    if _use_torch():
        return T().avgpool_nhwc(x, int(k))
    _need_f32_cuda("avgpool_nhwc", x)
    n, h, w, c = x.shape
    out = torch.empty((n, h // k, w // k, c), device=x.device, dtype=torch.float32)
    N.check(N.lib().estd_avgpool_nhwc(_p(x), _p(out), n, h, w, c, int(k), _stream()), "estd_avgpool_nhwc")
    return out


SMALL_CONV_SHAPES = {(32, 3, 2), (32, 1, 2), (32, 1, 1), (64, 1, 1), (128, 1, 1)}      # (cin, ksize, stride) instances of estd_conv2d_small_nhwc


def conv2d_small_nhwc(x, w_packed, scale, shift, cout, ksize, stride, relu):
    """small PSM convolutions (3x3 stride 2, 1x1 stride 1|2) + folded BN [+ ReLU] on an NHWC map -> NHWC [N,Ho,Wo,cout]."""
    if _use_torch():
        return T().conv2d_small_nhwc(x, w_packed, scale, shift, int(cout), int(ksize), int(stride), bool(relu))
    _need_f32_cuda("conv2d_small_nhwc", x, w_packed, scale, shift)
    n, h, w, c = x.shape
    pad = ksize // 2
    ho, wo = (h + 2 * pad - ksize) // stride + 1, (w + 2 * pad - ksize) // stride + 1
    out = torch.empty((n, ho, wo, cout), device=x.device, dtype=torch.float32)
    N.check(N.lib().estd_conv2d_small_nhwc(_p(x), _p(w_packed), _p(scale), _p(shift), _p(out), n, h, w, c, int(cout), int(ksize), int(stride),
                                           int(bool(relu)), _stream()), "estd_conv2d_small_nhwc")
    return out


def normalise_nhwc(imgs):
    """[N,3,H,W] images in 0..255 -> 2 * (imgs / 255) - 1 as an NHWC batch [N,H,W,3] (model_hybrid.py:119)."""
    if _use_torch():
        return T().normalise_nhwc(imgs)
    _need_f32_cuda("normalise_nhwc", imgs)
    n, c, h, w = imgs.shape
    if c != 3:
        raise RuntimeError("normalise_nhwc: [N,3,H,W] images expected")
    out = torch.empty((n, h, w, 3), device=imgs.device, dtype=torch.float32)
    N.check(N.lib().estd_normalise_nhwc(_p(imgs), _p(out), n, h * w, _stream()), "estd_normalise_nhwc")
    return out


def stem3x3s2_nhwc(x, weight, scale, shift):
    """Conv2d(3, 32, 3, stride 2, padding 1) + folded BatchNorm2d + ReLU on an NHWC image batch [N,H,W,3] -> [N,Ho,Wo,32]
    (networks/psm_submodule.py:47)."""
    if _use_torch():
        return T().stem3x3s2_nhwc(x, weight, scale, shift)
    _need_f32_cuda("stem3x3s2_nhwc", x, weight, scale, shift)
    n, h, w, c = x.shape
    if c != 3 or tuple(weight.shape) != (32, 3, 3, 3) or scale.numel() != 32 or shift.numel() != 32:
        raise RuntimeError("stem3x3s2_nhwc: NHWC x [N,H,W,3], weight [32,3,3,3], scale/shift [32] expected")
    out = torch.empty((n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, 32), device=x.device, dtype=torch.float32)
    N.check(N.lib().estd_stem3x3s2_nhwc(_p(x), _p(weight), _p(scale), _p(shift), _p(out), n, h, w, _stream()), "estd_stem3x3s2_nhwc")
    return out


def nhwc_to_planes(x):
    """NHWC map [N,H,W,C] -> contiguous NCHW planes [N,C,H,W] (hybrid_depth_decoder.py:162-184: the plane scores as scalar volumes)."""
    if _use_torch():
        return T().nhwc_to_planes(x)
    _need_f32_cuda("nhwc_to_planes", x)
    n, h, w, c = x.shape
    out = torch.empty((n, c, h, w), device=x.device, dtype=torch.float32)
    N.check(N.lib().estd_nhwc_to_planes(_p(x), c, _p(out), n, h * w, _stream()), "estd_nhwc_to_planes")
    return out


def planes_cat_nhwc(a, b, relu_b=False):
    """torch.cat([a, relu?(b)], 1) of NCHW stacks -> NHWC map [N,H,W,Ca+Cb] (hybrid_depth_decoder.py:268)."""
    if _use_torch():
        return T().planes_cat_nhwc(a, b, bool(relu_b))
    _need_f32_cuda("planes_cat_nhwc", a, b)
    n, ca, h, w = a.shape
    cb = b.shape[1]
    if tuple(b.shape) != (n, cb, h, w):
        raise RuntimeError("planes_cat_nhwc: two NCHW stacks of the same N, H, W expected")
    out = torch.empty((n, h, w, ca + cb), device=a.device, dtype=torch.float32)
    N.check(N.lib().estd_planes_cat_nhwc(_p(a), ca, _p(b), cb, int(bool(relu_b)), _p(out), n, h * w, _stream()), "estd_planes_cat_nhwc")
    return out


def upsample2_cat_nhwc(x, skip):
    """torch.cat([nearest_x2(x), skip], 1) on NHWC maps: x [N,H/2,W/2,Cx], skip [N,H,W,Cs] -> [N,H,W,Cx+Cs] (:269-272)."""
    if _use_torch():
        return T().upsample2_cat_nhwc(x, skip)
    _need_f32_cuda("upsample2_cat_nhwc", x, skip)
    n, h, w, cs = skip.shape
    if tuple(x.shape[:3]) != (n, h // 2, w // 2) or h % 2 or w % 2:
        raise RuntimeError("upsample2_cat_nhwc: NHWC x [N,H/2,W/2,Cx] and skip [N,H,W,Cs] expected")
    cx = x.shape[3]
    out = torch.empty((n, h, w, cx + cs), device=x.device, dtype=torch.float32)
    N.check(N.lib().estd_upsample2_cat_nhwc(_p(x), cx, _p(skip), cs, _p(out), n, h, w, _stream()), "estd_upsample2_cat_nhwc")
    return out


def disp_head_nhwc(x, weight, bias, depth_max, upscale=1):
    """depth_max * sigmoid(Conv2d(C,1,3,padding=1,bias)(x)) on an NHWC map, optionally nearest x2 -> [N,1,uH,uW] (:274, :279)."""
    if _use_torch():
        return T().disp_head_nhwc(x, weight, bias, float(depth_max), int(upscale))
    _need_f32_cuda("disp_head_nhwc", x, weight, bias)
    n, h, w, c = x.shape
    if tuple(weight.shape) != (1, c, 3, 3) or bias.numel() != 1:
        raise RuntimeError("disp_head_nhwc: NHWC x [N,H,W,C], weight [1,C,3,3], bias [1] expected")
    out = torch.empty((n, 1, upscale * h, upscale * w), device=x.device, dtype=torch.float32)
    N.check(N.lib().estd_disp_head_nhwc(_p(x), _p(weight), _p(bias), float(depth_max), _p(out), n, h, w, c, int(upscale), _stream()),
            "estd_disp_head_nhwc")
    return out


def spp_upsample_cat(raw, skip, branches):
    """NHWC tensors: raw [N,H,W,Cr], skip [N,H,W,Cs], branches [N,hk,wk,Cb] -> [N,H,W,Cr+Cs+nb*Cb] =
    cat(raw, skip, bilinear_up(branches...)) in one pass (psm_submodule.py:100-116)."""
    if _use_torch():
        return T().spp_upsample_cat(raw, skip, list(branches))
    n, h, w, cr = raw.shape
    cs, cb, nb = skip.shape[3], branches[0].shape[3], len(branches)
    for t in [raw, skip] + list(branches):
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("spp_upsample_cat: contiguous float32 NHWC CUDA tensors expected (no CPU path)")
    out = torch.empty((n, h, w, cr + cs + nb * cb), device=raw.device, dtype=torch.float32)
    arr = (ctypes.c_void_p * nb)(*[b.data_ptr() for b in branches])
    bh = (ctypes.c_int * nb)(*[b.shape[1] for b in branches])
    bw = (ctypes.c_int * nb)(*[b.shape[2] for b in branches])
    N.check(N.lib().estd_spp_upsample_cat(_p(raw), cr, _p(skip), cs, arr, bh, bw, nb, cb, _p(out), n, h, w, _stream()),
            "estd_spp_upsample_cat")
    return out


# ---------------------------------------------------------------------------------- layout converters
def cdhw_to_vol(src, dst, dst_stride, dst_off):
    """src [C,D,H,W] contiguous -> channels dst_off.. of the channels-last records of dst."""
    if _use_torch():
        return T().cdhw_to_vol(src, dst, dst_stride, dst_off)
    C = src.shape[0]
    S = src.numel() // C
    N.check(N.lib().estd_cdhw_to_vol(_p(_chk(src, "volume")), _p(dst), C, S, dst_stride, dst_off, _stream()), "estd_cdhw_to_vol")


def vol_to_cdhw(src, C, dims, src_stride, src_off):
    if _use_torch():
        return T().vol_to_cdhw(src, C, list(dims), src_stride, src_off)
    D, H, W = dims
    out = torch.empty((C, D, H, W), device=src.device, dtype=torch.float32)
    N.check(N.lib().estd_vol_to_cdhw(_p(src), _p(out), C, D * H * W, src_stride, src_off, _stream()), "estd_vol_to_cdhw")
    return out
