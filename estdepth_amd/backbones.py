"""2D networks around the hot path (NOT hand-written kernels: they run on PyTorch-ROCm / MIOpen).

SURVEY.md §2 marks these OUT OF SCOPE for HIP kernels; they exist so that
``DepthNetHybrid.forward`` is a complete drop-in and so that reference checkpoints load
unchanged (parameter names must match the reference's state dict):

  * ``PSMFeatures``   <-> networks/psm_submodule.py:40-116  (keys ``matchingFeature.*``)
  * ``ResNetTrunk`` / ``SemanticEncoder`` <-> hybrid_models/resnet_encoder.py:17-51 over a
    torchvision-layout ResNet (keys ``semanticFeature.encoder.*`` incl. the unused ``fc``)
  * ``conv_bn2d`` / ``UpBlock`` <-> networks/layers_op.py:10-27, hybrid_depth_decoder.py:17-30
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


SPP_FUSED = os.environ.get("ESTD_SPP_FUSED", "1") == "1"      # A/B switch for the fused upsample+concat of the PSM SPP tail


def conv_bn2d(cin, cout, k, stride, pad, dilation):
    """Conv2d(bias=False)+BatchNorm2d; padding = dilation when dilation > 1 (layers_op.py:10-13)."""
    return nn.Sequential(
        nn.Conv2d(cin, cout, k, stride, dilation if dilation > 1 else pad, dilation, bias=False),
        nn.BatchNorm2d(cout))


def _folded(bn):
    """(scale, shift) of an eval BatchNorm2d on its device, cached on the module until a parameter changes."""
    from . import packing
    key = (bn.weight.device, bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr())
    c = bn.__dict__.get("_estd_fold")
    if c is None or c[0] != key:
        sc, sh = packing.fold_bn_fp32(bn, list(range(bn.num_features)))
        c = (key, sc.to(bn.weight.device), sh.to(bn.weight.device))
        bn.__dict__["_estd_fold"] = c
    return c[1], c[2]


def fused_on(mod, x):
    """the fused BatchNorm/ReLU/residual epilogue (estd_bn_act_nhwc) applies: opted in, inference, on the GPU."""
    return mod.__dict__.get("_fuse_bn", False) and x.is_cuda and not mod.training


def conv1x1_gemm(conv, x):
    """1x1 convolution of an NHWC map as ONE library GEMM [pixels, Cin] x [Cin, Cout] (rocBLAS / hipBLASLt fp32): 1.3-2.5x
    faster than MIOpen's convolution kernels for the ResNet-50 / PSM shapes (tools/r50_conv_bench.py); stride-2 convs
    gather their pixels first.  Returns an NCHW-shaped tensor in channels_last memory."""
    key = (conv.weight.device, conv.weight._version, conv.weight.data_ptr())
    c = conv.__dict__.get("_estd_w2")
    if c is None or c[0] != key:
        c = (key, conv.weight.detach().reshape(conv.out_channels, conv.in_channels).t().contiguous())
        conv.__dict__["_estd_w2"] = c
    if conv.stride != (1, 1):
        x = x[:, :, ::conv.stride[0], ::conv.stride[1]]
    x = x.contiguous(memory_format=torch.channels_last)
    n, cin, h, w = x.shape
    y = torch.mm(x.permute(0, 2, 3, 1).reshape(n * h * w, cin), c[1])
    if conv.bias is not None:
        y = y + conv.bias
    return y.view(n, h, w, conv.out_channels).permute(0, 3, 1, 2)


HIP_STEM = os.environ.get("ESTD_HIP_STEM", "1") == "1"       # A/B switch: PSM first layer on csrc/refine2d.hip
HIP_STEM7 = os.environ.get("ESTD_HIP_STEM7", "1") == "1"     # A/B switch: the semantic ResNet's 7x7 stem on csrc/conv2d_taps.hip
HIP_POOL = os.environ.get("ESTD_HIP_POOL", "1") == "1"       # A/B switch: max / average pooling of the 2D branches on csrc/conv2d_taps.hip
GEMM_EPILOGUE = os.environ.get("ESTD_GEMM_EPILOGUE", "1") == "1"     # A/B switch, read once at import
# 1x1 convolutions of the fused-BN path on csrc/conv1x1.hip (conv + BN + residual + ReLU in one launch): "all" (default) = every 1x1
# convolution with cin % 16 == 0 and cout % 32 == 0 -- no library GEMM and no separate BatchNorm pass left in the ResNet branch's 1x1 layers;
# "auto" = only where that kernel beats the library GEMM (+ BN pass) stand-alone: every convolution with a residual, the stride-2
# downsample convolutions up to 512 input channels, 64 -> 64; "0" = library only.  tools/conv1x1_bench.py, profiles/r4_conv1x1_bench.txt:
# the kernel streams its operands from L2 without LDS reuse (L2-bandwidth bound at ~90 TFLOP/s); with the K ranges of the long-K /
# small-map layers split over the four waves of a workgroup the sixteen ResNet-50 shapes sum to 539 us against the library's 551, and
# "all" and "auto" time the same in the step (18.56 / 18.61 vs 18.55 / 18.59 ms; "0": 18.62).
HIP_1X1 = os.environ.get("ESTD_HIP_1X1", "all")


def _hip_1x1_wanted(conv, residual):
    if HIP_1X1 == "all" or HIP_1X1 == "1":
        return True
    if HIP_1X1 != "auto":
        return False
    return residual is not None or (conv.stride == (2, 2) and conv.in_channels <= 512) or (conv.in_channels <= 64 and conv.out_channels <= 64)


def conv1x1_bn_gemm(conv, bn, x, relu):
    """1x1 convolution + BatchNorm2d(eval) [+ ReLU] as ONE library GEMM with a bias [+ ReLU] epilogue: the BN scale is folded
    into the weight columns, the BN shift is the bias."""
    key = (conv.weight.device, conv.weight._version, conv.weight.data_ptr(), bn.weight._version, bn.bias._version,
           bn.running_mean._version, bn.running_var._version)
    c = conv.__dict__.get("_estd_w2bn")
    if c is None or c[0] != key:
        sc, sh = _folded(bn)
        w2 = conv.weight.detach().reshape(conv.out_channels, conv.in_channels).t() * sc[None, :]
        c = (key, w2.contiguous(), sh.contiguous())
        conv.__dict__["_estd_w2bn"] = c
    if conv.stride != (1, 1):
        x = x[:, :, ::conv.stride[0], ::conv.stride[1]]
    x = x.contiguous(memory_format=torch.channels_last)
    n, cin, h, w = x.shape
    x2 = x.permute(0, 2, 3, 1).reshape(n * h * w, cin)
    y = torch._addmm_activation(c[2], x2, c[1], use_gelu=False) if relu else torch.addmm(c[2], x2, c[1])
    return y.view(n, h, w, conv.out_channels).permute(0, 3, 1, 2)


# k x k convolutions outside the tiled kernels (stride 2, or maps with too few tiles) on csrc/conv2d_taps.hip: "1" (default) | "0" = library (A/B)
HIP_TAPS = os.environ.get("ESTD_HIP_TAPS", "1") == "1"


def _fits_32bit(x, conv):
    """the in-house NHWC kernels address a map with 32-bit byte offsets (buffer descriptors): N*H*W*C*4 of the wider side below 2 GiB;
    larger maps stay on the library path instead of failing with ESTD_ERR_UNSUPPORTED"""
    n, _, h, w = x.shape
    ho, wo = (h + conv.stride[0] - 1) // conv.stride[0], (w + conv.stride[1] - 1) // conv.stride[1]
    return max(n * h * w * conv.in_channels, n * ho * wo * conv.out_channels) * 4 < 0x7fffff00


def _hip_taps_ok(conv):
    k = conv.kernel_size[0]
    return HIP_TAPS and conv.kernel_size == (k, k) and k in (3, 5) and conv.stride in ((1, 1), (2, 2)) and conv.dilation == (1, 1) \
        and conv.groups == 1 and conv.bias is None and conv.padding[0] == conv.padding[1] and conv.in_channels % 16 == 0 \
        and conv.out_channels % 32 == 0 and conv.weight.is_cuda and conv.padding_mode == "zeros"


def _is_1x1(conv):
    return conv.kernel_size == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 and conv.dilation == (1, 1)


HIP_3X3_MIN_ITEMS = int(os.environ.get("ESTD_HIP3X3_MIN_ITEMS", "128"))     # work items (8x16-pixel tiles x 32-channel groups) below which the persistent MFMA kernel cannot fill 256 CUs


def _hip_3x3_plan(conv, bn, x, relu, has_residual):
    """Conv2dPlan (csrc/conv2d_mfma.hip: 3x3 / stride 1 / folded BN / ReLU / residual in ONE kernel) for a library-shaped
    Conv2d + BatchNorm2d pair when the module opted in (enable_hip_3x3) and the map offers enough tiles; else None.
    ResNet-50 stride-1 3x3 convolutions at 120x160 ... 30x40: 1.4-1.6x MIOpen's fp32 kernels (tools/r50_conv_bench.py)."""
    if not conv.__dict__.get("_hip3x3", False) or conv.kernel_size != (3, 3) or conv.stride != (1, 1) or conv.groups != 1:
        return None
    if conv.bias is not None or conv.dilation not in ((1, 1), (2, 2)) or conv.padding != conv.dilation:
        return None
    if conv.in_channels % 32 or conv.out_channels % 32:
        return None
    n, _, h, w = x.shape
    if n * ((h + 7) // 8) * ((w + 15) // 16) * (conv.out_channels // 32) < HIP_3X3_MIN_ITEMS:
        return None
    from . import ops
    key = (conv.weight.device, conv.weight._version, conv.weight.data_ptr(), bn.weight._version, bn.bias._version,
           bn.running_mean._version, bn.running_var._version, relu, has_residual)
    c = conv.__dict__.get("_estd_plan3x3")
    if c is None or c[0] != key:
        # epilogue order of the kernel: BN -> [ReLU] -> [+ residual] -> [ReLU]; with a residual the ReLU comes after the add
        c = (key, ops.Conv2dPlan(conv, bn, relu_before=relu and not has_residual, relu_after=relu and has_residual))
        conv.__dict__["_estd_plan3x3"] = c
    return c[1]


def enable_hip_3x3(root, enable=True):
    """Opt-in: stride-1 3x3 Conv2d+BN(+ReLU)(+residual) pairs under ``root`` that reach conv_bn_act run on the MFMA conv2d
    kernel where the map is large enough (ResNet encoder)."""
    for m in root.modules():
        if isinstance(m, nn.Conv2d):
            m.__dict__["_hip3x3"] = bool(enable)
    return root


def conv_bn_act(conv, bn, x, relu, residual=None):
    """library convolution (a GEMM for 1x1 kernels), then ONE in-place NHWC pass for BatchNorm2d(eval) [+ residual] [+ ReLU];
    or, for opted-in stride-1 3x3 convolutions on large enough maps, everything in the MFMA conv2d kernel."""
    from . import ops
    plan = _hip_3x3_plan(conv, bn, x, relu, residual is not None)
    if plan is not None:
        xn = x.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        rn = residual.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1) if residual is not None else None
        return plan.run(xn, residual=rn).permute(0, 3, 1, 2)
    if _is_1x1(conv) and conv.bias is None and conv.stride in ((1, 1), (2, 2)) and conv.in_channels % 16 == 0 \
            and conv.out_channels % 32 == 0 and _hip_1x1_wanted(conv, residual) and _fits_32bit(x, conv):
        # conv + BN [+ residual] [+ ReLU] in ONE launch of csrc/conv1x1.hip (the ResNet bottlenecks' conv1 / conv3 / downsample)
        key = (conv.weight.device, conv.weight._version, conv.weight.data_ptr())
        c = conv.__dict__.get("_estd_w1x1")
        if c is None or c[0] != key:
            c = (key, conv.weight.detach().reshape(conv.out_channels, conv.in_channels).contiguous())
            conv.__dict__["_estd_w1x1"] = c
        sc, sh = _folded(bn)
        xn = x.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        rn = residual.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1) if residual is not None else None
        return ops.conv1x1_nhwc(xn, c[1], sc, sh, conv.stride[0], relu, rn).permute(0, 3, 1, 2)
    if _hip_taps_ok(conv) and _fits_32bit(x, conv):
        # k x k, stride 1 | 2 (the stride-2 3x3 convolutions of layer2..4, 3x3 convolutions on maps too small for the tiled kernels):
        # conv + BN [+ residual] [+ ReLU] in ONE launch of csrc/conv2d_taps.hip
        from . import packing
        key = (conv.weight.device, conv.weight._version, conv.weight.data_ptr())
        c = conv.__dict__.get("_estd_wtaps")
        if c is None or c[0] != key:
            c = (key, packing.pack_conv2d_taps(conv.weight).to(conv.weight.device))
            conv.__dict__["_estd_wtaps"] = c
        sc, sh = _folded(bn)
        xn = x.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        rn = residual.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1) if residual is not None else None
        return ops.conv2d_taps_nhwc(xn, c[1], sc, sh, conv.kernel_size[0], conv.stride[0], conv.padding[0], relu, rn).permute(0, 3, 1, 2)
    if GEMM_EPILOGUE and residual is None and _is_1x1(conv) and conv.bias is None:
        return conv1x1_bn_gemm(conv, bn, x, relu)
    y = conv1x1_gemm(conv, x) if _is_1x1(conv) else conv(x)
    if not y.is_contiguous(memory_format=torch.channels_last):
        y = y.contiguous(memory_format=torch.channels_last)
    if residual is not None and not residual.is_contiguous(memory_format=torch.channels_last):
        residual = residual.contiguous(memory_format=torch.channels_last)
    sc, sh = _folded(bn)
    return ops.bn_act_nhwc_(y, sc, sh, relu, residual)


HIP_SMALL = os.environ.get("ESTD_HIP_SMALL_CONVS", "1") == "1"     # A/B switch: the PSM extractor's stride-2 / 1x1 convolutions on csrc/refine2d.hip


def small_conv_nhwc(conv, bn, x_nhwc, relu):
    """Conv2d (3x3 stride 2 or 1x1) [+ BatchNorm2d(eval)] [+ ReLU] of the PSM extractor on csrc/refine2d.hip::conv2d_small_kernel
    (psm_submodule.py:52,:72-74,:78-83,:100-110), NHWC in -> NHWC out; None when the shape has no instance (the caller then
    takes the library path)."""
    from . import ops, packing
    k, s = conv.kernel_size[0], conv.stride[0]
    if not HIP_SMALL or conv.bias is not None or conv.groups != 1 or conv.dilation != (1, 1) or conv.kernel_size != (k, k) \
            or conv.stride != (s, s) or conv.padding != (k // 2, k // 2) or (conv.in_channels, k, s) not in ops.SMALL_CONV_SHAPES \
            or conv.out_channels % 16 or not x_nhwc.is_cuda:
        return None
    key = (conv.weight.device, conv.weight._version, conv.weight.data_ptr()) + \
        ((bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version) if bn is not None else ())
    c = conv.__dict__.get("_estd_small")
    if c is None or c[0] != key:
        dev = conv.weight.device
        if bn is not None:
            sc, sh = _folded(bn)
        else:
            sc, sh = torch.ones(conv.out_channels, device=dev), torch.zeros(conv.out_channels, device=dev)
        c = (key, packing.pack_conv2d_small(conv.weight).to(dev), sc.float().contiguous(), sh.float().contiguous())
        conv.__dict__["_estd_small"] = c
    return ops.conv2d_small_nhwc(x_nhwc.contiguous(), c[1], c[2], c[3], conv.out_channels, k, s, relu)


def enable_fused_bn(root, enable=True):
    """Opt-in for every 2D block under ``root``: BatchNorm2d -> (add) -> ReLU after a library convolution become
    one estd_bn_act_nhwc launch.  Needs channels_last activations (DepthNetHybrid.use_channels_last_2d)."""
    for m in root.modules():
        if isinstance(m, (_PSMBlock, PSMFeatures, _Basic, _Bottleneck, SemanticEncoder, UpBlock)):
            m.__dict__["_fuse_bn"] = bool(enable)
    return root


class _PSMBlock(nn.Module):
    def __init__(self, cin, cout, stride, downsample, pad, dilation):
        super().__init__()
        self.conv1 = nn.Sequential(conv_bn2d(cin, cout, 3, stride, pad, dilation), nn.ReLU(inplace=True))
        self.conv2 = conv_bn2d(cout, cout, 3, 1, pad, dilation)
        self.downsample = downsample

    def shortcut(self, x):
        if self.downsample is None:
            return x
        if fused_on(self, x):
            return conv_bn_act(self.downsample[0], self.downsample[1], x, relu=False)
        return self.downsample(x)

    def first(self, x):
        if fused_on(self, x):
            return conv_bn_act(self.conv1[0][0], self.conv1[0][1], x, relu=True)
        return self.conv1(x)

    def forward(self, x):
        if fused_on(self, x):
            return conv_bn_act(self.conv2[0], self.conv2[1], self.first(x), relu=False, residual=self.shortcut(x))
        y = self.conv2(self.conv1(x))
        return y + (x if self.downsample is None else self.downsample(x))


class PSMFeatures(nn.Module):
    """PSMNet feature extractor with SPP -> [N,32,H/4,W/4], no final BN/ReLU."""

    def __init__(self):
        super().__init__()
        self._cin = 32
        self.firstconv = nn.Sequential(
            conv_bn2d(3, 32, 3, 2, 1, 1), nn.ReLU(inplace=True),
            conv_bn2d(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True),
            conv_bn2d(32, 32, 3, 1, 1, 1), nn.ReLU(inplace=True))
        self.layer1 = self._stage(32, 3, 1, 1, 1)
        self.layer2 = self._stage(64, 16, 2, 1, 1)
        self.layer3 = self._stage(128, 3, 1, 1, 1)
        self.layer4 = self._stage(128, 3, 1, 1, 2)
        for i, p in zip((1, 2, 3, 4), (32, 16, 8, 4)):
            setattr(self, "branch%d" % i, nn.Sequential(
                nn.AvgPool2d((p, p), stride=(p, p)), conv_bn2d(128, 32, 1, 1, 0, 1), nn.ReLU(inplace=True)))
        self.lastconv = nn.Sequential(conv_bn2d(320, 128, 3, 1, 1, 1), nn.ReLU(inplace=True),
                                      nn.Conv2d(128, 32, 1, 1, 0, bias=False))
        self.out_channels = [32]

    def _stage(self, planes, blocks, stride, pad, dilation):
        down = None
        if stride != 1 or self._cin != planes:
            down = nn.Sequential(nn.Conv2d(self._cin, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
        layers = [_PSMBlock(self._cin, planes, stride, down, pad, dilation)]
        self._cin = planes
        layers += [_PSMBlock(planes, planes, 1, None, pad, dilation) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    # ---- HIP path: every 3x3 / stride-1 conv+BN(+ReLU)(+residual) on csrc/conv2d_mfma.hip ----
    def use_hip_convs(self, enable=True):
        """Opt-in.  Stride-2 convs, 1x1 convs, pooling and upsampling stay on PyTorch-ROCm; activations travel as
        NHWC (channels_last) tensors so both sides share buffers without copies."""
        self._hip = bool(enable)
        self._hip_plans = None
        if enable:
            self.to(memory_format=torch.channels_last)
        return self

    def _plans(self):
        from . import ops
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if getattr(self, "_hip_plans", None) is None or self._hip_plans[0] != key:
            plans = {}
            fc = self.firstconv
            plans["first1"] = ops.Conv2dPlan(fc[2][0], fc[2][1], relu_before=True)
            plans["first2"] = ops.Conv2dPlan(fc[4][0], fc[4][1], relu_before=True)
            for lname in ("layer1", "layer2", "layer3", "layer4"):
                for bi, blk in enumerate(getattr(self, lname)):
                    c1 = blk.conv1[0]
                    if c1[0].stride == (1, 1):
                        plans[(lname, bi, 1)] = ops.Conv2dPlan(c1[0], c1[1], relu_before=True)
                    plans[(lname, bi, 2)] = ops.Conv2dPlan(blk.conv2[0], blk.conv2[1])       # + residual, no ReLU (psm_submodule.py:26-37)
            plans["last"] = ops.Conv2dPlan(self.lastconv[0][0], self.lastconv[0][1], relu_before=True)
            self._hip_plans = (key, plans)
        return self._hip_plans[1]

    @staticmethod
    def _nhwc(t):      # NCHW tensor in channels_last memory -> contiguous [N,H,W,C] view
        return t.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)

    @staticmethod
    def _nchw(t):      # contiguous [N,H,W,C] -> NCHW view (channels_last memory)
        return t.permute(0, 3, 1, 2)

    def _forward_hip(self, x):
        P = self._plans()
        fc = self.firstconv
        x = x.contiguous(memory_format=torch.channels_last)
        c0 = fc[0][0]
        if HIP_STEM and c0.in_channels == 3 and c0.out_channels == 32 and c0.kernel_size == (3, 3) and c0.stride == (2, 2) \
                and c0.padding == (1, 1) and c0.dilation == (1, 1) and c0.bias is None:
            from . import ops                                                     # 3 -> 32 stride 2 + BN + ReLU: one VALU pass
            key = (c0.weight.data_ptr(), c0.weight._version, c0.weight.device)
            cw = c0.__dict__.get("_estd_w_nchw")
            if cw is None or cw[0] != key:
                cw = (key, c0.weight.detach().contiguous(memory_format=torch.contiguous_format).clone())
                c0.__dict__["_estd_w_nchw"] = cw
            sc, sh = _folded(fc[0][1])
            x = ops.stem3x3s2_nhwc(self._nhwc(x), cw[1], sc, sh)
        else:
            x = conv_bn_act(c0, fc[0][1], x, relu=True) if fused_on(self, x) else fc[1](fc[0](x))        # 3->32 stride 2: MIOpen
            x = self._nhwc(x)
        x = P["first2"].run(P["first1"].run(x))
        for lname in ("layer1", "layer2", "layer3", "layer4"):
            for bi, blk in enumerate(getattr(self, lname)):
                if (lname, bi, 1) in P:
                    y = P[(lname, bi, 1)].run(x)
                else:                                                             # stride-2 conv1 (layer2.0)
                    c1 = blk.conv1[0]
                    y = None
                    if _hip_taps_ok(c1[0]) and c1[0].stride == (2, 2):           # 3x3 stride 2: csrc/conv2d_taps.hip (0.066 -> 0.050 ms alone)
                        y = self._nhwc(conv_bn_act(c1[0], c1[1], self._nchw(x), relu=True))
                    if y is None:
                        y = small_conv_nhwc(c1[0], c1[1], x, relu=True)
                    if y is None:
                        y = self._nhwc(blk.first(self._nchw(x)))                  # library path
                if blk.downsample is None:
                    res = x
                else:
                    res = small_conv_nhwc(blk.downsample[0], blk.downsample[1], x, relu=False)
                    if res is None:
                        res = self._nhwc(blk.shortcut(self._nchw(x)))
                x = P[(lname, bi, 2)].run(y, residual=res)
            if lname == "layer2":
                raw = x
        skip = x
        skip_nchw = self._nchw(skip)
        size = skip_nchw.shape[2:]
        pooled = None
        if fused_on(self, skip_nchw):
            # SPP pyramid (windows 4, 8, 16, 32; psm_submodule.py:100-110) from ONE pass over the map: the coarser windows are
            # aligned unions of 4x4 cells, so their means are means of cell means (ATen's NHWC avg_pool2d walks the 49 MB map
            # once per branch and takes up to 290 us for the 32x32 windows)
            if HIP_POOL and skip.shape[3] % 4 == 0:
                from . import ops
                p4 = ops.avgpool_nhwc(skip.contiguous(), 4)
                pooled = {4: self._nchw(p4), 3: self._nchw(ops.avgpool_nhwc(p4, 2)), 2: self._nchw(ops.avgpool_nhwc(p4, 4)),
                          1: self._nchw(ops.avgpool_nhwc(p4, 8))}
            else:
                p4 = F.avg_pool2d(skip_nchw, 4, 4)
                pooled = {4: p4, 3: F.avg_pool2d(p4, 2, 2), 2: F.avg_pool2d(p4, 4, 4), 1: F.avg_pool2d(p4, 8, 8)}
        if pooled is not None and SPP_FUSED:
            from . import ops
            brs = [self._nhwc(self._branch(i, skip_nchw, pooled)) for i in (4, 3, 2, 1)]
            cat_nhwc = ops.spp_upsample_cat(raw.contiguous(), skip.contiguous(), [b.contiguous() for b in brs])     # upsample x4 + cat, one pass
        else:
            ups = [F.interpolate(self._branch(i, skip_nchw, pooled), size=size, mode="bilinear", align_corners=False) for i in (4, 3, 2, 1)]
            cat_nhwc = self._nhwc(torch.cat([self._nchw(raw), skip_nchw] + ups, 1))
        y = P["last"].run(cat_nhwc)
        last = self.lastconv[2]                                                   # 1x1, 128 -> 32, no BN
        z = small_conv_nhwc(last, None, y, relu=False)
        if z is not None:
            return self._nchw(z)
        return conv1x1_gemm(last, self._nchw(y)) if fused_on(self, y) else last(self._nchw(y))

    def _branch(self, i, skip, pooled=None):
        """SPP branch: AvgPool -> 1x1 conv -> BN -> ReLU (psm_submodule.py:100-110)."""
        br = getattr(self, "branch%d" % i)
        if getattr(self, "_hip", False) and skip.is_cuda and not self.training:      # folded running statistics: eval only
            p = br[0](skip) if pooled is None else pooled[i]
            z = small_conv_nhwc(br[1][0], br[1][1], self._nhwc(p), relu=True)
            if z is not None:
                return self._nchw(z)
        if fused_on(self, skip):
            return conv_bn_act(br[1][0], br[1][1], br[0](skip) if pooled is None else pooled[i], relu=True)
        return br(skip)

    def forward(self, x):
        if getattr(self, "_hip", False) and x.is_cuda and not self.training:
            return self._forward_hip(x)
        x = self.firstconv(x)
        x = self.layer1(x)
        raw = self.layer2(x)
        skip = self.layer4(self.layer3(raw))
        size = skip.shape[2:]
        # F.upsample(mode='bilinear') == interpolate(align_corners=False) in the oracle torch version
        ups = [F.interpolate(getattr(self, "branch%d" % i)(skip), size=size, mode="bilinear", align_corners=False)
               for i in (4, 3, 2, 1)]
        return self.lastconv(torch.cat([raw, skip] + ups, 1))


# ------------------------------------------------------------------ torchvision-layout ResNet
class _Basic(nn.Module):
    expansion = 1

    def __init__(self, cin, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        if fused_on(self, x):
            sc = x if self.downsample is None else conv_bn_act(self.downsample[0], self.downsample[1], x, relu=False)
            y = conv_bn_act(self.conv1, self.bn1, x, relu=True)
            return conv_bn_act(self.conv2, self.bn2, y, relu=True, residual=sc)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)   # stride on the 3x3 (torchvision v1.5)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        if fused_on(self, x):
            sc = x if self.downsample is None else conv_bn_act(self.downsample[0], self.downsample[1], x, relu=False)
            y = conv_bn_act(self.conv1, self.bn1, x, relu=True)
            y = conv_bn_act(self.conv2, self.bn2, y, relu=True)
            return conv_bn_act(self.conv3, self.bn3, y, relu=True, residual=sc)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return self.relu(y + (x if self.downsample is None else self.downsample(x)))


_RESNET_CFG = {18: (_Basic, (2, 2, 2, 2)), 34: (_Basic, (3, 4, 6, 3)), 50: (_Bottleneck, (3, 4, 6, 3)),
               101: (_Bottleneck, (3, 4, 23, 3)), 152: (_Bottleneck, (3, 8, 36, 3))}


class ResNetTrunk(nn.Module):
    """ResNet with torchvision's attribute names (conv1,bn1,relu,maxpool,layer1..4,avgpool,fc)."""

    def __init__(self, depth):
        super().__init__()
        if depth not in _RESNET_CFG:
            raise ValueError("{} is not a valid number of resnet layers".format(depth))
        block, reps = _RESNET_CFG[depth]
        self._cin = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._stage(block, 64, reps[0], 1)
        self.layer2 = self._stage(block, 128, reps[1], 2)
        self.layer3 = self._stage(block, 256, reps[2], 2)
        self.layer4 = self._stage(block, 512, reps[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, 1000)   # unused; kept for checkpoint key parity

    def _stage(self, block, planes, n, stride):
        down = None
        if stride != 1 or self._cin != planes * block.expansion:
            down = nn.Sequential(nn.Conv2d(self._cin, planes * block.expansion, 1, stride, bias=False),
                                 nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self._cin, planes, stride, down)]
        self._cin = planes * block.expansion
        layers += [block(self._cin, planes, 1, None) for _ in range(1, n)]
        return nn.Sequential(*layers)


class SemanticEncoder(nn.Module):
    """resnet_encoder.py:17-51: five ReLU-activated feature scales (1/2 .. 1/32)."""

    def __init__(self, num_layers, pretrained=False, num_input_images=1):
        super().__init__()
        self.num_ch_enc = np.array([64, 64, 128, 256, 512])
        self.encoder = ResNetTrunk(num_layers)     # no network here: weights come from a checkpoint
        if num_layers > 34:
            self.num_ch_enc[1:] *= 4

    def _stem_hip(self, x):
        """conv1 (3 -> 64, 7x7, stride 2, padding 3) + bn1 + relu on csrc/conv2d_taps.hip::stem7x7s2_nhwc_kernel; None when the layer
        is not that shape (e.g. num_input_images > 1) or the switch is off."""
        c0 = self.encoder.conv1
        if not HIP_STEM7 or (c0.in_channels, c0.out_channels, c0.kernel_size, c0.stride, c0.padding, c0.dilation) != \
                (3, 64, (7, 7), (2, 2), (3, 3), (1, 1)) or c0.bias is not None:
            return None
        from . import ops, packing
        key = (c0.weight.data_ptr(), c0.weight._version, c0.weight.device)
        cw = c0.__dict__.get("_estd_w7")
        if cw is None or cw[0] != key:
            cw = (key, packing.pack_stem7x7(c0.weight).to(c0.weight.device))
            c0.__dict__["_estd_w7"] = cw
        sc, sh = _folded(self.encoder.bn1)
        xn = x.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        return ops.stem7x7s2_nhwc(xn, cw[1], sc, sh).permute(0, 3, 1, 2)

    def forward(self, x):
        e = self.encoder
        if fused_on(self, x):
            f0 = self._stem_hip(x)
            if f0 is None:
                f0 = conv_bn_act(e.conv1, e.bn1, x, relu=True)
            mp = e.maxpool
            if HIP_POOL and (mp.kernel_size, mp.stride, mp.padding, mp.dilation, mp.ceil_mode) == (3, 2, 1, 1, False) and f0.shape[1] % 4 == 0:
                from . import ops
                p0 = ops.maxpool3x3s2_nhwc(f0.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
            else:
                p0 = mp(f0)
        else:
            f0 = e.relu(e.bn1(e.conv1(x)))
            p0 = e.maxpool(f0)
        f1 = e.layer1(p0)
        f2 = e.layer2(f1)
        f3 = e.layer3(f2)
        f4 = e.layer4(f3)
        return [f0, f1, f2, f3, f4]


class UpBlock(nn.Module):
    """ConvBlock of the 2D decoder: conv3x3+BN+ReLU (hybrid_depth_decoder.py:17-30)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = conv_bn2d(int(cin), int(cout), 3, 1, 1, 1)
        self.nonlin = nn.ReLU(inplace=True)

    def forward_to16(self, x_nhwc, upsample):
        """conv3x3 (16|32 -> 16) + BN + ReLU of an NHWC map [N,H,W,C] on csrc/refine2d.hip, optionally reading the nearest-x2
        upsampling of ``x_nhwc`` without materialising it.  Returns NHWC [N,uH,uW,16]."""
        from . import ops, packing
        conv, bn = self.conv[0], self.conv[1]
        key = tuple((p.data_ptr(), p._version) for p in self.parameters()) + (bn.running_mean._version, bn.running_var._version)
        c = self.__dict__.get("_plan16")
        if c is None or c[0] != key:
            sc, sh = _folded(bn)
            c = (key, packing.pack_conv2d_to16(conv.weight).to(conv.weight.device), sc, sh)
            self.__dict__["_plan16"] = c
        return ops.conv2d_k3_to16_nhwc(x_nhwc, c[1], c[2], c[3], upsample)

    def to16_ok(self):
        conv = self.conv[0]
        return conv.out_channels == 16 and conv.in_channels in (16, 32) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) \
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None

    def forward(self, x):
        if getattr(self, "_hip", False) and x.is_cuda and not self.training:
            conv = self.conv[0]
            n, _, h, w = x.shape
            items = n * ((h + 7) // 8) * ((w + 15) // 16) * (conv.out_channels // 32)
            if conv.in_channels % 32 == 0 and conv.out_channels % 32 == 0 and items >= HIP_3X3_MIN_ITEMS:     # enough tiles to fill 256 CUs
                from . import ops
                key = tuple((p.data_ptr(), p._version) for p in self.parameters())
                if getattr(self, "_plan", None) is None or self._plan[0] != key:
                    self._plan = (key, ops.Conv2dPlan(conv, self.conv[1], relu_before=True))
                y = self._plan[1].run(x.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1))
                return y.permute(0, 3, 1, 2)
        if fused_on(self, x):
            return conv_bn_act(self.conv[0], self.conv[1], x, relu=True)
        return self.nonlin(self.conv(x))


def up2(x):
    return F.interpolate(x, scale_factor=2, mode="nearest")
