"""DepthNetHybrid: drop-in for hybrid_models/model_hybrid.py:14-184 (constructor, state-dict keys,
``forward`` signature and return values), with the plane-sweep / cost-volume build, 3D regularisation,
EST fusion and soft-argmin executed by hand-written gfx950 kernels.

Inference only ('val' / 'test' style calls); the loss/metric bookkeeping of the reference's 'train'
mode is outside this path.  abs_rel (model_hybrid.py:306) is provided for the benchmark report.
"""
import os

import torch
import torch.nn as nn

from . import camera, ops
from .backbones import PSMFeatures, SemanticEncoder
from .hybrid_depth_decoder import DepthHybridDecoder
from .homo_utils import homo_warping  # noqa: F401  (re-exported like the reference module does)
from .layers_op import PlanCache, convbn_3d, convbnrelu_3d

Align_Corners_Range = False


# A/B switches, read once at import
# ESTD_FAST_PATH=0: a model moved to a ROCm device keeps the plain module path (NCHW MIOpen 2D networks, one stream) instead of
# switching the accelerators on by itself; the individual switches below / the use_*() methods still apply
FAST_PATH = os.environ.get("ESTD_FAST_PATH", "1") == "1"
FUSED_NORM = os.environ.get("ESTD_FUSED_NORM", "1") == "1"     # image normalisation + NHWC layout in one kernel
MIX_GEMM = os.environ.get("ESTD_MIX_GEMM", "1") == "1"         # pre0 halves as two library GEMMs on NHWC features
MIX_HIP = os.environ.get("ESTD_MIX_HIP", "1") == "1"           # ... as two 1x1 convolutions on csrc/conv1x1.hip instead (A/B switch)
R50_HIP = os.environ.get("ESTD_R50_HIP", "1") == "1"           # ResNet stride-1 3x3 convolutions on the MFMA conv2d kernel
HIP_REFINE = os.environ.get("ESTD_HIP_REFINE", "1") == "1"     # decoder 2D tail glue kernels (csrc/refine2d.hip)


class DepthNetHybrid(nn.Module):
    def __init__(self, ndepths=64, depth_min=0.01, depth_max=10.0, resnet=50, IF_EST_transformer=True):
        super().__init__()
        self.ndepths = ndepths
        self.depth_min = depth_min
        self.depth_max = depth_max
        self.depth_interval = (depth_max - depth_min) / (ndepths - 1)
        # same fp32 arithmetic as model_hybrid.py:32-33 (plain attribute, not a buffer)
        self.depth_cands = torch.arange(0, ndepths, requires_grad=False).reshape(1, -1).to(
            torch.float32) * self.depth_interval + self.depth_min
        self.IF_EST_transformer = IF_EST_transformer
        self.matchingFeature = PSMFeatures()
        self.semanticFeature = SemanticEncoder(resnet, "pretrained")
        self.stage_infos = {"stage1": {"scale": 4.0}, "stage2": {"scale": 2.0}, "stage3": {"scale": 1.0}}
        self.CostRegNet = DepthHybridDecoder(self.semanticFeature.num_ch_enc, num_output_channels=1, use_skips=True,
                                             ndepths=self.ndepths, depth_max=self.depth_max,
                                             IF_EST_transformer=self.IF_EST_transformer)
        self.pre0 = convbn_3d(64, 32, 1, 1, 0)
        self.pre1 = convbnrelu_3d(32, 32, 3, 1, 1)
        self.pre2 = convbn_3d(32, 32, 3, 1, 1)
        self._cache = PlanCache()
        # "host" (default): camera matrices with the reference's own torch-CPU calls, bit-identical (estdepth_amd/camera.py);
        # "device": estd_cam_* kernels, no host synchronisation, boundary samples may flip vs the reference
        self._camera_algebra = "host"
        self.CostRegNet.camera_algebra = "host"
        # weights epoch: bumped whenever parameters are replaced wholesale (load_state_dict, .to()/.cuda()); captured
        # hipGraphs (estdepth_amd.graph) bake packed-weight addresses in and re-capture when it changes
        self._estd_weights_epoch = 0
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._bump_weights_epoch())

    def _bump_weights_epoch(self):
        self._estd_weights_epoch += 1

    @property
    def camera_algebra(self):
        return self._camera_algebra

    @camera_algebra.setter
    def camera_algebra(self, mode):
        if mode not in ("host", "device"):
            raise RuntimeError("camera_algebra must be 'host' or 'device', got %r" % (mode,))
        self._camera_algebra = mode
        self.CostRegNet.camera_algebra = mode            # the decoder's stand-alone fallback follows the model

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._estd_weights_epoch = getattr(self, "_estd_weights_epoch", 0) + 1
        # A model that lands on a ROCm device runs the benchmarked configuration by default: NHWC 2D networks, PSM / ResNet
        # stride-1 3x3 convolutions on the MFMA conv2d kernels, fused BN epilogues, semantic branch and stereo heads on a side
        # stream.  Every one of them passes the same golden / oracle parity tests as the plain path (tests/test_gpu_*).
        if FAST_PATH and not getattr(self, "_estd_accel_decided", False) and self.pre0[0].weight.is_cuda:
            self._estd_accel_decided = True
            self.accelerate()
        return out

    def accelerate(self, enable=True):
        """switch every accelerator of the 2D networks / stream layout on (the default on a ROCm device) or off (``plain_path()``)."""
        self._estd_accel_decided = True
        self.use_hip_psm(enable)
        self.fuse_bn_2d(enable)
        self.overlap_semantic_branch(enable)
        self.use_channels_last_2d(enable)
        return self

    def plain_path(self):
        """the plain module path: NCHW library convolutions for the 2D networks, separate BatchNorm passes, one stream (what
        ``ESTD_FAST_PATH=0`` keeps).  The 3D hot path is the HIP library either way."""
        return self.accelerate(False)

    # ------------------------------------------------------------------------------ packed weights
    def _plans(self):
        def build():
            dev = self.pre0[0].weight.device
            w0 = self.pre0[0].weight.detach().reshape(32, 64).float().cpu()
            sc, sh = self.pre0.folded()
            # pre0(cat[ref, warped]) = (sc*W[:, :32]) ref + sh  +  warp((sc*W[:, 32:]) src)      (:93-94)
            return {"w_ref": (sc[:, None] * w0[:, :32]).contiguous().to(dev), "b_ref": sh.contiguous().to(dev),
                    "w_src": (sc[:, None] * w0[:, 32:]).contiguous().to(dev),
                    "pre1": self.pre1.plan(), "pre2": self.pre2.plan(), "pre2x2": self.pre2.plan().with_shift_scaled(2)}
        return self._cache.get((self.pre0, self.pre1, self.pre2), build)      # only these three layers feed the plans

    def camera_matrices(self, cam_poses, cam_intr_stage1, pre_cam_poses=None):
        """Every camera matrix one forward needs (estdepth_amd/camera.py): {"sweep": [T,2,12], "vol": [T,n,30] or None}
        on the model's device.  ``vol`` (frustum-to-frustum maps of the EST fusion) is only formed when the transformer
        branch will run (hybrid_depth_decoder.py:423)."""
        T = cam_poses.shape[1] - 2
        dev = self.pre0[0].weight.device
        est = self.IF_EST_transformer and pre_cam_poses is not None
        plist = [cam_poses[:, t + 1] for t in range(T)] + (list(pre_cam_poses) if est else [])
        if self.camera_algebra == "device":
            return {"sweep": camera.sweep_projections_device(cam_poses.to(dev), cam_intr_stage1.to(dev)),
                    "vol": camera.volume_matrices_device([p.to(dev) for p in plist], T, cam_intr_stage1.to(dev)) if est else None}
        return camera.forward_matrices(cam_poses, cam_intr_stage1, pre_cam_poses, est, dev)

    def _costvolumes(self, ref_mixes, src_mix_pairs, sweep, depth_values, P=None):
        """Fused get_costvolume for T targets at once on pre-mixed 2D features (model_hybrid.py:76-99).
        ref_mixes: T tensors [H,W,32]; src_mix_pairs: T lists of source mixes; sweep [T,n_src,12] homographies;
        returns [T,D,H,W,32].
        The k-th sources of all targets share one batched convolution launch (N = T), and the running mean
        over sources is the second launch's accumulate epilogue -- no race, same arithmetic as :97-99."""
        P = self._plans() if P is None else P
        T = len(ref_mixes)
        H, W, _ = ref_mixes[0].shape
        D = self.ndepths
        dims = (T, D, H, W)
        dev = ref_mixes[0].device
        cost = torch.empty((T, D, H, W, 32), device=dev, dtype=torch.float32)
        y = torch.empty_like(cost)
        n_src = len(src_mix_pairs[0])

        def run_sweep(k, x):
            for t in range(T):      # proj = :74-88 + homo_utils.py:469-471 (camera.py)
                ops.homo_warp_costvol(src_mix_pairs[t][k], ref_mixes[t], sweep[t, k], depth_values, D, out=x[t])   # :90-94

        if n_src == 2:
            # pre2 = conv + BN is LINEAR (no activation, :60), so  sum_k pre2(y_k) = conv(sum_k y_k)*s + n*t :
            # ONE pre2 convolution per target instead of one per source.  cost = (x_0 + x_1 + pre2sum(y_0 + y_1)) / 2.
            xs = [torch.empty_like(cost), torch.empty_like(cost)]
            for k in range(2):
                run_sweep(k, xs[k])
                P["pre1"].run(xs[k], dims, out=y, out_stride=32, accumulate=(k > 0))              # y = y_0 + y_1   (:95)
            P["pre2x2"].run(y, dims, out=cost, out_stride=32, residual=xs[0], residual2=xs[1], out_scale=0.5)   # :95-99
        else:
            x = torch.empty_like(cost)
            for k in range(n_src):
                run_sweep(k, x)
                P["pre1"].run(x, dims, out=y, out_stride=32)                                      # :95
                P["pre2"].run(y, dims, out=cost, out_stride=32, residual=x,
                              out_scale=1.0 / n_src, accumulate=(k > 0))                          # :95-99
        return cost


    def _mix(self, feature_chw, which, P=None):
        P = self._plans() if P is None else P
        if which == "ref":
            return ops.mix1x1(feature_chw, P["w_ref"], P["b_ref"])
        return ops.mix1x1(feature_chw, P["w_src"], None)

    def get_costvolume(self, features, cam_poses, cam_intr, depth_values):
        """model_hybrid.py:62-102.  features: sequence of V tensors [1,32,H,W] (middle = reference view);
        cam_poses [1,V,4,4]; cam_intr [1,3,3] at 1/4 scale; depth_values [1,D,1,1].
        Returns [1,32,D,H,W] (a channels-last view: same values/shape as the reference's tensor)."""
        num_views = len(features)
        if features[0].shape[0] != 1:
            raise RuntimeError("estdepth_amd runs one sequence per call (batch 1)")
        mid = num_views // 2
        dev = features[0].device
        dv = depth_values.reshape(-1)[:self.ndepths].contiguous().float()
        ref_mix = self._mix(features[mid][0].contiguous(), "ref")
        srcs = [v for v in range(num_views) if v != mid]
        src_mixes = [self._mix(features[v][0].contiguous(), "src") for v in srcs]
        if self.camera_algebra == "device":
            poses, K = cam_poses[0].contiguous().float(), cam_intr[0].contiguous().float()
            sweep = torch.stack([ops.cam_sweep_proj(poses[mid], poses[v], K) for v in srcs])[None]
        else:       # the reference's own torch-CPU composition (camera.py)
            sweep = camera.sweep_projection_set(cam_poses, cam_intr, mid, srcs, dev)[None]
        cost = self._costvolumes([ref_mix], [src_mixes], sweep, dv)[0]
        return cost.permute(3, 0, 1, 2).unsqueeze(0)

    def scale_cam_intr(self, cam_intr, scale):
        """intrinsics of a down-scaled image: focal lengths and principal point (rows 0,1) times ``scale`` (model_hybrid.py:104-108)."""
        k = cam_intr.clone()
        k[:, 0:2] = k[:, 0:2] * scale
        return k

    def use_channels_last_2d(self, enable=True):
        """Opt-in: run the 2D backbones (PSM, ResNet, 2D decoder -- MIOpen, outside the hot path) in NHWC.
        ~1.4 ms/step faster at cfg2; feature differences vs NCHW are ~1e-5 (different MIOpen solvers)."""
        self._channels_last_2d = bool(enable)
        fmt = torch.channels_last if enable else torch.contiguous_format
        for mod in (self.matchingFeature, self.semanticFeature):
            mod.to(memory_format=fmt)
        for name, child in self.CostRegNet.named_children():
            if name.startswith("upconv") or name.startswith("dispconv"):
                child.to(memory_format=fmt)
        return self

    def fuse_bn_2d(self, enable=True):
        """Opt-in: BatchNorm2d -> (residual add) -> ReLU after the library convolutions of the 2D backbones as ONE
        in-place NHWC pass (estd_bn_act_nhwc) instead of two or three MIOpen/ATen launches.  Implies NHWC backbones."""
        from .backbones import enable_fused_bn
        self.use_channels_last_2d(True)
        for mod in (self.matchingFeature, self.semanticFeature, self.CostRegNet):
            enable_fused_bn(mod, enable)
        return self

    def normalise_images(self, imgs):
        """2*(imgs/255)-1 (model_hybrid.py:119), exposed for callers that cache per-frame matching features."""
        return 2 * (imgs / 255.) - 1.

    def overlap_semantic_branch(self, enable=True):
        """Opt-in: run the semantic branch on a second HIP stream (captured as a parallel graph branch)."""
        self._overlap_semantic = bool(enable)
        self.CostRegNet._overlap_heads = bool(enable)      # stereo-head convs + soft-argmin on a side stream as well
        return self

    def use_hip_psm(self, enable=True):
        """Opt-in: the 3x3 convolutions of the PSM matching-feature extractor on the MFMA conv2d kernel
        (SURVEY §8f rank 2).  Implies NHWC 2D backbones."""
        from .backbones import enable_hip_3x3
        self.use_channels_last_2d(True)
        self.matchingFeature.use_hip_convs(enable)
        if R50_HIP:                                        # A/B switch
            enable_hip_3x3(self.semanticFeature, enable)   # ResNet stride-1 3x3 convs (with fuse_bn_2d())
        for name, child in self.CostRegNet.named_children():      # 2D decoder ConvBlocks with enough tiles (120x160 and up)
            if name.startswith("upconv"):
                child._hip = bool(enable)
        self.CostRegNet._hip_refine = bool(enable) and HIP_REFINE                                    # A/B switch (csrc/refine2d.hip)
        return self

    # The forward pass in two stages, so that the host can evaluate the camera matrices while the GPU is busy:
    #   forward_2d  -- everything that does not depend on the cameras: PSM matching features of every frame and the semantic
    #                  branch (ResNet + 2D decoder scales 4..2) of the target frames, ~25 % of a step;
    #   forward_3d  -- plane sweeps, cost volumes, 3D regularisation, EST fusion, soft-argmin, 2D refinement.
    # forward() starts the (asynchronous) device-to-host copy of the poses, launches stage 1, and only then waits for the
    # copy and composes the matrices (estdepth_amd/camera.py): the synchronisation the exact host algebra needs costs no GPU time.
    def _matching(self, flat, matching_features):
        """PSM matching features of the V frames of a call (model_hybrid.py:128).  ``matching_features`` [Vc,32,H/4,W/4], Vc <= V: features
        of the LEADING Vc frames a caller already holds (overlapping windows / clips: estdepth_amd.streaming) -- only the trailing V - Vc
        frames go through the extractor, on the stream this is called on (beside the semantic branch), and the result is the full stack."""
        if matching_features is None:
            return self.matchingFeature(flat)
        vc = matching_features.shape[0]
        if vc == flat.shape[0]:
            return matching_features
        if vc > flat.shape[0]:
            raise RuntimeError("matching_features holds %d frames, the call has %d" % (vc, flat.shape[0]))
        new = self.matchingFeature(flat[vc:])
        if matching_features.is_contiguous(memory_format=torch.channels_last) != new.is_contiguous(memory_format=torch.channels_last):
            matching_features = matching_features.contiguous(memory_format=torch.channels_last if new.is_contiguous(memory_format=torch.channels_last)
                                                              else torch.contiguous_format)
        return torch.cat([matching_features, new], 0)

    @torch.no_grad()
    def forward_2d(self, imgs, matching_features=None, join=False):
        """imgs [1,V,3,Hi,Wi] in 0..255 -> the camera-independent features.  ``join``: wait for the side stream before returning
        (a captured hipGraph has to end with its streams joined).  ``matching_features``: see ``_matching``."""
        batch_size, views_num, _, height_img, width_img = imgs.shape
        assert views_num > 2  # the views_num should be larger than 2 (model_hybrid.py:123)
        if batch_size != 1:
            raise RuntimeError("estdepth_amd runs one sequence per call (the reference's view() also fails for batch > 1)")
        fused_norm = getattr(self, "_channels_last_2d", False) and imgs.is_cuda and imgs.dtype == torch.float32 and imgs.shape[2] == 3 \
            and FUSED_NORM
        if fused_norm:                                   # :119 and the NHWC layout of the 2D networks in one pass (csrc/refine2d.hip)
            imgs = ops.normalise_nhwc(imgs.reshape(views_num, 3, height_img, width_img).contiguous()).permute(0, 3, 1, 2)[None]
        else:
            imgs = 2 * (imgs / 255.) - 1.                                                                     # :119
        target_num = views_num - 2
        flat = imgs.reshape(batch_size * views_num, 3, height_img, width_img)
        if getattr(self, "_channels_last_2d", False):
            flat = flat.contiguous(memory_format=torch.channels_last)      # MIOpen NHWC kernels for the 2D backbones
        sv_pre = None
        if getattr(self, "_overlap_semantic", False):
            # fork: ResNet + 2D decoder scales 4..2 (many small kernels) on a side stream, concurrently with the
            # PSM -> plane sweep -> pre1/pre2 chain; joined in the decoder right before dres2 needs the plane scores
            main = torch.cuda.current_stream()
            if getattr(self, "_side_stream", None) is None:
                self._side_stream = torch.cuda.Stream()          # (stream priorities changed nothing: stage A is the sum of its kernels' work, DESIGN)
            side = self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side):
                semantic_features = self.semanticFeature(flat[1:1 + target_num])                           # :138-139
                sv = self.CostRegNet._semantic_vs(semantic_features).contiguous()     # [T,D,H,W] planes (the decoder's layout)
            for t_ in list(semantic_features) + [sv]:
                t_.record_stream(main)
            matching = self._matching(flat, matching_features)                                             # :128
            if join:
                main.wait_stream(side)
            sv_pre = (None if join else side, sv)
        else:
            matching = self._matching(flat, matching_features)                                             # :128
            semantic_features = self.semanticFeature(flat[1:1 + target_num])                               # :138-139 (batch 1)
        self.last_matching = matching      # [V,32,H/4,W/4]: what a streaming caller slices its next call's ``matching_features`` from
        feats = {"matching": matching, "semantic_features": semantic_features, "sv_pre": sv_pre, "views_num": views_num,
                 "device": imgs.device, "dtype": imgs.dtype}
        self.last_features2d = feats       # (references only: the feature-level parity bar of bench.py / tests reads them after the call)
        return feats

    @torch.no_grad()
    def forward_3d(self, feats, cam_poses, cam_intr, sample, pre_costs=None, pre_cam_poses=None, mode="val", cam_mats=None):
        matching, semantic_features, views_num = feats["matching"], feats["semantic_features"], feats["views_num"]
        target_num = views_num - 2
        cam_intr_stage1 = self.scale_cam_intr(cam_intr, scale=1. / self.stage_infos["stage1"]["scale"])     # :142
        dkey = (feats["device"], feats["dtype"])
        if getattr(self, "_dv_cache", None) is None or self._dv_cache[0] != dkey:      # one H2D copy, not one per call
            self._dv_cache = (dkey, self.depth_cands.view(1, self.ndepths, 1, 1).to(dkey[1]).to(dkey[0]))
        depth_values = self._dv_cache[1]                                                                     # :144-145
        dv = depth_values.reshape(-1).contiguous()
        if cam_mats is None:                                 # :74-88, homo_utils.py:469, decoder :235 (estdepth_amd/camera.py)
            cam_mats = self.camera_matrices(cam_poses, cam_intr_stage1, pre_cam_poses)
        self.CostRegNet._vol_mats_pre = cam_mats.get("vol")
        self.CostRegNet._semantic_vs_pre = feats["sv_pre"]

        # every view is a source for up to two targets: mix each 2D feature once (pre0 pushed in front of the warp)
        P = self._plans()                                   # one cache-key check per forward
        if matching.is_cuda and matching.dim() == 4 and matching.shape[1] == 32 and \
                matching.is_contiguous(memory_format=torch.channels_last) and MIX_GEMM:
            # NHWC matching features (HIP PSM path): pre0's two halves are plain [pixels, 32] x [32, 32] library GEMMs on the
            # records as they lie -- two launches for all views instead of a CHW copy + one mix kernel per view and role
            V, _, Hf, Wf = matching.shape
            rec = matching.permute(0, 2, 3, 1).reshape(V * Hf * Wf, 32)
            if MIX_HIP:      # pre0's two halves as 1x1 convolutions of the NHWC records on csrc/conv1x1.hip (no library GEMM in the hot path)
                nhwc = matching.permute(0, 2, 3, 1)
                src_all = ops.conv1x1_nhwc(nhwc, P["w_src"], None, None)
                ref_all = ops.conv1x1_nhwc(nhwc[1:target_num + 1], P["w_ref"], None, P["b_ref"])
            else:
                src_all = torch.mm(rec, P["w_src"].t()).view(V, Hf, Wf, 32)
                ref_all = torch.addmm(P["b_ref"], rec[Hf * Wf:(target_num + 1) * Hf * Wf], P["w_ref"].t()).view(target_num, Hf, Wf, 32)
            src_mix = [src_all[v] for v in range(views_num)]
            ref_mix = [ref_all[t] for t in range(target_num)]
        else:
            src_mix = [self._mix(matching[v].contiguous(), "src", P) for v in range(views_num)]
            ref_mix = [self._mix(matching[t + 1].contiguous(), "ref", P) for t in range(target_num)]
        costs = self._costvolumes(ref_mix, [[src_mix[t], src_mix[t + 2]] for t in range(target_num)],
                                  cam_mats["sweep"], dv, P)                                             # :152-156
        cost_volumes = [costs[t].permute(3, 0, 1, 2).unsqueeze(0) for t in range(target_num)]
        target_cam_poses = [cam_poses[:, t + 1, :, :] for t in range(target_num)]                            # :161

        outputs, cur_costs, cur_cam_poses = self.CostRegNet(cost_volumes, semantic_features, target_cam_poses,
                                                            cam_intr_stage1, depth_values, self.depth_min,
                                                            self.depth_interval, pre_costs, pre_cam_poses, mode)   # :166
        if mode == 'test':                                                                                   # :179-181
            gts = [sample["dmaps"][:, t + 1] for t in range(target_num)]
            masks = [sample["dmasks"][:, t + 1] for t in range(target_num)]
            return outputs, self.depth_metrics(outputs, [0, 2], gts, masks, target_num)
        return outputs, cur_costs, cur_cam_poses

    def camera_begin(self, cam_poses, cam_intr, pre_cam_poses=None):
        """start the asynchronous device-to-host copy of everything the host camera algebra reads (None in "device" mode)."""
        if self.camera_algebra != "host":
            return None
        k4 = self.scale_cam_intr(cam_intr, scale=1. / self.stage_infos["stage1"]["scale"])
        est = self.IF_EST_transformer and pre_cam_poses is not None
        return camera.begin(cam_poses, k4, pre_cam_poses, est, self.pre0[0].weight.device)

    @torch.no_grad()        # inference-only implementation: the HIP operators do not record autograd graphs
    def forward(self, imgs, cam_poses, cam_intr, sample, pre_costs=None, pre_cam_poses=None, mode='train',
                matching_features=None, cam_mats=None):
        """model_hybrid.py:110-184.  imgs [1,V,3,Hi,Wi] in 0..255; cam_poses [1,V,4,4] camera-to-world; cam_intr [1,3,3]
        full-resolution pixels; returns (outputs, cur_costs, cur_cam_poses) for inference modes.  Two optional extensions:
        ``matching_features`` (estdepth_amd.streaming) = PSM features [Vc,32,H/4,W/4] of the leading Vc <= V frames, so overlapping
        windows / clips do not recompute them (the trailing frames are extracted in this call; ``last_matching`` holds the full stack
        afterwards); ``cam_mats`` = the result of ``camera_matrices`` for these poses."""
        if mode == 'train' or self.training:
            raise RuntimeError("estdepth_amd implements the inference path (mode='val'/'test'); training is out of scope")
        pending = self.camera_begin(cam_poses, cam_intr, pre_cam_poses) if cam_mats is None else None
        feats = self.forward_2d(imgs, matching_features)
        if pending is not None:
            cam_mats = camera.finish(pending)
        return self.forward_3d(feats, cam_poses, cam_intr, sample, pre_costs, pre_cam_poses, mode, cam_mats)

    METRIC_NAMES = ("a1", "a2", "a3", "abs_diff", "abs_rel", "sq_rel", "rmse", "rmse_log")

    def metrics(self, gt, pred):
        """the eight depth-error figures of model_hybrid.py:305-320 on flat (masked) tensors, in METRIC_NAMES order."""
        ratio = torch.max(gt / pred, pred / gt)
        within = [(ratio < 1.25 ** k).float().mean() for k in (1, 2, 3)]
        err = gt - pred
        log_err = torch.log(gt) - torch.log(pred)
        return (*within, err.abs().mean(), (err.abs() / gt).mean(), (err ** 2 / gt).mean(),
                (err ** 2).mean().sqrt(), (log_err ** 2).mean().sqrt())

    def depth_metrics(self, outputs, scales, depth_gt_ms, gt_masks, target_num):
        """model_hybrid.py:264-303: per scale, the mean over targets of every figure, keys '<name>_<scale>';
        accumulated as 0 + sum_t value_t / target_num in target order like the reference."""
        out = {}
        for scale in scales:
            acc = [torch.zeros((), dtype=torch.float32, device=depth_gt_ms[0].device) for _ in self.METRIC_NAMES]
            for t in range(target_num):
                pred, gt, mask = outputs[("depth", t, scale)], depth_gt_ms[t], gt_masks[t]
                for a, v in zip(acc, self.metrics(gt[mask], pred[mask])):
                    a += v / target_num
            for name, a in zip(self.METRIC_NAMES, acc):
                out["{}_{}".format(name, scale)] = a
        return out


def abs_rel(pred, gt):
    """mean(|gt - pred| / gt)  (model_hybrid.py:306, metric.py:131-150)."""
    return torch.mean(torch.abs(gt - pred) / gt)
