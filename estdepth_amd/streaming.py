"""Streaming (ESTM) harness: the per-frame protocol of the reference's eval_hybrid_seq.py as a reusable class.

    eval_hybrid_seq.py:160-193  frames arrive one by one; when ``lwindow`` (3) frames are buffered the model runs on
                                that window (middle frame = target) with the memory of up to ``memory_size`` (2)
                                earlier windows; the returned (costs, pose) joins the memory; the oldest frame leaves.
    eval_hybrid_seq.py:76-120   lw2batch: stack the window on dim 1, build pre_costs / pre_cam_poses from the memory.

SURVEY §8(f) rank 1: consecutive windows share lwindow-1 frames, and the PSM matching features of a frame do not
depend on the window, so they are computed ONCE per frame and handed to ``DepthNetHybrid.forward`` through its
``matching_features`` extension (406 of ~1320 GFLOP per depth frame at cfg3).  Everything else is the unchanged forward.
"""
import torch


class ESTMStream:
    def __init__(self, model, lwindow=3, memory_size=2, cache_features=True, graph=False):
        """``graph=True``: replay captured hipGraphs instead of ~330 eager launches per window -- one graph per number of
        memory volumes (0, 1, ..memory_size) for the window forward and one for the per-frame PSM extraction
        (estdepth_amd.graph); same kernels, same results, returned ``outputs`` live until the next push.  The harness owns the
        memory protocol, so the replay runs with ``zero_copy_memory=True``: the (costs, poses) a push returns lie in a ring of
        ``memory_size + 1`` buffers and are read back in place -- they stay valid for ``memory_size`` further pushes (as long as the
        harness itself uses them); clone them to keep them longer."""
        if lwindow < 3:
            raise RuntimeError("a window needs at least 3 frames (model_hybrid.py:123)")
        self._psm = None                     # None = model.matchingFeature, looked up at call time
        if graph:
            from .graph import GraphedForward, GraphedModule
            if not isinstance(model, GraphedForward):
                model = GraphedForward(model, zero_copy_memory=True)
            self._psm = GraphedModule(model.matchingFeature, owner=model.model)     # keyed on the model's weights epoch
        self.model = model
        self.lwindow = lwindow
        self.memory_size = memory_size
        self.cache_features = cache_features
        self.reset()

    def reset(self):
        """start of a new sequence (eval_hybrid_seq.py:162-166)."""
        self._frames = []          # dicts: img [1,3,H,W], pose [1,4,4], dmap, dmask, feat
        self._mem_costs = []
        self._mem_poses = []
        self.windows = 0

    @torch.no_grad()
    def push(self, img, cam_pose, cam_intr, dmap=None, dmask=None):
        """img [3,Hi,Wi] or [1,3,Hi,Wi] in 0..255; cam_pose [4,4]; cam_intr [3,3] (full resolution).
        Returns None while the window is filling, else (outputs, costs, poses) of the window's target frame."""
        img = img.reshape(1, *img.shape[-3:])
        pose = cam_pose.reshape(1, 4, 4)
        hi, wi = img.shape[-2:]
        fr = {"img": img, "pose": pose,
              "dmap": dmap if dmap is not None else torch.ones(1, 1, hi, wi, device=img.device),
              "dmask": dmask if dmask is not None else torch.ones(1, 1, hi, wi, device=img.device, dtype=torch.bool),
              "feat": None}
        if self.cache_features:
            x = self.model.normalise_images(img)
            if getattr(self.model, "_channels_last_2d", False):
                x = x.contiguous(memory_format=torch.channels_last)
            fr["feat"] = (self._psm or self.model.matchingFeature)(x)                                       # [1,32,H/4,W/4], once per frame
        self._frames.append(fr)
        if len(self._frames) < self.lwindow:
            return None
        win = self._frames[-self.lwindow:]
        imgs = torch.stack([f["img"][0] for f in win], 0)[None]                # lw2batch: stack on dim 1
        poses = torch.stack([f["pose"][0] for f in win], 0)[None]
        sample = {"dmaps": torch.stack([f["dmap"][0] for f in win], 0)[None],
                  "dmasks": torch.stack([f["dmask"][0] for f in win], 0)[None]}
        if self._mem_poses:
            pre_costs = {"keys": [c["keys"][0] for c in self._mem_costs], "values": [c["values"][0] for c in self._mem_costs]}
            pre_poses = [p[0] for p in self._mem_poses]
        else:
            pre_costs, pre_poses = None, None
        feats = torch.cat([f["feat"] for f in win], 0) if self.cache_features else None
        outputs, costs, cposes = self.model(imgs, poses, cam_intr.reshape(1, 3, 3), sample, pre_costs, pre_poses,
                                            mode="val", matching_features=feats)
        self._mem_costs.append(costs)
        self._mem_poses.append(cposes)
        self._frames.pop(0)                                                     # eval_hybrid_seq.py:190
        if len(self._mem_costs) > self.memory_size:                             # :191-193
            self._mem_costs.pop(0)
            self._mem_poses.pop(0)
        self.windows += 1
        return outputs, costs, cposes
