"""HIP-graph replay of ``DepthNetHybrid.forward`` (opt-in accelerator; same kernels, same results).

A forward pass is ~330 kernel launches (MIOpen 2D backbones + our HIP kernels) with static shapes; launched
eagerly the GPU idles ~12 % of the step between launches (profiles/).  ``GraphedForward`` captures one forward
per call signature into TWO hipGraphs on first use -- stage A: the camera-independent 2D networks, stage B: everything from
the plane sweep on -- and replays them afterwards; between the two launches the host composes the camera matrices
(estdepth_amd/camera.py) while the GPU is busy with stage A, so the exact host-side algebra costs no GPU time.  It is a drop-in for the model in
every inference call site of the reference (``eval_hybrid.py:229-243``, ``eval_hybrid_seq.py:160-193``) and in
``estdepth_amd.streaming.ESTMStream`` (attribute access falls through to the wrapped model).

Contract differences from the eager call (documented, not hidden):
  * the tensors of the returned ``outputs`` dict are the graph's static output buffers: they are overwritten by
    the next call with the same signature -- consume or clone them first (the reference's eval loops copy results
    to the host right after each call, eval_hybrid_seq.py:210-236), or construct with ``clone_outputs=True``;
  * the returned memory ``(costs, cam_poses)`` ARE fresh tensors (one 157 MB device copy at cfg2 size): callers
    keep them across calls (memory_size = 2 windows in the ESTM protocol), so they must not alias each other;
  * memory volumes handed in as ``pre_costs`` are copied into static input buffers (one device copy per volume);
  * ``mode='test'`` (metrics on boolean-masked ground truth) synchronises with the host and cannot be captured:
    it raises.  Run ``mode='val'`` through the graph and evaluate the metrics on the outputs.
A capture is keyed by (input shape, number of memory volumes, matching-features given?, mode, convolution arithmetic,
weights epoch of the model): ``load_state_dict`` / ``.to()`` bump the epoch and force a re-capture; call
``invalidate()`` after editing parameters in place.
"""
import torch

from . import camera
from .hybrid_depth_decoder import kv_from_pair, kv_views
from .layers_op import PlanCache


class GraphedForward:
    def __init__(self, model, warmup=2, clone_outputs=False):
        """``clone_outputs=True``: the returned ``outputs`` dict holds fresh tensors (18 device copies of [1,1,Hi,Wi] maps per
        Joint call) instead of the graph's static output buffers -- a true drop-in for callers that keep outputs across calls."""
        self.model = model
        self.warmup = warmup
        self.clone_outputs = clone_outputs
        self.memory_logits = None                # after a call: fresh copy of DepthHybridDecoder.memory_logits of that call
        self._graphs = {}

    def __getattr__(self, name):                 # normalise_images, matchingFeature, ndepths, ... of the wrapped model
        if name in ("model", "warmup", "_graphs", "clone_outputs", "memory_logits"):
            raise AttributeError(name)
        return getattr(self.model, name)

    def invalidate(self):
        """drop every captured graph (after in-place edits of parameters, which no epoch counter can see)."""
        self._graphs.clear()

    def _signature(self, imgs, pre_costs, mode, matching_features):
        n_mem = 0 if pre_costs is None else len(pre_costs["keys"])
        from . import ops
        return (tuple(imgs.shape), n_mem, matching_features is not None, mode, ops.CONV3D_ARITH, ops.CONV2D_ARITH,
                ops.CONV3D_ALGO, getattr(ops, "CONV2D_ALGO", None), getattr(ops, "CONV2D_NT", None),      # a graph bakes the kernel choice in
                self.model.camera_algebra,
                getattr(self.model, "_estd_weights_epoch", 0))     # (last) a captured graph bakes kernel choice and weight buffers in

    def _capture(self, key, imgs, cam_poses, cam_intr, sample, pre_costs, pre_cam_poses, mode, matching_features):
        m = self.model
        st = {"imgs": imgs.clone(), "poses": cam_poses.clone(), "intr": cam_intr.clone(),
              "sample": {k: v.clone() for k, v in sample.items()},
              "feats": matching_features.clone() if matching_features is not None else None}
        if pre_costs is not None:
            st["kv"] = [kv_from_pair(k, v).clone() for k, v in zip(pre_costs["keys"], pre_costs["values"])]
            st["mem_poses"] = [p.clone() for p in pre_cam_poses]
        # Host camera algebra (estdepth_amd/camera.py): the matrices enter stage B as static inputs.  In "device" mode they
        # are formed by kernels inside stage B from the static pose buffers.
        pending = m.camera_begin(cam_poses, cam_intr, pre_cam_poses)
        st["cam"] = camera.finish(pending) if pending is not None else None

        def run_a():                                     # stage A: camera-independent 2D networks, streams joined at the end
            return m.forward_2d(st["imgs"], st["feats"], join=True)

        def run_b(feats):                                # stage B: everything downstream of the plane sweep
            pc, pp = None, None
            if pre_costs is not None:
                pairs = [kv_views(kv) for kv in st["kv"]]
                pc = {"keys": [k for k, _ in pairs], "values": [v for _, v in pairs]}
                pp = list(st["mem_poses"])
            return m.forward_3d(feats, st["poses"], st["intr"], st["sample"], pc, pp, mode, cam_mats=st["cam"])

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):      # MIOpen algorithm search, hipFuncSetAttribute, plan packing: all before capture
                run_b(run_a())
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # thread_local: other threads (e.g. the RCCL watchdog polling events) may keep issuing HIP calls during capture
        with torch.no_grad(), torch.cuda.graph(ga, capture_error_mode="thread_local"):
            feats = run_a()
        with torch.no_grad(), torch.cuda.graph(gb, pool=ga.pool(), capture_error_mode="thread_local"):
            out = run_b(feats)
        st["graph_a"], st["graph_b"], st["feats2d"], st["out"] = ga, gb, feats, out
        st["memory_logits"] = getattr(m.CostRegNet, "memory_logits", None)     # static buffer of graph B (rewritten by every replay)
        # the replay reads the packed-weight buffers that existed at capture time: keep them alive even if a PlanCache
        # is rebuilt later (stale-but-valid until the epoch check re-captures), never a use-after-free
        st["keepalive"] = [c._plans for c in (getattr(mod, "_cache", None) for mod in m.modules()) if isinstance(c, PlanCache)]
        for k in [k for k in self._graphs if k[:-1] == key[:-1]]:      # same call shape, older weights epoch
            del self._graphs[k]
        self._graphs[key] = st
        return st

    def __call__(self, imgs, cam_poses, cam_intr, sample, pre_costs=None, pre_cam_poses=None, mode="val",
                 matching_features=None):
        if mode != "val":
            raise RuntimeError("GraphedForward replays mode='val' only: mode=%r needs host-side masking/metrics "
                               "(call the model eagerly, or evaluate the metrics on the returned outputs)" % (mode,))
        key = self._signature(imgs, pre_costs, mode, matching_features)
        st = self._graphs.get(key)
        if st is None:
            st = self._capture(key, imgs, cam_poses, cam_intr, sample, pre_costs, pre_cam_poses, mode, matching_features)
        # 1. the asynchronous device-to-host copy of the poses goes FIRST into the stream ...
        pending = self.model.camera_begin(cam_poses, cam_intr, pre_cam_poses)
        st["imgs"].copy_(imgs)
        st["poses"].copy_(cam_poses)
        st["intr"].copy_(cam_intr)
        for k, v in sample.items():           # unused by mode='val' arithmetic, refreshed anyway so the buffers never go stale
            st["sample"][k].copy_(v)
        if matching_features is not None:
            st["feats"].copy_(matching_features)
        if pre_costs is not None:
            # the memory volumes (157 MB each at cfg2 size) are inputs of stage B only: their copies run on a side stream
            # beside stage A instead of in front of it.  The side stream starts after everything queued so far (the previous
            # replay of stage B, which reads these buffers; the kernels that produced pre_costs) and stage B waits for it.
            main = torch.cuda.current_stream()
            side = st.get("copy_stream")
            if side is None:
                side = st["copy_stream"] = torch.cuda.Stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                for dst, k, v in zip(st["kv"], pre_costs["keys"], pre_costs["values"]):
                    src = kv_from_pair(k, v)
                    if src.data_ptr() != dst.data_ptr():
                        dst.copy_(src)
                for dst, p in zip(st["mem_poses"], pre_cam_poses):
                    dst.copy_(p)
        # 2. ... then stage A (the 2D networks, ~25 % of a step) is launched; 3. while it runs the host waits for the copy,
        # composes the camera matrices with the reference's own torch-CPU calls and queues their upload; 4. stage B.
        st["graph_a"].replay()
        if pre_costs is not None:
            main.wait_stream(side)
        if pending is not None:
            # the upload of the composed matrices runs on its own stream BESIDE stage A (the host is ready long before stage A ends):
            # ordered after the previous replay of stage B, which read these buffers; stage B waits for it
            main = torch.cuda.current_stream()
            up = st.get("upload_stream")
            if up is None:
                up = st["upload_stream"] = torch.cuda.Stream()
            if st.get("b_done") is not None:
                up.wait_event(st["b_done"])
            else:
                up.wait_stream(main)                     # first replay: after the capture-time runs that read these buffers
            with torch.cuda.stream(up):
                cam = camera.finish(pending)
                for name, t in cam.items():
                    if t is not None:
                        st["cam"][name].copy_(t)
            main.wait_stream(up)
        st["graph_b"].replay()
        if st.get("b_done") is None:
            st["b_done"] = torch.cuda.Event()
        st["b_done"].record()
        outputs, costs, cposes = st["out"]
        # the logit volume that travels with the memory bank (parallel.allgather_memory_bank*): a fresh 4.9 MB tensor, like the memory
        ml = st.get("memory_logits")
        self.memory_logits = ml.clone() if ml is not None else None
        if self.clone_outputs:
            outputs = {k: v.clone() for k, v in outputs.items()}
        # memory handed back to the caller: fresh tensors (they outlive the next replay)
        key_t, value_t = costs["keys"][0], costs["values"][0]
        kv = getattr(value_t, "_estd_kv", None)
        if kv is not None and getattr(key_t, "_estd_kv", None) is kv:
            k2, v2 = kv_views(kv.clone())
        else:
            k2, v2 = key_t.clone(), value_t.clone()
        return outputs, {"keys": [k2], "values": [v2]}, [p.clone() for p in cposes]


class GraphedModule:
    """hipGraph replay of a single-tensor-in / single-tensor-out module call with a static input shape (used for the
    per-frame PSM matching-feature extraction of the streaming harness).  The returned tensor is a fresh clone.
    ``owner``: the model whose weights the module reads -- a capture is keyed on its weights epoch (``load_state_dict`` /
    ``.to()`` bump it and force a re-capture, as in GraphedForward) and keeps the packed-plan buffers of ``owner`` alive;
    call ``invalidate()`` after editing parameters in place."""

    def __init__(self, fn, warmup=2, owner=None):
        self.fn = fn
        self.warmup = warmup
        self.owner = owner
        self._graphs = {}

    def invalidate(self):
        self._graphs.clear()

    def __call__(self, x):
        from . import ops
        key = (tuple(x.shape), x.is_contiguous(memory_format=torch.channels_last), ops.CONV2D_ARITH, ops.CONV2D_ALGO,
               getattr(self.owner, "_estd_weights_epoch", 0))
        st = self._graphs.get(key)
        if st is None:
            st = {"x": x.clone(memory_format=torch.preserve_format)}
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(self.warmup):
                    self.fn(st["x"])
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.no_grad(), torch.cuda.graph(g, capture_error_mode="thread_local"):
                st["y"] = self.fn(st["x"])
            st["graph"] = g
            mods = self.owner.modules() if self.owner is not None else (self.fn.modules() if hasattr(self.fn, "modules") else [])
            st["keepalive"] = [c._plans for c in (getattr(mod, "_cache", None) for mod in mods) if isinstance(c, PlanCache)]
            for k in [k for k in self._graphs if k[:-1] == key[:-1]]:      # same call shape, older weights epoch
                del self._graphs[k]
            self._graphs[key] = st
        st["x"].copy_(x)
        st["graph"].replay()
        return st["y"].clone()
