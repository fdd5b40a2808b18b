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
``zero_copy_memory=True`` (opt-in; what bench.py times) removes the two 157 MB device copies per call that the memory contract above
costs (0.14 ms of a 17.7 ms Joint step):
  * the memory record a call returns is a VIEW of one of a small ring of output buffers the graph writes directly (no clone).  A
    buffer is reused only when it is neither an input of the current call nor the buffer the previous call returned, oldest first,
    and the ring grows on demand -- so a caller that keeps the records it passes back as ``pre_costs`` (the reference's protocols:
    eval_hybrid_seq.py:160-193 keeps the last ``memory_size`` windows) never sees one change; a record that is NOT passed back is
    overwritten two calls later at the earliest: clone it to keep it longer;
  * memory records handed in are read where they lie: a capture is additionally keyed by their addresses and holds a reference to
    them (ring buffers: a bounded set; at most ``MAX_FOREIGN`` other address sets per call shape, then the static-copy path);
  * the tensors of ``outputs`` are overwritten by the next call of ANY signature (the captures of a call shape share nothing, but
    replays alternate between them).
``pipeline=True`` (opt-in, round 6; bit-identical to the serial replay -- tests/test_gpu_pipeline.py -- but NOT a robust gain: Joint step +0.2 .. +1.2 % in
40-step A/B pairs on four boxes, -2 .. -7 % in a 10-step run on a fifth, 200-step buckets between 15.7 and 16.3 ms where the serial replay holds
15.5 +- 0.05, profiles/r6_pipeline_ab.txt): consecutive calls are software-pipelined -- stage A of call k + 1 runs BESIDE stage B of
call k.  The only dependence between consecutive calls of the reference's protocols is the memory record stage B hands on (eval_hybrid.py:229-243,
eval_hybrid_seq.py:160-193): stage A reads the images alone.  Stage A is replayed on one internal stream, stage B on another; calls alternate
between two LANES of captures (own static inputs, own 2D feature buffers, own graph memory pool: the intermediates of lane 1's stage A never
alias lane 0's stage B); stage A of a lane waits for the lane's previous stage B (two calls back), stage B follows its own stage A and the
previous call's stage B in stream order -- which is also what orders the memory record.  WHICH part of stage B the next stage A meets decides
the sign: the first 6.8 ms of a Joint stage B are back-to-back 3D convolutions that own every CU (one 512-thread workgroup with the whole
register file and ~150 KB of LDS per CU, static tile ranges) -- a stage-A workgroup between two of their launches only displaces one of theirs
(released at the start of stage B: 15.9 vs 15.55 ms; with a high-priority stream for either stage 18 ms).  So stage B is captured as TWO graphs,
cut by the decoder behind the key||value convolution, and the next call's stage A is released by the first: it runs beside the heads, the EST
fusion loop (gather / HBM bound kernels between single-volume convolutions), soft-argmin and the 2D refinement.  Contract differences:
  * a call returns when its launches are queued and does NOT make the caller's stream wait for them (that wait is what would serialise the next
    call's stage A behind this call's stage B): call ``join()`` before consuming the returned tensors on the caller's stream (memory records
    passed back as ``pre_costs`` need no join: stage B reads them in its own stream order);
  * the tensors of ``outputs`` survive ONE further call (they are overwritten two calls later).
A capture is keyed by (input shape, number of memory volumes, number of frames whose matching features are handed in, mode, convolution arithmetic,
weights epoch of the model): ``load_state_dict`` / ``.to()`` bump the epoch and force a re-capture; call
``invalidate()`` after editing parameters in place.
"""
import os

import torch

from . import camera
from .hybrid_depth_decoder import kv_from_pair, kv_views
from .layers_op import PlanCache


PIPE_RELEASE = os.environ.get("ESTD_PIPE_RELEASE", "mid")              # pipeline mode A/B: "mid" = the next stage A waits for the previous call's B1; "start" = it does not
SHARE_POOL = os.environ.get("ESTD_GRAPH_SHARE_POOL", "1") == "1"        # one graph memory pool per call shape (0: one per capture)


class GraphedForward:
    MAX_FOREIGN = 2          # zero-copy mode: address sets of memory records that do not lie in the ring, per call shape

    def __init__(self, model, warmup=2, clone_outputs=False, zero_copy_memory=False, reserve_cus=None, pipeline=False):
        """``clone_outputs=True``: the returned ``outputs`` dict holds fresh tensors (18 device copies of [1,1,Hi,Wi] maps per
        Joint call) instead of the graph's static output buffers -- a true drop-in for callers that keep outputs across calls.
        ``zero_copy_memory=True``: see the module docstring.
        ``reserve_cus=(a, b)``: compute units the persistent convolution grids leave free (``estd_set_reserved_cus``) in stage A and in
        stage B -- a grid size is baked into a capture.  A collective that overlaps the NEXT call (the memory-bank exchange of a multi-GPU
        run) starts when stage B ends and runs beside stage A of the next call: an exchange shorter than stage A needs the reserve there
        only, ``(8, 0)``; None = whatever the process-wide setting is at capture time."""
        self.model = model
        self.reserve_cus = reserve_cus
        self.pipeline = bool(pipeline)               # see the module docstring; join() before consuming results on the caller's stream
        self._lane = 0                               # pipeline mode: the lane the NEXT call takes
        self._lane_done = [None, None]               # ... event behind the last stage B of each lane
        self._pipe_streams = None                    # ... (stage-A stream, stage-B stream)
        self.last_event = None                       # ... event behind the stage B of the LAST call (join(event=...))
        self._mid_event = None                       # ... event between the two stage-B graphs of the LAST call
        self.warmup = warmup
        self.clone_outputs = clone_outputs
        self.zero_copy_memory = zero_copy_memory
        self.memory_logits = None                # after a call: fresh copy of DepthHybridDecoder.memory_logits of that call
        self.last_features2d = None              # after a call: stage A's static output buffers {"matching": [V,32,H/4,W/4], "semantic_features": 5 maps}
                                                 # (valid until the next call; the feature-level parity bar of bench.py / the tests reads them)
        self.last_matching = None                # after a call: the PSM features [V,32,H/4,W/4] of its frames (stage A's static buffer: valid until
                                                 # the next call; estdepth_amd.streaming copies the frames it shares with the next call out of it)
        self._graphs = {}
        self._pools = {}                         # call shape (signature without placement / weights epoch) -> graph memory pool shared by its captures
        self._ring = {}                          # zero-copy mode: kv shape -> {"bufs": [...], "stamp": [...], "last": slot, "clock": n}

    def __getattr__(self, name):                 # normalise_images, matchingFeature, ndepths, ... of the wrapped model
        if name in ("model", "warmup", "_graphs", "clone_outputs", "memory_logits", "zero_copy_memory", "_ring", "last_matching", "reserve_cus", "_pools",
                    "last_features2d", "pipeline", "_lane", "_lane_done", "_pipe_streams", "last_event", "_mid_event"):
            raise AttributeError(name)
        return getattr(self.model, name)

    def invalidate(self):
        """drop every captured graph (after in-place edits of parameters, which no epoch counter can see)."""
        self._graphs.clear()
        self._pools.clear()

    def join(self, event=None):
        """pipeline mode: make the CURRENT stream wait for everything the calls so far have queued -- or, with ``event`` = the
        ``last_event`` attribute read right after a call, for THAT call's stage B only (a consumer that waits for call k alone does not
        put the stage A of call k + 2 behind the stage B of call k + 1).  No-op outside pipeline mode."""
        if self._pipe_streams is not None:
            cur = torch.cuda.current_stream()
            if event is not None:
                cur.wait_event(event)
                return
            for s_ in self._pipe_streams:
                cur.wait_stream(s_)

    def _signature(self, imgs, pre_costs, mode, matching_features, placement=(None, None), lane=0):
        n_mem = 0 if pre_costs is None else len(pre_costs["keys"])
        from . import ops, epipolar_transformer as ET
        return (tuple(imgs.shape), n_mem, None if matching_features is None else int(matching_features.shape[0]), mode, ops.CONV3D_ARITH, ops.CONV2D_ARITH,
                ops.CONV3D_ALGO, getattr(ops, "CONV2D_ALGO", None), getattr(ops, "CONV2D_NT", None),      # a graph bakes the kernel choice in:
                # ... the module-level kernel switches tests and tools flip at run time, and the grid sizes of the two stages
                (ops.W3, ops.W3_EXTRA, ops.W2X, ops.W2_XOUT, ET.GATE_IN_CONV), self.reserve_cus,
                lane,                                              # pipeline mode: the lane's own buffers and graph memory pool (0 otherwise)
                self.model.camera_algebra,
                placement,                                         # zero-copy mode: (addresses of the memory records, output ring slot)
                getattr(self.model, "_estd_weights_epoch", 0))     # (last) a captured graph bakes kernel choice and weight buffers in

    def _place_memory(self, imgs, pre_costs, mode, matching_features, lane=0):
        """zero-copy mode: where this call reads its memory records and which ring buffer it writes -> (addresses or None, slot)."""
        m = self.model
        V, Hi, Wi = imgs.shape[1], imgs.shape[3], imgs.shape[4]
        shape = (V - 2, m.ndepths, Hi // 4, Wi // 4, 32)
        ring = self._ring.setdefault(shape, {"bufs": [], "stamp": [], "last": None, "clock": 0})
        resident = {b[-1].data_ptr(): i for i, b in enumerate(ring["bufs"])}      # a call returns the record of its LAST target
        ptrs, busy = None, set()
        if pre_costs is not None:
            kvs = [getattr(v, "_estd_kv", None) if getattr(k, "_estd_kv", None) is getattr(v, "_estd_kv", None) else None
                   for k, v in zip(pre_costs["keys"], pre_costs["values"])]
            if all(kv is not None for kv in kvs):
                ptrs = tuple(kv.data_ptr() for kv in kvs)
                busy = {resident[q] for q in ptrs if q in resident}
                if any(q not in resident for q in ptrs):           # records from elsewhere: a bounded number of captures, then the copy path
                    base = self._signature(imgs, pre_costs, mode, matching_features, lane=lane)
                    seen = {k[-2][0] for k in self._graphs if k[:-2] == base[:-2] and k[-2][0] is not None
                            and any(q not in resident for q in k[-2][0])}
                    if ptrs not in seen and len(seen) >= self.MAX_FOREIGN:
                        ptrs = None
        free = [i for i in range(len(ring["bufs"])) if i not in busy and i != ring["last"]]
        if free:
            slot = min(free, key=lambda i: ring["stamp"][i])
        else:
            ring["bufs"].append(torch.empty(shape, device=imgs.device, dtype=torch.float32))
            ring["stamp"].append(-1)
            slot = len(ring["bufs"]) - 1
        return shape, ptrs, slot

    @staticmethod
    def _mark_written(ring, slot):
        ring["clock"] += 1
        ring["stamp"][slot] = ring["clock"]
        ring["last"] = slot

    def _capture(self, key, imgs, cam_poses, cam_intr, sample, pre_costs, pre_cam_poses, mode, matching_features, kv_out=None):
        m = self.model
        in_place = key[-2][0] is not None            # zero-copy mode: the memory records are read where they lie (a reference is held)
        st = {"imgs": imgs.clone(), "poses": cam_poses.clone(), "intr": cam_intr.clone(),
              "sample": {k: v.clone() for k, v in sample.items()},
              "feats": matching_features.clone() if matching_features is not None else None}
        if pre_costs is not None:
            st["kv"] = [kv_from_pair(k, v) if in_place else kv_from_pair(k, v).clone()
                        for k, v in zip(pre_costs["keys"], pre_costs["values"])]
            st["mem_poses"] = [p.clone() for p in pre_cam_poses]
        # Host camera algebra (estdepth_amd/camera.py): the matrices enter stage B as static inputs.  In "device" mode they
        # are formed by kernels inside stage B from the static pose buffers.
        pending = m.camera_begin(cam_poses, cam_intr, pre_cam_poses)
        st["cam"] = camera.finish(pending) if pending is not None else None

        from . import ops
        if self.pipeline:
            torch.cuda.synchronize()                 # nothing of an earlier call runs beside the eager warm-up / the capture
        prev_reserve = ops.get_reserved_cus() if self.reserve_cus is not None else None

        def reserve(stage):                              # the persistent grids of this stage leave that many CUs free (baked into the capture)
            if self.reserve_cus is not None:
                ops.set_reserved_cus(self.reserve_cus[stage])

        def run_a():                                     # stage A: camera-independent 2D networks, streams joined at the end
            reserve(0)
            return m.forward_2d(st["imgs"], st["feats"], join=True)

        def run_b(feats):                                # stage B: everything downstream of the plane sweep
            reserve(1)
            pc, pp = None, None
            if pre_costs is not None:
                pairs = [kv_views(kv) for kv in st["kv"]]
                pc = {"keys": [k for k, _ in pairs], "values": [v for _, v in pairs]}
                pp = list(st["mem_poses"])
            m.CostRegNet.kv_out = kv_out             # zero-copy mode: the kv records of the targets are written into this ring buffer
            try:
                return m.forward_3d(feats, st["poses"], st["intr"], st["sample"], pc, pp, mode, cam_mats=st["cam"])
            finally:
                m.CostRegNet.kv_out = None

        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(self.warmup):      # hipFuncSetAttribute, plan packing (weight forms are packed on first use): all before capture
                    run_b(run_a())
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            ga, gb, feats, out = self._capture_graphs(key, run_a, run_b)
        finally:
            # the stage reserves are process-wide state (estd_set_reserved_cus): restored whether or not warm-up / capture succeeded --
            # an exception in there must not leave every later eager launch and capture on the wrong grid size
            if prev_reserve is not None:
                ops.set_reserved_cus(prev_reserve)
        if kv_out is not None:
            # zero-copy contract: the record this capture hands out IS the ring slot it was told to write
            rec = getattr(out[1]["values"][0], "_estd_kv", None)
            if rec is None or rec.data_ptr() != kv_out[-1].data_ptr():
                raise RuntimeError("zero-copy memory: the captured forward did not write its memory record into the ring buffer it was given")
        st["graph_a"], st["graph_b"], st["feats2d"], st["out"] = ga, gb, feats, out
        return self._finish_capture(key, st)

    def _capture_graphs(self, key, run_a, run_b):
        ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # ONE memory pool for every capture of a call shape (zero-copy mode makes one capture per (memory addresses, ring slot): the ESTM
        # stream settles at 3 + the start-up signatures): the captures never run concurrently, so the intermediates of one replay may
        # reuse the blocks of another's; what a capture hands out (outputs, 2D features, logits) stays allocated through the references
        # ``st`` keeps.  Before: one private pool of a whole forward's activations per capture (tens of GB each at cfg5 size).
        pool = self._pools.get(key[:-2]) if SHARE_POOL else None
        if pool is None:
            pool = torch.cuda.graph_pool_handle()
            if SHARE_POOL:
                self._pools[key[:-2]] = pool
        # thread_local: other threads (e.g. the RCCL watchdog polling events) may keep issuing HIP calls during capture
        with torch.no_grad(), torch.cuda.graph(ga, pool=pool, capture_error_mode="thread_local"):
            feats = run_a()
        if not self.pipeline:
            with torch.no_grad(), torch.cuda.graph(gb, pool=pool, capture_error_mode="thread_local"):
                out = run_b(feats)
            return ga, gb, feats, out
        # pipeline mode: stage B as TWO graphs, cut by the decoder's split hook behind the key || value convolution: the next call's stage A is
        # released when the first one (B1: the 3D-convolution chain) has finished and runs beside the second (B2: heads, EST loop, refinement)
        gb2 = torch.cuda.CUDAGraph()
        ctx = [torch.cuda.graph(gb, pool=pool, capture_error_mode="thread_local")]
        cut = []

        def split():
            ctx[0].__exit__(None, None, None)
            ctx[0] = torch.cuda.graph(gb2, pool=pool, capture_error_mode="thread_local")
            ctx[0].__enter__()
            cut.append(True)
        self.model.CostRegNet.__dict__["_stage_split"] = split
        try:
            with torch.no_grad():
                ctx[0].__enter__()
                try:
                    out = run_b(feats)
                finally:
                    ctx[0].__exit__(None, None, None)
        finally:
            self.model.CostRegNet.__dict__.pop("_stage_split", None)
        if not cut:
            raise RuntimeError("pipeline mode: the decoder never reached its stage-split point")
        return ga, (gb, gb2), feats, out

    def _finish_capture(self, key, st):
        m = self.model
        st["memory_logits"] = getattr(m.CostRegNet, "memory_logits", None)     # static buffer of graph B (rewritten by every replay)
        # the logit volumes a caller asked the decoder to keep (``keep_logits``): buffers of THIS capture -- the decoder's attribute is pointed
        # at them again after every replay (it would otherwise name the buffers of whichever capture was made last, which a shared pool
        # lets other captures reuse)
        st["last_logits"] = getattr(m.CostRegNet, "last_logits", None) if getattr(m.CostRegNet, "keep_logits", False) else None
        # the replay reads the packed-weight buffers that existed at capture time: keep them alive even if a PlanCache
        # is rebuilt later (stale-but-valid until the epoch check re-captures), never a use-after-free
        st["keepalive"] = [c._plans for c in (getattr(mod, "_cache", None) for mod in m.modules()) if isinstance(c, PlanCache)]
        for k in [k for k in self._graphs if k[:-2] == key[:-2] and k[-1] != key[-1]]:      # same call shape, older weights epoch
            del self._graphs[k]
        self._graphs[key] = st
        return st

    def __call__(self, imgs, cam_poses, cam_intr, sample, pre_costs=None, pre_cam_poses=None, mode="val",
                 matching_features=None):
        if mode != "val":
            raise RuntimeError("GraphedForward replays mode='val' only: mode=%r needs host-side masking/metrics "
                               "(call the model eagerly, or evaluate the metrics on the returned outputs)" % (mode,))
        placement, kv_out, ring = (None, None), None, None
        lane = 0
        if self.pipeline:
            lane, self._lane = self._lane, self._lane ^ 1
        if self.zero_copy_memory:
            shape, ptrs, slot = self._place_memory(imgs, pre_costs, mode, matching_features, lane)
            ring = self._ring[shape]
            placement, kv_out = (ptrs, slot), ring["bufs"][slot]
        key = self._signature(imgs, pre_costs, mode, matching_features, placement, lane)
        st = self._graphs.get(key)
        if st is None:
            st = self._capture(key, imgs, cam_poses, cam_intr, sample, pre_costs, pre_cam_poses, mode, matching_features, kv_out)
        # Stage A needs the images only.  Everything else a call hands over -- the poses (their asynchronous device-to-host copy for
        # the host camera algebra first), intrinsics, ground-truth maps, memory volumes and memory poses -- is an input of the host or of
        # stage B: those copies go to a side stream that starts after everything queued so far (the previous replay of stage B, which
        # reads these buffers; the kernels that produced the arguments) and runs BESIDE stage A instead of in front of it (a dozen
        # 5-10 us copy nodes in series were 0.1 ms of every step with nothing else on the GPU); stage B waits for it.
        main = torch.cuda.current_stream()
        side = st.get("copy_stream")
        if side is None:
            side = st["copy_stream"] = torch.cuda.Stream()
        side.wait_stream(main)
        s_a = s_b = main
        if self.pipeline:
            # stage A / stage B on streams of their own.  Both start behind the producers of this call's arguments (the caller's stream
            # as it stands -- which has NOT been made to wait for the previous call's stage B) and behind the lane's previous stage B
            # (two calls back), which read the static buffers written below and the 2D features stage A is about to overwrite, and whose
            # intermediates share this lane's graph memory pool.
            if self._pipe_streams is None:
                self._pipe_streams = (torch.cuda.Stream(), torch.cuda.Stream())      # (a high-priority stream for either stage: 18.1 / 18.6 ms, measured)
            s_a, s_b = self._pipe_streams
            s_a.wait_stream(main)
            if self._mid_event is not None and PIPE_RELEASE == "mid":
                s_a.wait_event(self._mid_event)          # the previous call's 3D-convolution chain (B1) has the GPU to itself
            done = self._lane_done[lane]
            if done is not None:
                s_a.wait_event(done)
                side.wait_event(done)
        with torch.cuda.stream(side):
            pending = self.model.camera_begin(cam_poses, cam_intr, pre_cam_poses)
            st["poses"].copy_(cam_poses)
            st["intr"].copy_(cam_intr)
            for k, v in sample.items():       # unused by mode='val' arithmetic, refreshed anyway so the buffers never go stale
                st["sample"][k].copy_(v)
            if pre_costs is not None:
                # (157 MB each at cfg2 size; zero-copy mode: the capture reads the records where they lie -- same address, no copy)
                for dst, k, v in zip(st["kv"], pre_costs["keys"], pre_costs["values"]):
                    src = kv_from_pair(k, v)
                    if src.data_ptr() != dst.data_ptr():
                        dst.copy_(src)
                for dst, p in zip(st["mem_poses"], pre_cam_poses):
                    dst.copy_(p)
        with torch.cuda.stream(s_a):
            st["imgs"].copy_(imgs)
            if matching_features is not None:
                st["feats"].copy_(matching_features)
            # stage A (the 2D networks, ~30 % of a step) is launched; while it runs the host waits for the pose copy, composes the
            # camera matrices with the reference's own torch-CPU calls and queues their upload; then stage B.
            st["graph_a"].replay()
        if pending is not None:
            # the upload of the composed matrices goes to the same side stream (the host is ready long before stage A ends): it is ordered
            # after the previous replay of stage B, which read these buffers (the side stream waited for it above), runs BESIDE stage A
            # on a hardware queue of its own, and stage B waits for it.  (A third stream for the uploads shared the main stream's
            # hardware queue: its three 5 us copies ran behind stage A, in series with stage B's first kernels.)
            with torch.cuda.stream(side):
                cam = camera.finish(pending)
                for name, t in cam.items():
                    if t is not None:
                        st["cam"][name].copy_(t)
        if s_b is not s_a:
            s_b.wait_stream(s_a)                         # stage B behind its own stage A (and, in stream order, behind the previous call's stage B)
        s_b.wait_stream(side)
        with torch.cuda.stream(s_b):
            if isinstance(st["graph_b"], tuple):         # pipeline mode: B1, the event that releases the next call's stage A, B2
                st["graph_b"][0].replay()
                self._mid_event = torch.cuda.Event()
                self._mid_event.record(s_b)
                st["graph_b"][1].replay()
            else:
                st["graph_b"].replay()
            outputs, costs, cposes = st["out"]
            self.last_matching = st["feats2d"]["matching"]
            self.last_features2d = st["feats2d"]
            if st.get("last_logits") is not None:
                self.model.CostRegNet.last_logits = st["last_logits"]
            # the logit volume that travels with the memory bank (parallel.allgather_memory_bank*): a fresh 4.9 MB tensor, like the memory
            ml = st.get("memory_logits")
            self.memory_logits = ml.clone() if ml is not None else None
            if self.clone_outputs:
                outputs = {k: v.clone() for k, v in outputs.items()}
            key_t, value_t = costs["keys"][0], costs["values"][0]
            poses_out = [p.clone() for p in cposes]
            if ring is not None:
                # zero-copy mode: the record lies in the ring buffer this capture writes; it is handed out as it is
                self._mark_written(ring, placement[1])
                costs_out = {"keys": [key_t], "values": [value_t]}
            else:
                # memory handed back to the caller: fresh tensors (they outlive the next replay)
                kv = getattr(value_t, "_estd_kv", None)
                if kv is not None and getattr(key_t, "_estd_kv", None) is kv:
                    k2, v2 = kv_views(kv.clone())
                else:
                    k2, v2 = key_t.clone(), value_t.clone()
                costs_out = {"keys": [k2], "values": [v2]}
            if self.pipeline:
                self._lane_done[lane] = torch.cuda.Event()
                self._lane_done[lane].record(s_b)
                self.last_event = self._lane_done[lane]
                # fresh tensors allocated on the stage-B stream are the caller's from here on (after join()): tell the allocator
                for t_ in poses_out + ([self.memory_logits] if self.memory_logits is not None else []) + (list(outputs.values()) if self.clone_outputs else []) \
                        + ([] if ring is not None else [getattr(costs_out["values"][0], "_estd_kv", None), costs_out["keys"][0], costs_out["values"][0]]):
                    if t_ is not None:
                        t_.record_stream(main)
        return outputs, costs_out, poses_out


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
