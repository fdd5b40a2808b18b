"""HIP-graph replay of ``DepthNetHybrid.forward`` (opt-in accelerator; same kernels, same results).

A forward pass is ~500 kernel launches (MIOpen 2D backbones + our HIP kernels) with static shapes; launched
eagerly the GPU idles ~12 % of the step between launches (profiles/).  ``GraphedForward`` captures one forward
per (input shape, number of memory volumes) into a hipGraph on first use and replays it afterwards.

Contract differences from the eager call (documented, not hidden):
  * returned tensors are the graph's static output buffers: they are overwritten by the next call with the
    same signature -- consume or clone them first (eval loops of the reference copy results to the host
    right after each call, eval_hybrid_seq.py:210-236);
  * memory volumes handed in as ``pre_costs`` are copied into static input buffers (one 157 MB device copy
    per volume at cfg2 size), so they may alias the previous call's outputs.
"""
import torch

from .hybrid_depth_decoder import kv_from_pair, kv_views


class GraphedForward:
    def __init__(self, model, warmup=2):
        self.model = model
        self.warmup = warmup
        self._graphs = {}

    def _signature(self, imgs, pre_costs, mode):
        n_mem = 0 if pre_costs is None else len(pre_costs["keys"])
        from . import ops
        return (tuple(imgs.shape), n_mem, mode, ops.CONV3D_ARITH, ops.CONV2D_ARITH)     # a captured graph bakes the kernel choice in

    def _capture(self, key, imgs, cam_poses, cam_intr, sample, pre_costs, pre_cam_poses, mode):
        m = self.model
        st = {"imgs": imgs.clone(), "poses": cam_poses.clone(), "intr": cam_intr.clone(),
              "sample": {k: v.clone() for k, v in sample.items()}}
        if pre_costs is not None:
            st["kv"] = [kv_from_pair(k, v).clone() for k, v in zip(pre_costs["keys"], pre_costs["values"])]
            st["mem_poses"] = [p.clone() for p in pre_cam_poses]

        def run():
            pc, pp = None, None
            if pre_costs is not None:
                pairs = [kv_views(kv) for kv in st["kv"]]
                pc = {"keys": [k for k, _ in pairs], "values": [v for _, v in pairs]}
                pp = list(st["mem_poses"])
            return m(st["imgs"], st["poses"], st["intr"], st["sample"], pc, pp, mode=mode)

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):      # MIOpen algorithm search, hipFuncSetAttribute, plan packing: all before capture
                run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        # thread_local: other threads (e.g. the RCCL watchdog polling events) may keep issuing HIP calls during capture
        with torch.no_grad(), torch.cuda.graph(g, capture_error_mode="thread_local"):
            out = run()
        st["graph"], st["out"] = g, out
        self._graphs[key] = st
        return st

    def __call__(self, imgs, cam_poses, cam_intr, sample, pre_costs=None, pre_cam_poses=None, mode="val"):
        key = self._signature(imgs, pre_costs, mode)
        st = self._graphs.get(key)
        if st is None:
            st = self._capture(key, imgs, cam_poses, cam_intr, sample, pre_costs, pre_cam_poses, mode)
        st["imgs"].copy_(imgs)
        st["poses"].copy_(cam_poses)
        st["intr"].copy_(cam_intr)
        if pre_costs is not None:
            for dst, k, v in zip(st["kv"], pre_costs["keys"], pre_costs["values"]):
                src = kv_from_pair(k, v)
                if src.data_ptr() != dst.data_ptr():
                    dst.copy_(src)
            for dst, p in zip(st["mem_poses"], pre_cam_poses):
                dst.copy_(p)
        st["graph"].replay()
        return st["out"]
