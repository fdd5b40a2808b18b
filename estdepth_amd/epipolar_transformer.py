"""EpipolarTransformer (attention + ConvGRU) with the reference's constructor, parameter names and
forward signature (transformer/epipolar_transformer.py:10-83), executed by HIP kernels:

    attention                      -> estd_warp_attention (fused with the volume warps) / estd_attention_prewarped
    gate_conv, output_conv         -> estd_conv3d_k3 (fp32 MFMA) with GroupNorm partial sums in the epilogue
    GroupNorm(1,16) x3             -> estd_groupnorm_finalize + the two elementwise GRU kernels
    sigmoid / tanh / blend         -> estd_gru_reset_apply, estd_gru_blend
"""
import os

import torch
import torch.nn as nn

from . import ops
from .layers_op import PlanCache


GATE_IN_CONV = os.environ.get("ESTD_GATE_IN_CONV", "1") == "1"      # A/B: 0 = the reset gate as a pass of its own (estd_gru_reset_apply)


class EpipolarTransformer(nn.Module):
    def __init__(self, input_channel, output_channel, kernel_size):
        super().__init__()
        if input_channel != 16 or output_channel != 16 or kernel_size != 3:
            raise RuntimeError("the HIP EpipolarTransformer is specialised for 16+16 channels, kernel 3 "
                               "(hybrid_depth_decoder.py:82)")
        gru_input_channel = input_channel + output_channel
        self.output_channel = output_channel
        self.gate_conv = nn.Conv3d(gru_input_channel, output_channel * 2, kernel_size, padding=1)
        self.reset_gate_norm = nn.GroupNorm(1, output_channel, 1e-5, True)
        self.update_gate_norm = nn.GroupNorm(1, output_channel, 1e-5, True)
        self.output_conv = nn.Conv3d(gru_input_channel, output_channel, kernel_size, padding=1)
        self.output_norm = nn.GroupNorm(1, output_channel, 1e-5, True)
        self._cache = PlanCache()

    def _plans(self):
        def build():
            dev = self.gate_conv.weight.device
            one32, one16 = torch.ones(32), torch.ones(16)
            gate = ops.Conv3dPlan(self.gate_conv.weight, list(range(32)), None, list(range(32)), 2,
                                  one32, self.gate_conv.bias.detach().cpu(), device=dev)
            outp = ops.Conv3dPlan(self.output_conv.weight, list(range(32)), None, list(range(16)), 1,
                                  one16, self.output_conv.bias.detach().cpu(), device=dev)
            return gate, outp
        return self._cache.get(self, build)

    def gru(self, xh, dims, out_value, out_stride, before_write=None):
        """xh [D,H,W,32] = [x | h]  ->  writes u*h + (1-u)*tanh(GN(o)) to out_value (16 ch, out_stride)."""
        D, H, W = dims
        gate, outp = self._plans()
        n_vox = D * H * W
        nblk = ops.conv3d_grid(1, D, H, W)
        part = torch.empty(nblk * 4, device=xh.device, dtype=torch.float64)
        ru = torch.empty((D, H, W, 32), device=xh.device, dtype=torch.float32)
        gate.run(xh, (1, D, H, W), out=ru, out_stride=32, stats_partials=part)                   # :36-37
        st_ru = ops.groupnorm_finalize(part, nblk, 16.0 * n_vox, self.reset_gate_norm.eps)       # :44-45 statistics
        part2 = torch.empty(nblk * 4, device=xh.device, dtype=torch.float64)
        o_raw = torch.empty((D, H, W, 16), device=xh.device, dtype=torch.float32)
        if GATE_IN_CONV and ops.CONV3D_ALGO == "wino2" and ops.CONV3D_ARITH == "f32":
            # :46,:51 the reset gate sigmoid(GN(r)) * h is formed in the output convolution's own plane loads (no [x | r*h] volume: one
            # 393 MB pass less per target)
            outp.run(xh, (1, D, H, W), out=o_raw, out_stride=16, stats_partials=part2,
                     gate=(ru, st_ru, self.reset_gate_norm.weight, self.reset_gate_norm.bias))       # :52
        else:
            xrh = ops.gru_reset_apply(xh, ru, st_ru, self.reset_gate_norm.weight, self.reset_gate_norm.bias)   # :46,:51
            outp.run(xrh, (1, D, H, W), out=o_raw, out_stride=16, stats_partials=part2)              # :52
        st_o = ops.groupnorm_finalize(part2, nblk, 16.0 * n_vox, self.output_norm.eps)           # :53
        if before_write is not None:
            before_write()            # e.g. join a side stream that still reads the value half we are about to overwrite
        ops.gru_blend(xh, ru, o_raw, st_ru, st_o, self.update_gate_norm.weight, self.update_gate_norm.bias,
                      self.output_norm.weight, self.output_norm.bias, out_value, out_stride)     # :47,:82-83

    def fuse_kv(self, kv_target, kv_sources, mats, depth_values, depth_min, depth_interval, before_write=None):
        """Fast path used by DepthHybridDecoder: warp every source kv into the target frustum, attend,
        run the GRU and overwrite the VALUE half of ``kv_target`` in place (values[i] = fused,
        hybrid_depth_decoder.py:253)."""
        D, H, W, _ = kv_target.shape
        xh = ops.warp_attention(kv_target, kv_sources, mats, depth_values, depth_min, depth_interval)
        self.gru(xh, (D, H, W), kv_target, 32, before_write=before_write)
        return kv_target

    def forward(self, target_key, target_value, warped_values=None, warped_keys=None):
        """Level-1 signature: NCDHW tensors, already-warped lists (transformer/epipolar_transformer.py:56)."""
        B, C, D, H, W = target_value.shape
        outs = []
        for b in range(B):
            kv_t = torch.empty((D, H, W, 32), device=target_value.device, dtype=torch.float32)
            ops.cdhw_to_vol(target_value[b].contiguous(), kv_t, 32, 0)
            ops.cdhw_to_vol(target_key[b].contiguous(), kv_t, 32, 16)
            if warped_values is not None:
                srcs = []
                for wv, wk in zip(warped_values, warped_keys):
                    kv = torch.empty_like(kv_t)
                    ops.cdhw_to_vol(wv[b].contiguous(), kv, 32, 0)
                    ops.cdhw_to_vol(wk[b].contiguous(), kv, 32, 16)
                    srcs.append(kv)
                xh = ops.attention_prewarped(kv_t, srcs)
            else:
                xh = kv_t.clone()
                xh[..., 16:] = 0                                                                 # :78-79 h = 0
            fused = torch.empty((D, H, W, 16), device=kv_t.device, dtype=torch.float32)
            self.gru(xh, (D, H, W), fused, 16)
            outs.append(fused.permute(3, 0, 1, 2))
        return torch.stack(outs, 0)
