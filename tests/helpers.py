"""Shared test plumbing: adapters that let the numpy oracle call the 2D networks (plain torch modules
of the product package, run on CPU) and small comparison utilities."""
import numpy as np
import torch


def sd_numpy(module):
    return {k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


class Nets2D:
    """2D stages for oracle.ref_model: PSM / ResNet / 2D decoder evaluated with torch on CPU."""

    def __init__(self, model=None, decoder=None):
        self.model = model
        self.decoder = decoder if decoder is not None else (model.CostRegNet if model is not None else None)

    @staticmethod
    def _t(a):
        return torch.from_numpy(np.ascontiguousarray(a, np.float32))

    def matching(self, x):
        with torch.no_grad():
            return self.model.matchingFeature(self._t(x)).numpy()

    def semantic(self, x):
        with torch.no_grad():
            return [f.numpy() for f in self.model.semanticFeature(self._t(x))]

    def semantic_vs(self, feats):
        with torch.no_grad():
            return self.decoder._semantic_vs([self._t(f) for f in feats]).numpy()

    def refine(self, semantic_vs, logits, feats):
        with torch.no_grad():
            s1, s0 = self.decoder._refine(self._t(semantic_vs), self._t(logits), [self._t(f) for f in feats])
        return s1.numpy(), s0.numpy()


def checksum(a):
    a = np.asarray(a, np.float64).reshape(-1)
    idx = torch.linspace(0, a.size - 1, 64).long().numpy()      # identical to tools/gen_golden.py
    return np.concatenate([[a.sum(), np.abs(a).sum()], a[idx]])


def err_stats(a, b):
    d = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))
    return float(d.max()), float(d.mean())


def checksum_close(a, ref, tol=2e-4):
    """sum / abs-sum compared relatively, the 64 sampled voxels absolutely."""
    a, ref = np.asarray(a, np.float64), np.asarray(ref, np.float64)
    rel = np.abs(a[:2] - ref[:2]) / np.maximum(np.abs(ref[:2]), 1.0)
    return bool(rel.max() < 1e-5 and np.abs(a[2:] - ref[2:]).max() < tol)
