"""Adapters that let the numpy oracle (oracle.ref_model) call the 2D networks, which are NOT part of the hot path
(SURVEY.md §2): PSM extractor, ResNet encoder and the Monodepth2-style 2D decoder are evaluated with torch on the CPU
from the product package's plain nn.Modules (same parameters as the GPU run).

ORACLE = test infrastructure only (see oracle/__init__.py): used by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.
"""
import numpy as np
import torch


def sd_numpy(module):
    return {k: v.detach().cpu().numpy() for k, v in module.state_dict().items()}


class Nets2D:
    """2D stages for oracle.ref_model: PSM / ResNet / 2D decoder evaluated with torch on CPU."""

    def __init__(self, model=None, decoder=None):
        self.model = model
        self.decoder = decoder if decoder is not None else (model.CostRegNet if model is not None else None)
        # what the last matching() / semantic() call returned: the feature-level parity bar of bench.py / tests/test_gpu_full_config.py
        # compares the GPU path's PSM features and ResNet scales with these (references to the arrays handed to the oracle: no copy)
        self.last_matching = None
        self.last_semantic = None

    @staticmethod
    def _t(a):
        return torch.from_numpy(np.ascontiguousarray(a, np.float32))

    def matching(self, x):
        with torch.no_grad():
            self.last_matching = self.model.matchingFeature(self._t(x)).numpy()
        return self.last_matching

    def semantic(self, x):
        with torch.no_grad():
            self.last_semantic = [f.numpy() for f in self.model.semanticFeature(self._t(x))]
        return self.last_semantic

    def semantic_vs(self, feats):
        with torch.no_grad():
            return self.decoder._semantic_vs([self._t(f) for f in feats]).numpy()

    def refine(self, semantic_vs, logits, feats):
        with torch.no_grad():
            s1, s0 = self.decoder._refine(self._t(semantic_vs), self._t(logits), [self._t(f) for f in feats])
        return s1.numpy(), s0.numpy()
