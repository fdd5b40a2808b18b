"""Shared test plumbing: adapters that let the numpy oracle call the 2D networks (plain torch modules
of the product package, run on CPU) and small comparison utilities."""
import numpy as np
import torch


from oracle.nets2d import Nets2D, sd_numpy  # noqa: F401  (shared with bench.py's cpu_baseline leg)


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
