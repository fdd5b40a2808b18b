"""Depth-error suite of the reference (metric.py), restated on numpy fp64-safe reductions.

Same names, arguments and return dictionaries as metric.py so an eval script can switch imports:
    compute_valid_depth_mask   metric.py:4-17     (0.3 m < d < 5 m window, BOTH maps when two are given)
    compute_errors             metric.py:220-259  (11 distances + 'num_valid')
    compute_depth_scale_factor metric.py:262-300
    evaluate_depth             metric.py:303-352
Pinned by tests/golden/g10_metrics.npz (tools/gen_golden.py runs the reference's functions on seeded maps).
Host code: no GPU involved.
"""
import numpy as np

DEFAULT_DISTANCES = ("l1", "l1_inverse", "scale_invariant", "abs_relative", "sq_relative", "avg_log10",
                     "rmse_log", "rmse", "ratio_threshold_1.25", "ratio_threshold_1.5625",
                     "ratio_threshold_1.953125")


def compute_valid_depth_mask(d1, d2=None, min_thred=0.3, max_thred=5.0):
    if d2 is None:
        return (d1 < max_thred) & (d1 > min_thred) & np.isfinite(d1)
    # NaN compares False on both sides, so non-finite pixels drop out exactly as in the reference
    return (d1 < max_thred) & (d2 < max_thred) & (d1 > min_thred) & (d2 > min_thred)


def _checked(a, b):
    ok = np.isfinite(a) & np.isfinite(b) & (a > 0) & (b > 0)
    assert np.all(ok), "depth maps must be preprocessed (finite, positive)"
    return float(a.size)


def _mean_abs(x, n):
    return np.nan if n == 0 else np.sum(np.absolute(x)) / n


def l1(depth1, depth2):
    return _mean_abs(depth1 - depth2, _checked(depth1, depth2))


def l1_inverse(depth1, depth2):
    return _mean_abs(np.reciprocal(depth1) - np.reciprocal(depth2), _checked(depth1, depth2))


def avg_log10(depth1, depth2):
    return _mean_abs(np.log10(depth1) - np.log10(depth2), _checked(depth1, depth2))


def rmse(depth1, depth2):
    n = _checked(depth1, depth2)
    return np.nan if n == 0 else np.sqrt(np.sum(np.square(depth1 - depth2)) / n)


def rmse_log(depth1, depth2):
    n = _checked(depth1, depth2)
    return np.nan if n == 0 else np.sqrt(np.sum(np.square(np.log(depth1) - np.log(depth2))) / n)


def scale_invariant(depth1, depth2):
    n = _checked(depth1, depth2)
    if n == 0:
        return np.nan
    g = np.log(depth1) - np.log(depth2)
    return np.sqrt(np.sum(np.square(g)) / n - np.square(np.sum(g)) / np.square(n))


def abs_relative(depth_pred, depth_gt):
    n = _checked(depth_pred, depth_gt)
    return np.nan if n == 0 else np.sum(np.absolute(depth_pred - depth_gt) / depth_gt) / n


def sq_relative(depth_pred, depth_gt):
    n = _checked(depth_pred, depth_gt)
    return np.nan if n == 0 else np.sum(np.square(depth_pred - depth_gt) / depth_gt) / n


def ratio_threshold(depth1, depth2, threshold):
    assert threshold > 0.
    n = _checked(depth1, depth2)
    if n == 0:
        return np.nan
    return float(np.sum(np.absolute(np.log(depth1) - np.log(depth2)) < np.log(threshold))) / n


_DIST = {"l1": l1, "l1_inverse": l1_inverse, "scale_invariant": scale_invariant, "abs_relative": abs_relative,
         "sq_relative": sq_relative, "avg_log10": avg_log10, "rmse_log": rmse_log, "rmse": rmse}


def compute_errors(depth_pred, depth_gt, distances_to_compute=None):
    """masked prediction/ground-truth -> {'num_valid', <distance>: value, ...} (metric.py:220-259; note the mask call
    passes (gt, pred), which is symmetric)."""
    mask = compute_valid_depth_mask(depth_gt, depth_pred)
    p, g = depth_pred[mask], depth_gt[mask]
    res = {"num_valid": np.sum(mask)}
    for name in (DEFAULT_DISTANCES if distances_to_compute is None else distances_to_compute):
        if name.startswith("ratio_threshold"):
            res[name] = ratio_threshold(p, g, float(name.split("_")[-1]))
        else:
            res[name] = _DIST[name](p, g)
    return res


def compute_depth_scale_factor(depth1, depth2, depth_scaling="abs"):
    """least-squares scale applied to depth1 to match depth2 (metric.py:262-300)."""
    _checked(depth1, depth2)
    if depth_scaling == "log":
        return np.exp(np.mean(np.log(depth2) - np.log(depth1)))
    if depth_scaling not in ("abs", "inv"):
        raise Exception("Unknown depth scaling method")
    a, b = (depth1, depth2) if depth_scaling == "abs" else (np.reciprocal(depth1), np.reciprocal(depth2))
    aa, ab = a * a, a * b
    m = compute_valid_depth_mask(ab)            # the reference windows the PRODUCT map (0.3..5), kept as is
    s_aa, s_ab = np.sum(aa[m]), np.sum(ab[m])
    if not s_aa > 0.:
        print("compute_depth_scale_factor: Norm=0 during scaling")
        return 1.
    return s_ab / s_aa if depth_scaling == "abs" else np.reciprocal(s_ab / s_aa)


def evaluate_depth(translation_gt, depth_gt_in, depth_pred_in, distances_to_compute=None, inverse_gt=True,
                   inverse_pred=True, depth_scaling="abs", depth_pred_max=np.inf):
    """errors without and with optimal scaling of the prediction (metric.py:303-352)."""
    mask = compute_valid_depth_mask(depth_pred_in, depth_gt_in)
    pred, gt = depth_pred_in[mask], depth_gt_in[mask]
    if inverse_gt:
        gt = np.reciprocal(gt)
    if inverse_pred:
        pred = np.reciprocal(pred)
    norm = np.sqrt(translation_gt.dot(translation_gt))
    if not np.isclose(1.0, norm):
        gt = gt / norm
    errs = compute_errors(pred, gt, distances_to_compute)
    scale = compute_depth_scale_factor(pred, gt, depth_scaling=depth_scaling)
    return errs, compute_errors(pred * scale, gt, distances_to_compute)


class RunningErrors:
    """per-frame compute_errors accumulated the way the eval scripts report them: mean over frames of each distance
    (frames without valid pixels are skipped)."""

    def __init__(self):
        self.sums, self.frames = {}, 0

    def add(self, depth_pred, depth_gt):
        e = compute_errors(np.asarray(depth_pred, dtype=np.float64), np.asarray(depth_gt, dtype=np.float64))
        if e["num_valid"] == 0:
            return e
        for k, v in e.items():
            self.sums[k] = self.sums.get(k, 0.0) + float(v)
        self.frames += 1
        return e

    def mean(self):
        return {k: v / max(self.frames, 1) for k, v in self.sums.items()}
