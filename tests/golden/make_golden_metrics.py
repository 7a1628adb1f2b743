"""Generate tests/golden/metrics.npz by RUNNING the reference's utils/metric.py (numpy only, imported unmodified from
/root/reference) on seeded depth maps, with the per-frame call sequence of eval.py:1031-1045 (cal_metrics).
Authoring container only:   python tests/golden/make_golden_metrics.py"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
m = importlib.import_module("utils.metric")


def depth_pair(seed, shape=(40, 56)):
    """Seeded inputs (stored in the fixture next to the reference's outputs)."""
    rng = np.random.default_rng(seed)
    gt = rng.uniform(0.1, 9.5, shape)
    gt[rng.random(shape) < 0.06] = 0.0                                   # Kinect holes
    pred = np.abs(gt * rng.uniform(0.8, 1.3, shape) + rng.normal(0, 0.08, shape)) + 1e-3
    pred[rng.random(shape) < 0.02] = 12.0                                # a few far outliers
    return gt, pred


def main():
    out = {}
    for seed in (1, 2, 3):
        gt, pred = depth_pair(seed)
        out["s%d_gt" % seed], out["s%d_pred" % seed] = gt, pred           # the inputs travel with the fixture
        e = m.compute_errors(pred.copy(), gt.copy())
        for k, v in e.items():
            out["s%d_compute_errors_%s" % (seed, k)] = np.float64(v)
        mask2 = m.compute_valid_depth_mask(pred, gt)
        out["s%d_mask2_count" % seed] = np.int64(mask2.sum())
        for s in ("abs", "log", "inv"):
            out["s%d_scale_%s" % (seed, s)] = np.float64(m.compute_depth_scale_factor(pred[mask2], gt[mask2], s))
        # cal_metrics, per frame (eval.py:1031-1045): clamp the prediction, mask on the ground truth only
        p = pred.copy(); p[p < 0.3] = 0.3; p[p > 8.0] = 8.0
        vm = m.compute_valid_depth_mask(gt, min_thred=0.3, max_thred=8.0)
        g, p = gt[vm], p[vm]
        out["s%d_cal_mean_l1_error" % seed] = np.float64(m.l1(g, p))
        out["s%d_cal_abs.rel" % seed] = np.float64(m.abs_relative(depth_gt=g, depth_pred=p))
        out["s%d_cal_rmse" % seed] = np.float64(m.rmse(g, p))
        out["s%d_cal_scale.inv" % seed] = np.float64(m.scale_invariant(g, p))
        out["s%d_cal_sq.rel" % seed] = np.float64(m.sq_relative(p, g))
        out["s%d_cal_rmse_log" % seed] = np.float64(m.rmse_log(g, p))
        out["s%d_cal_a<1.25" % seed] = np.float64(m.ratio_threshold(g, p, 1.25))
        out["s%d_cal_a<1.25^2" % seed] = np.float64(m.ratio_threshold(g, p, 1.25 * 1.25))
        out["s%d_cal_a<1.25^3" % seed] = np.float64(m.ratio_threshold(g, p, 1.25 * 1.25 * 1.25))
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), seeds=np.array([1, 2, 3]), **out)
    print("metrics.npz: %d values" % len(out))


if __name__ == "__main__":
    main()
