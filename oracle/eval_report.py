"""CPU restatement of the evaluation report (TEST INFRASTRUCTURE ONLY -- never imported by the
product path; see oracle/cilrs_oracle.py's header).

Schema: /root/reference/evaluation_report.json:1-73 (overall_metrics per channel MAE / MSE / RMSE /
Correlation; per_command_metrics n, steer/throttle/brake MAE, steer correlation; steer |error|
percentiles P50..P99; fraction of rows within 0.01 / 0.02 / 0.05 / 0.1).

PARITY UNPINNED for the definitions: the reference ships the report but not the code that wrote
it, so the standard definitions are used (arithmetic means, Pearson correlation, numpy's default
linear-interpolation percentile, inclusive `<=` buckets).  What the published numbers do pin is
checked in tests/test_host.py: RMSE**2 == MSE for every channel, sum of per-command n ==
val_samples, and the key structure.
"""
from __future__ import annotations

import numpy as np

CHANNELS = ("Steer", "Throttle", "Brake", "Speed")
COMMANDS = ("FOLLOW", "LEFT", "RIGHT", "STRAIGHT")     # autonomous_drive.py:406-407
PERCENTILES = (50, 75, 90, 95, 99)
BUCKETS = (0.01, 0.02, 0.05, 0.1)


def _corr(p, t):
    n = p.size
    sp, st = p.sum(), t.sum()
    cov = n * (p * t).sum() - sp * st
    vp = n * (p * p).sum() - sp * sp
    vt = n * (t * t).sum() - st * st
    return float(cov / np.sqrt(vp * vt)) if vp > 0 and vt > 0 else float("nan")


def evaluation_report(controls, pred_speed, target_controls, target_speed, command,
                      checkpoint_epoch=None, model_name="CILRS (ResNet-34)"):
    """All inputs are arrays over the whole validation set (float32 predictions / targets,
    integer commands); arithmetic in float64."""
    pc = np.asarray(controls, dtype=np.float32).astype(np.float64).reshape(-1, 3)
    tc = np.asarray(target_controls, dtype=np.float32).astype(np.float64).reshape(-1, 3)
    ps = np.asarray(pred_speed, dtype=np.float32).astype(np.float64).reshape(-1)
    ts = np.asarray(target_speed, dtype=np.float32).astype(np.float64).reshape(-1)
    cmd = np.asarray(command).astype(np.int64).reshape(-1)
    n = pc.shape[0]
    preds = [pc[:, 0], pc[:, 1], pc[:, 2], ps]
    tgts = [tc[:, 0], tc[:, 1], tc[:, 2], ts]
    overall = {}
    for name, p, t in zip(CHANNELS, preds, tgts):
        d = p - t
        mse = float((d * d).mean())
        overall[name] = {"MAE": float(np.abs(d).mean()), "MSE": mse, "RMSE": float(np.sqrt(mse)),
                         "Correlation": _corr(p, t)}
    per_cmd = {}
    for k, cname in enumerate(COMMANDS):
        m = cmd == k
        nk = int(m.sum())
        if nk == 0:
            continue
        per_cmd[cname] = {
            "n": nk,
            "steer_mae": float(np.abs(pc[m, 0] - tc[m, 0]).mean()),
            "throttle_mae": float(np.abs(pc[m, 1] - tc[m, 1]).mean()),
            "brake_mae": float(np.abs(pc[m, 2] - tc[m, 2]).mean()),
            "steer_corr": _corr(pc[m, 0], tc[m, 0]),
        }
    err = np.abs(pc[:, 0] - tc[:, 0])
    err32 = err.astype(np.float32).astype(np.float64)     # the device keeps |error| as fp32
    return {
        "model": model_name,
        "checkpoint_epoch": checkpoint_epoch,
        "val_samples": int(n),
        "overall_metrics": overall,
        "per_command_metrics": per_cmd,
        "steer_percentiles": {f"P{q}": float(np.percentile(err32, q)) for q in PERCENTILES},
        "steer_accuracy_buckets": {f"within_{b}": float((err <= b).mean()) for b in BUCKETS},
    }
