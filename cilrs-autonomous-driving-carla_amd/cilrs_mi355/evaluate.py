"""Evaluation harness: the metrics of the reference's evaluation_report.json (1-73), accumulated on
the device.

The reference publishes the report (overall MAE / MSE / RMSE / correlation per output, per-command
errors, steer |error| percentiles and accuracy buckets) but not the code that produced it; this
module emits the same JSON schema.  Every batch costs one eval forward (the HIP plan) plus one
`cilrs_eval_accumulate` launch that adds the batch's sums to a 72-double table in HBM; nothing is
copied to the host until `report()`.
"""
from __future__ import annotations

import json

import numpy as np
import torch

from . import _lib as L

CHANNELS = ("Steer", "Throttle", "Brake", "Speed")
COMMANDS = ("FOLLOW", "LEFT", "RIGHT", "STRAIGHT")
PERCENTILES = (50, 75, 90, 95, 99)
BUCKETS = ("within_0.01", "within_0.02", "within_0.05", "within_0.1")


def _corr(n, sp, st, spt, spp, stt):
    cov = n * spt - sp * st
    vp = n * spp - sp * sp
    vt = n * stt - st * st
    return float(cov / np.sqrt(vp * vt)) if vp > 0 and vt > 0 else float("nan")


class Evaluator:
    """acc = Evaluator(model); acc.update(imgs, speeds, cmds, target_controls[, target_speed]);
    acc.report()"""

    def __init__(self, model, capacity: int = 1 << 16):
        self.model = model
        self.device = next(model.parameters()).device
        if self.device.type != "cuda":
            raise RuntimeError("Evaluator needs the model on an MI355X (no CPU fallback)")
        self.nacc = L.lib().cilrs_eval_acc_doubles()
        self.acc = torch.zeros(self.nacc, dtype=torch.float64, device=self.device)
        self.err = torch.empty(capacity, dtype=torch.float32, device=self.device)
        self.count = 0

    def reset(self):
        self.acc.zero_()
        self.count = 0

    def update_predictions(self, controls, pred_speed, target_controls, target_speed, command):
        """Accumulate already-computed predictions (device tensors)."""
        b = int(controls.size(0))
        if b == 0:
            return
        for t, shape in ((controls, (b, 3)), (target_controls, (b, 3)), (pred_speed, (b,)),
                         (target_speed, (b,)), (command, (b,))):
            if tuple(t.shape) != shape or t.device != self.device:
                raise RuntimeError(f"evaluate: expected shape {shape} on {self.device}, got "
                                   f"{tuple(t.shape)} on {t.device}")
        if command.dtype != torch.int64:
            raise RuntimeError("evaluate: command must be int64")
        if self.count + b > self.err.numel():                 # grow the |steer error| log
            grown = torch.empty(max(2 * self.err.numel(), self.count + b), dtype=torch.float32,
                                device=self.device)
            grown[:self.count] = self.err[:self.count]
            self.err = grown
        f32 = lambda t: t.to(torch.float32).contiguous()
        pc, ps, tc, ts = f32(controls), f32(pred_speed), f32(target_controls), f32(target_speed)
        cmd = command.contiguous()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        L.check(L.lib().cilrs_eval_accumulate(
            L.ptr(pc), L.ptr(ps), L.ptr(tc), L.ptr(ts), L.ptr(cmd), b, L.ptr(self.acc),
            L.C.c_void_p(self.err.data_ptr() + 4 * self.count), L.C.c_void_p(stream)))
        self.count += b

    def update(self, images, speeds, commands, target_controls, target_speed=None):
        """One validation batch: eval-mode forward, then accumulate.  target_speed defaults to the
        input speed (the reference trains the speed head to reproduce it,
        notebook/notebook.ipynb:550, 574)."""
        was_training = self.model.training
        self.model.eval()
        with torch.no_grad():
            pc, ps = self.model(images, speeds, commands)
        if was_training:
            self.model.train()
        self.update_predictions(pc, ps, target_controls,
                                speeds if target_speed is None else target_speed, commands)

    def report(self, checkpoint_epoch=None, model_name="CILRS (ResNet-34)"):
        """The evaluation_report.json dictionary (one device->host copy of the table and of the
        |steer error| log)."""
        a = self.acc.cpu().numpy()
        n_total = int(round(a[0]))
        overall = {}
        for c, name in enumerate(CHANNELS):
            n, sp, st, spt, spp, stt, sad, sdd = a[8 * c:8 * c + 8]
            if n == 0:
                raise RuntimeError("evaluate: no samples accumulated")
            mse = float(sdd / n)
            overall[name] = {"MAE": float(sad / n), "MSE": mse, "RMSE": float(np.sqrt(mse)),
                             "Correlation": _corr(n, sp, st, spt, spp, stt)}
        per_cmd = {}
        for k, cname in enumerate(COMMANDS):
            n, s0, s1, s2, sp, st, spt, spp, stt = a[32 + 9 * k:32 + 9 * k + 9]
            if n == 0:
                continue
            per_cmd[cname] = {"n": int(round(n)), "steer_mae": float(s0 / n),
                              "throttle_mae": float(s1 / n), "brake_mae": float(s2 / n),
                              "steer_corr": _corr(n, sp, st, spt, spp, stt)}
        err = self.err[:self.count].cpu().numpy().astype(np.float64)
        return {
            "model": model_name,
            "checkpoint_epoch": checkpoint_epoch,
            "val_samples": n_total,
            "overall_metrics": overall,
            "per_command_metrics": per_cmd,
            "steer_percentiles": {f"P{q}": float(np.percentile(err, q)) for q in PERCENTILES},
            "steer_accuracy_buckets": {b: float(a[68 + i] / n_total)
                                       for i, b in enumerate(BUCKETS)},
        }

    def write_json(self, path, **kw):
        rep = self.report(**kw)
        with open(path, "w") as f:
            json.dump(rep, f, indent=2)
        return rep


def evaluate(model, batches, **report_kw):
    """batches yields (images, speeds, commands, target_controls) like the reference's val_loader
    (notebook/notebook.ipynb:563-585)."""
    ev = Evaluator(model)
    for imgs, speeds, cmds, tgts in batches:
        ev.update(imgs, speeds, cmds, tgts)
    return ev.report(**report_kw)
