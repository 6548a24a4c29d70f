"""Epoch loop services around the fused train step -- the counterpart of the notebook's driver
code (reference notebook/notebook.ipynb:597-667): per-epoch train/validate, StepLR, best / latest
checkpoints in the reference's dictionary layout, early stopping (patience), and
``training_history.csv`` with the reference's column names (nb:615-625, 666-667).  Unlike the
reference it can also resume from ``checkpoint_latest.pth``.
"""
from __future__ import annotations

import csv
import os
import time

import torch

from . import checkpoint
from .train import LOSS_KEYS

HISTORY_COLUMNS = ["epoch", "lr", "train_total", "train_steer", "train_throttle", "train_brake",
                   "train_speed", "val_total", "val_steer", "val_throttle", "val_brake",
                   "val_speed", "cmd_FOLLOW", "cmd_LEFT", "cmd_RIGHT", "cmd_STRAIGHT", "time"]


def train_one_epoch(trainer, batches):
    """nb:541-561: mean over batches of the per-batch loss terms.  The per-step losses stay on
    the device; they are summed there and read back once per epoch."""
    trainer.model.train()
    acc = None
    n = 0
    for imgs, speeds, cmds, tgts in batches:
        buf = trainer.train_step(imgs, speeds, cmds, tgts)
        acc = buf[:6].double().clone() if acc is None else acc + buf[:6].double()
        n += 1
    if trainer.reducer is not None:
        # data parallel: the epoch mean over ALL ranks' batches (every rank logs / checks the same).
        # EVERY rank takes part, also one that drew no batch (zeros, n = 0): a rank that skipped
        # the collective would leave the others blocked in it.  Ranks must also have run the SAME
        # number of steps -- each step's gradient all-reduce pairs them up -- so differing counts
        # (a sharded loader that was not cut to a multiple of the global batch) raise here instead
        # of hanging in the next epoch.
        dev = acc.device if acc is not None else trainer.eng.device
        mine = acc if acc is not None else torch.zeros(6, dtype=torch.float64, device=dev)
        cnt = torch.tensor([float(n), float(n) * float(n)], dtype=torch.float64, device=dev)
        packed = trainer.all_reduce_sum(torch.cat([mine, cnt]))
        world = trainer.reducer.world_size
        tot, tot_sq = float(packed[6]), float(packed[7])
        if abs(tot_sq * world - tot * tot) > 0.5:        # sum n^2 * W == (sum n)^2  <=>  all n equal
            raise RuntimeError(f"data-parallel ranks ran different numbers of train steps this epoch "
                               f"(rank {trainer.rank}: {n}; mean {tot / world:.2f}): shard the loader "
                               "to a multiple of the global batch")
        if tot > 0:
            acc, n = packed[:6], tot
    vals = (acc / max(n, 1)).tolist() if acc is not None else [float("nan")] * 6
    if acc is not None:
        trainer.eng.check_status()                  # out-of-range command in the last batch
        if not all(v == v and abs(v) != float("inf") for v in vals):
            raise FloatingPointError(f"non-finite training loss in epoch {trainer.epoch + 1}: "
                                     f"{dict(zip(LOSS_KEYS, vals))}")
    return dict(zip(LOSS_KEYS, vals))


def fit(trainer, train_batches, val_batches, epochs=20, patience=6, out_dir=".", resume=None,
        log=print):
    """``train_batches`` / ``val_batches``: callables returning an iterable of
    (imgs, speeds, cmds, tgts) device batches for one epoch."""
    # data parallel: metrics are all-reduced (Trainer.validate / train_one_epoch), so every rank
    # takes the same decisions; rank 0 alone writes the checkpoints and the history
    writer = trainer.rank == 0

    def barrier():
        if trainer.reducer is not None:
            import torch.distributed as dist
            dist.barrier(group=trainer.reducer.pg)
    if writer:
        os.makedirs(out_dir, exist_ok=True)
    barrier()
    best_path = os.path.join(out_dir, "checkpoint_best.pth")
    latest_path = os.path.join(out_dir, "checkpoint_latest.pth")
    start_epoch, best_val, best_epoch, bad = 1, float("inf"), 0, 0
    history = []
    if resume:
        ck = checkpoint.load(resume, trainer.model, trainer)
        start_epoch = int(ck["epoch"]) + 1
        st = ck.get("loop_state")
        if st is not None:      # keep the best checkpoint / patience / history across the restart
            best_val, best_epoch, bad = float(st["best_val"]), int(st["best_epoch"]), int(st["bad"])
            history = [dict(r) for r in st["history"]]
        log(f"resumed from {resume} at epoch {start_epoch}")
    for epoch in range(start_epoch, epochs + 1):
        t0 = time.time()
        lr = trainer.lr
        tr = train_one_epoch(trainer, train_batches())
        va, cmd = trainer.validate(val_batches())
        trainer.scheduler_step()                                   # nb:604
        dt = time.time() - t0
        row = {"epoch": epoch, "lr": lr, "time": dt}
        for k in ("total", "steer", "throttle", "brake", "speed"):
            row[f"train_{k}"] = tr[k]
            row[f"val_{k}"] = va[k]
        row.update({f"cmd_{k}": v for k, v in cmd.items()})
        history.append(row)
        log(f"epoch {epoch}/{epochs} lr={lr:.6f} train {tr['total']:.4f} val {va['total']:.4f} "
            f"({dt:.1f}s)")
        if va["total"] < best_val:                                 # nb:627-637
            best_val, best_epoch, bad = va["total"], epoch, 0
            if writer:
                checkpoint.save_best(best_path, trainer.model, trainer, epoch, va["total"],
                                     va["steer"], cmd)
        else:
            bad += 1
        if writer:
            checkpoint.save_latest(latest_path, trainer.model, trainer, epoch,    # nb:642-646
                                   loop_state={"best_val": float(best_val),
                                               "best_epoch": best_epoch, "bad": bad,
                                               "history": history})
        barrier()
        if bad >= patience:                                        # nb:650-652
            log(f"early stopping at epoch {epoch}")
            break
    if writer:
        with open(os.path.join(out_dir, "training_history.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=HISTORY_COLUMNS)
            w.writeheader()
            for row in history:
                w.writerow({k: row.get(k, "") for k in HISTORY_COLUMNS})
    barrier()
    return dict(best_val_loss=best_val, best_epoch=best_epoch, history=history)
