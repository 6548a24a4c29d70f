"""checkpoint_best.pth / checkpoint_latest.pth in the reference's dictionary layout.

Write side: notebook/notebook.ipynb:631-636 (best) and :642-646 (latest).  Read side:
model/autonomous_drive.py:496-500 needs ``model_state_dict``, ``epoch``, ``val_loss`` and loads
with ``strict=True``.  Tensors are saved as contiguous fp32 CPU tensors in torch's logical layout
(OIHW conv weights), int64 scalar ``num_batches_tracked``; scalars are plain Python floats (the
reference stored numpy scalars, which is why its loader needs a numpy._core shim, :35-44).
``optimizer_state_dict`` is in torch.optim.Adam's own format (142 entries, parameters() order).
"""
from __future__ import annotations

import torch


def model_state_dict(model):
    return {k: v.detach().cpu().contiguous().clone() for k, v in model.state_dict().items()}


def optimizer_state_dict(trainer):
    eng, cfg = trainer.eng, trainer.cfg
    from .engine import _arena_view
    state = {}
    for i, (_, off, numel, shape) in enumerate(eng.params_layout):
        state[i] = {
            "step": torch.tensor(float(trainer.step_count)),
            "exp_avg": _arena_view(trainer.exp_avg, off, numel, shape).cpu().contiguous().clone(),
            "exp_avg_sq": _arena_view(trainer.exp_avg_sq, off, numel, shape).cpu().contiguous().clone(),
        }
    group = dict(lr=trainer.lr, betas=tuple(cfg.betas), eps=cfg.eps,
                 weight_decay=cfg.weight_decay, amsgrad=False, maximize=False, foreach=None,
                 capturable=False, differentiable=False, fused=None, initial_lr=cfg.lr,
                 params=list(range(len(eng.params_layout))))
    return {"state": state, "param_groups": [group]}


def load_optimizer_state_dict(trainer, sd):
    eng = trainer.eng
    from .engine import _arena_view
    steps = set()
    with torch.no_grad():
        for i, (_, off, numel, shape) in enumerate(eng.params_layout):
            st = sd["state"][i]
            _arena_view(trainer.exp_avg, off, numel, shape).copy_(st["exp_avg"])
            _arena_view(trainer.exp_avg_sq, off, numel, shape).copy_(st["exp_avg_sq"])
            steps.add(int(float(st["step"])))
    if len(steps) != 1:
        raise RuntimeError("per-tensor Adam step counts differ; the flat Adam needs one step")
    trainer.step_count = steps.pop()
    trainer.lr = float(sd["param_groups"][0]["lr"])


def save_best(path, model, trainer, epoch, val_loss, val_steer, cmd_steer_errors, config=None):
    """notebook/notebook.ipynb:631-636."""
    cfg = config if config is not None else dict(trainer.cfg.__dict__)
    torch.save({
        "epoch": int(epoch), "model_state_dict": model_state_dict(model),
        "optimizer_state_dict": optimizer_state_dict(trainer),
        "val_loss": float(val_loss), "val_steer": float(val_steer),
        "config": {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()},
        "cmd_steer_errors": {k: float(v) for k, v in cmd_steer_errors.items()},
    }, path)


def save_latest(path, model, trainer, epoch):
    """notebook/notebook.ipynb:642-646 (+ what a resume needs)."""
    torch.save({
        "epoch": int(epoch), "model_state_dict": model_state_dict(model),
        "optimizer_state_dict": optimizer_state_dict(trainer),
        "scheduler_state_dict": {"step_size": trainer.cfg.lr_step_size,
                                 "gamma": trainer.cfg.lr_gamma, "last_epoch": trainer.epoch,
                                 "base_lrs": [trainer.cfg.lr], "_last_lr": [trainer.lr]},
    }, path)


def load(path, model, trainer=None, map_location=None):
    """autonomous_drive.py:496-497 (+ the resume path the reference lacks)."""
    ck = torch.load(path, map_location=map_location or "cpu", weights_only=True)
    model.load_state_dict(ck["model_state_dict"], strict=True)
    if trainer is not None and "optimizer_state_dict" in ck:
        load_optimizer_state_dict(trainer, ck["optimizer_state_dict"])
        if "scheduler_state_dict" in ck:
            trainer.epoch = int(ck["scheduler_state_dict"]["last_epoch"])
    return ck
