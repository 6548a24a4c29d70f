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


def save_latest(path, model, trainer, epoch, loop_state=None):
    """notebook/notebook.ipynb:642-646 (+ what a resume needs: `loop_state` carries the epoch
    loop's best_val / best_epoch / bad-epoch count / history rows, plain Python values)."""
    d = {
        "epoch": int(epoch), "model_state_dict": model_state_dict(model),
        "optimizer_state_dict": optimizer_state_dict(trainer),
        "scheduler_state_dict": {"step_size": trainer.cfg.lr_step_size,
                                 "gamma": trainer.cfg.lr_gamma, "last_epoch": trainer.epoch,
                                 "base_lrs": [trainer.cfg.lr], "_last_lr": [trainer.lr]},
    }
    if loop_state is not None:
        d["loop_state"] = loop_state
    torch.save(d, path)


def _numpy_scalar_globals():
    """The only non-tensor globals a checkpoint written the reference's way contains: it stores
    ``np.mean(...)`` values (np.float64) in ``cmd_steer_errors`` (notebook/notebook.ipynb:584,
    631-636), which is why the reference's own loader carries a numpy._core shim
    (autonomous_drive.py:35-44).  They are allow-listed for the weights-only unpickler under both
    module spellings (numpy 1.x wrote ``numpy.core``, 2.x writes ``numpy._core``); nothing else
    from the file is ever executed."""
    import numpy as np
    try:
        from numpy._core.multiarray import scalar
    except ImportError:                                   # numpy 1.x
        from numpy.core.multiarray import scalar
    allow = [(scalar, "numpy._core.multiarray.scalar"), (scalar, "numpy.core.multiarray.scalar"),
             np.dtype]
    for name in ("Float64DType", "Float32DType", "Int64DType", "Int32DType", "BoolDType"):
        t = getattr(getattr(np, "dtypes", None), name, None)
        if t is not None:
            allow.append(t)
    return allow


def load_file(path, map_location=None):
    """torch.load with the weights-only unpickler (executes nothing from the file); numpy scalar
    reconstruction is allow-listed so the reference's own checkpoint_best.pth loads."""
    with torch.serialization.safe_globals(_numpy_scalar_globals()):
        return torch.load(path, map_location=map_location or "cpu", weights_only=True)


def load(path, model, trainer=None, map_location=None):
    """autonomous_drive.py:496-497 (+ the resume path the reference lacks)."""
    ck = load_file(path, map_location)
    model.load_state_dict(ck["model_state_dict"], strict=True)
    if trainer is not None and "optimizer_state_dict" in ck:
        load_optimizer_state_dict(trainer, ck["optimizer_state_dict"])
        if "scheduler_state_dict" in ck:
            trainer.epoch = int(ck["scheduler_state_dict"]["last_epoch"])
    return ck


# torchvision.models.resnet34 children in order (conv1, bn1, relu, maxpool, layer1..4, avgpool, fc):
# the reference drops fc and wraps the rest in nn.Sequential (autonomous_drive.py:366-370), so child
# i becomes ``visual_encoder.i``
_TV_CHILD = {"conv1": 0, "bn1": 1, "layer1": 4, "layer2": 5, "layer3": 6, "layer4": 7}


def trunk_state_from_torchvision(resnet_state_dict):
    """Re-key a ``torchvision.models.resnet34().state_dict()`` (e.g. the ImageNet weights the
    executed notebook starts from, ``ResNet34_Weights.DEFAULT``, notebook/notebook.ipynb:444) the
    way the reference's ``nn.Sequential(*list(resnet.children())[:-1])`` does:
    ``conv1.weight -> visual_encoder.0.weight``, ``layer2.0.bn1.bias -> visual_encoder.5.0.bn1.bias``
    ...; ``fc.*`` is dropped.  Use with ``model.load_state_dict(sd, strict=False)`` (the heads keep
    their initialisation), exactly what constructing the reference class with pretrained weights
    amounts to.  Also works for the ResNet-50 variant (same child order)."""
    out = {}
    for k, v in resnet_state_dict.items():
        head, _, rest = k.partition(".")
        if head == "fc":
            continue
        if head not in _TV_CHILD:
            raise KeyError(f"unexpected torchvision ResNet key {k!r}")
        out[f"visual_encoder.{_TV_CHILD[head]}.{rest}"] = v
    return out
