"""Fused training step -- the MI355X counterpart of the body of ``train_one_epoch``
(reference notebook/notebook.ipynb:545-558): forward, loss, backward, [clip], Adam, in that
order, all as HIP kernels on one stream with no host synchronisation (the reference's six
``.item()`` syncs per step, nb:523-526, become one device buffer read on demand).

Also the evaluation loop body (``validate``, nb:563-585) and StepLR (nb:535-536, 604).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import torch

from . import _lib as L

LOSS_KEYS = ("total", "control", "steer", "throttle", "brake", "speed")
CMD_NAMES = {0: "FOLLOW", 1: "LEFT", 2: "RIGHT", 3: "STRAIGHT"}


@dataclass
class TrainConfig:
    """A = documented config (README.md:98-108, configs/train_config.json:24-35; BASELINE.json);
    B = the config the notebook actually executed (notebook/notebook.ipynb:489-502)."""
    name: str = "A"
    lr: float = 2e-4
    weight_decay: float = 1e-4
    loss: str = "mse"                                  # "mse" | "l1"
    loss_weights: tuple = (1.0, 1.0, 1.0, 0.05)        # steer, throttle, brake, speed
    grad_clip: float = 0.0
    dropout: float = 0.0
    betas: tuple = (0.9, 0.999)
    eps: float = 1e-8
    lr_step_size: int = 8
    lr_gamma: float = 0.5


def dropout_seed(torch_seed: int, call: int, rank: int = 0) -> int:
    """64-bit key of one forward's dropout masks (cilrs_net_forward `seed`)."""
    return ((torch_seed * 1000003 + call) ^ (rank * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF


def steplr(cfg: TrainConfig, epoch: int) -> float:
    """lr in effect AFTER `epoch` scheduler steps of torch.optim.lr_scheduler.StepLR(step_size,
    gamma) (nb:535-536: StepLR(8, 0.5), stepped once per epoch, nb:604)."""
    return cfg.lr * (cfg.lr_gamma ** (epoch // cfg.lr_step_size))


CONFIG_A = TrainConfig()
CONFIG_B = TrainConfig(name="B", lr=1e-4, loss="l1", loss_weights=(5.0, 1.0, 1.0, 0.5),
                       grad_clip=1.0, dropout=0.5)


class Trainer:
    def __init__(self, model, cfg: TrainConfig = CONFIG_A, process_group=None, precision="fp32"):
        """precision="bf16": BASELINE.json configs[3]'s "bf16 MFMA path" -- the trunk convolutions
        of the train step multiply bf16 operands (fp32 accumulation); master weights, BatchNorm,
        heads, loss and Adam stay fp32.  Not the reference's arithmetic: agrees with the fp32 step
        to bf16 rounding, not to 1e-4."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.model = model
        self.cfg = cfg
        self.eng = model.engine()
        self.eng.train_precision = precision
        dev = self.eng.device
        n = self.eng.n_arena
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.epoch = 0
        self.lr = cfg.lr
        self.loss_buf = torch.zeros(8, dtype=torch.float32, device=dev)
        self.clip_out = torch.zeros(2, dtype=torch.float32, device=dev)
        self.norm_scratch = torch.empty(L.lib().cilrs_sqnorm_scratch_bytes(), dtype=torch.uint8,
                                        device=dev)
        self._dgrads = {}
        self._w = (C.c_float * 4)(*cfg.loss_weights)
        self._kind = 1 if cfg.loss == "l1" else 0
        self._seed_calls = 0
        self.arena_grad_scale = 1.0
        # True: single-GPU steps without clipping take cilrs_net_backward_step (a segment's Adam
        # update enqueued behind its gradients, on the weight-gradient stream) instead of backward +
        # one Adam launch over the arena -- the same numbers, element by element.  Off by default:
        # the overlapped step is bound by the kernels' resource time, not by its critical path,
        # and the six extra stream forks cost more than the overlap gains (9.24 vs 9.215 ms,
        # tools/dp_overhead.sh).
        self.fuse_optimizer = False
        # data parallel without clipping: Adam per all-reduce bucket as soon as the bucket's
        # averaged gradient is there (the last, small bucket's collective runs under the first two
        # buckets' updates)
        self.bucket_optimizer = True
        self.reducer = None
        self.rank = 0
        if process_group is not None:
            import torch.distributed as dist
            from .parallel import BucketedAllReduce
            self.reducer = BucketedAllReduce(self.eng.grads, process_group,
                                             variant=self.eng.variant)
            self.rank = dist.get_rank(process_group)

    # -- pieces ---------------------------------------------------------------------------------
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.eng.device).cuda_stream)

    def _ensure_engine(self):
        eng = self.model.engine()
        if eng is not self.eng:
            raise RuntimeError("the model was moved/re-created after the Trainer was built")
        return eng

    def loss(self, controls, targets, pred_speed, target_speed, want_grads=True):
        """CILRSLoss (nb:514-527) / the documented MSE loss; returns device buffer [6] in
        LOSS_KEYS order, plus (dcontrols, dpred_speed) when want_grads."""
        b = controls.size(0)
        dc = dp = None
        if want_grads:
            if b not in self._dgrads:
                self._dgrads[b] = (torch.empty(b, 3, device=controls.device),
                                   torch.empty(b, device=controls.device))
            dc, dp = self._dgrads[b]
        L.check(L.lib().cilrs_loss_fwd_bwd(
            L.ptr(controls), L.ptr(targets.contiguous()), L.ptr(pred_speed),
            L.ptr(target_speed.contiguous()), b, self._kind, self._w, 1.0, L.ptr(dc), L.ptr(dp),
            L.ptr(self.loss_buf), self._stream()))
        return self.loss_buf, dc, dp

    def optimizer_step(self, grad_scale=1.0):
        """clip_grad_norm_ (nb:553-554) + Adam.step (nb:555) over the flat arena."""
        lib = L.lib()
        eng, cfg = self.eng, self.cfg
        clip_ptr = None
        # arena * arena_grad_scale = the (rank-averaged) gradient this step's clip + Adam consume
        self.arena_grad_scale = 1.0 if cfg.grad_clip > 0 else float(grad_scale)
        if cfg.grad_clip > 0:
            if grad_scale != 1.0:
                L.check(lib.cilrs_scale(L.ptr(eng.grads), eng.n_arena, None, grad_scale,
                                        self._stream()))
                grad_scale = 1.0
            L.check(lib.cilrs_grad_sqnorm(L.ptr(eng.grads), eng.n_arena, cfg.grad_clip,
                                          L.ptr(self.norm_scratch), L.ptr(self.clip_out),
                                          self._stream()))
            clip_ptr = L.ptr(self.clip_out)
        self.step_count += 1
        eng.weights_epoch += 1                # the fused Adam writes the parameter arena in place
        L.check(lib.cilrs_adam_step(L.ptr(eng.params), L.ptr(eng.grads), L.ptr(self.exp_avg),
                                    L.ptr(self.exp_avg_sq), eng.n_arena, self.lr, cfg.betas[0],
                                    cfg.betas[1], cfg.eps, cfg.weight_decay, self.step_count,
                                    clip_ptr, grad_scale, self._stream()))

    # -- one iteration of train_one_epoch's loop (nb:549-555) ------------------------------------
    def train_step(self, imgs, speeds, cmds, tgts):
        """Returns the device loss buffer [>=6] (LOSS_KEYS order); reading it is the only sync."""
        eng = self._ensure_engine()
        if not self.model.training:
            self.model.train()
        seed = self.next_dropout_seed() if self.cfg.dropout > 0 else 0
        controls, pred_speed, pl = eng.run_forward(imgs, speeds, cmds, True, self.cfg.dropout,
                                                   seed)
        # `speeds` is both an input and the speed head's regression target (nb:550)
        _, dc, dp = self.loss(controls, tgts, pred_speed, speeds)
        if self.reducer is None and self.cfg.grad_clip <= 0 and self.fuse_optimizer:
            # backward + Adam in one call: a segment's update runs as soon as its gradients are
            # complete (clipping needs the global norm first and keeps the two-call path below)
            self.step_count += 1
            self.arena_grad_scale = 1.0
            eng.run_backward_step(pl, dc, dp, self.exp_avg, self.exp_avg_sq, self.lr,
                                  self.cfg.betas, self.cfg.eps, self.cfg.weight_decay,
                                  self.step_count)
        elif self.reducer is None:
            eng.run_backward(pl, dc, dp)
            self.optimizer_step(1.0)
        elif self.cfg.grad_clip <= 0 and self.bucket_optimizer:
            # data parallel without clipping: Adam per all-reduce bucket, as soon as the bucket's
            # averaged gradient is there (the last bucket's collective runs under the first two
            # buckets' updates); same numbers as one launch over the arena
            self.step_count += 1
            eng.weights_epoch += 1
            scale = 1.0 / self.reducer.world_size
            self.arena_grad_scale = scale
            self.reducer.backward_and_reduce(
                eng, pl, dc, dp, after_bucket=lambda _i, b, e: self._adam_range(b, e, scale))
        else:
            self.reducer.backward_and_reduce(eng, pl, dc, dp)
            self.optimizer_step(1.0 / self.reducer.world_size)
        return self.loss_buf

    def _adam_range(self, begin, end, grad_scale):
        """cilrs_adam_step over the arena range [begin, end) (step_count already advanced)."""
        eng, cfg = self.eng, self.cfg
        L.check(L.lib().cilrs_adam_step(
            L.ptr(eng.params[begin:end]), L.ptr(eng.grads[begin:end]),
            L.ptr(self.exp_avg[begin:end]), L.ptr(self.exp_avg_sq[begin:end]), end - begin, self.lr,
            cfg.betas[0], cfg.betas[1], cfg.eps, cfg.weight_decay, self.step_count, None,
            float(grad_scale), self._stream()))

    def next_dropout_seed(self):
        """Seed of the next train step's dropout masks: torch's seed, the step count and the
        data-parallel rank (replicas must not drop the same units)."""
        self._seed_calls += 1
        return dropout_seed(torch.initial_seed(), self._seed_calls, self.rank)

    def losses(self, check=True):
        """Host dict of the last step's loss terms (one device->host copy).  With check=True this
        synchronisation point also surfaces what the reference would have raised or shown during
        the step: an out-of-range command (torch.gather, autonomous_drive.py:397-398) and a
        non-finite loss (the reference prints NaN from its per-step .item(), nb:523-526)."""
        v = self.loss_buf[:6].tolist()
        out = dict(zip(LOSS_KEYS, v))
        if check:
            self.eng.check_status()
            if not all(x == x and abs(x) != float("inf") for x in v):
                raise FloatingPointError(f"non-finite loss after step {self.step_count}: {out}")
        return out

    def grad_norm(self):
        return float(self.clip_out[0])

    # -- StepLR (nb:535-536, 604) ----------------------------------------------------------------
    def scheduler_step(self):
        self.epoch += 1
        self.lr = steplr(self.cfg, self.epoch)

    # -- validate() (nb:563-585) -------------------------------------------------------------------
    @torch.no_grad()
    def validate(self, batches):
        """mean of batch means + per-command mean |steer error|, like the reference."""
        self.model.eval()
        sums = torch.zeros(6, dtype=torch.float64)
        # (the reference's validate is written for its four commands, nb:566; a model built with more
        #  branches gets one row per command, named cmd<i> beyond the reference's four names)
        nc = max(4, int(getattr(self.model, "num_commands", 4)))
        cmd_sum = torch.zeros(nc, dtype=torch.float64, device=self.eng.device)
        cmd_cnt = torch.zeros(nc, dtype=torch.float64, device=self.eng.device)
        n = 0
        for imgs, speeds, cmds, tgts in batches:
            pc, ps = self.model(imgs, speeds, cmds)
            buf, _, _ = self.loss(pc, tgts, ps, speeds, want_grads=False)
            sums += buf[:6].double().cpu()
            self.eng.check_status()
            n += 1
            serr = (pc[:, 0] - tgts[:, 0]).abs().double()
            cmd_sum.index_add_(0, cmds, serr)
            cmd_cnt.index_add_(0, cmds, torch.ones_like(serr))
        # data parallel: every rank validates its own shard; the SUMS are all-reduced so that all
        # ranks see the same metrics and take the same early-stopping / checkpoint decisions (a
        # rank that stopped alone would leave the others hanging in the next gradient all-reduce)
        packed = torch.cat([sums.to(self.eng.device), torch.tensor([float(n)], dtype=torch.float64,
                                                                  device=self.eng.device),
                            cmd_sum, cmd_cnt])
        packed = self.all_reduce_sum(packed)
        sums, n = packed[:6].cpu(), float(packed[6])
        cs, cc = packed[7:7 + nc].cpu(), packed[7 + nc:7 + 2 * nc].cpu()
        out = {k: float(sums[i]) / max(n, 1.0) for i, k in enumerate(LOSS_KEYS)}
        names = [CMD_NAMES.get(i, f"cmd{i}") for i in range(nc)]
        cmd_avg = {names[i]: (float(cs[i] / cc[i]) if cc[i] > 0 else float("nan"))
                   for i in range(nc)}
        return out, cmd_avg

    def all_reduce_sum(self, t):
        """Sum of a (device) tensor over the data-parallel ranks; the tensor itself without them."""
        if self.reducer is None:
            return t
        import torch.distributed as dist
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.reducer.pg)
        return t
