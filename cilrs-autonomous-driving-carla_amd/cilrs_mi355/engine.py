"""Host side of the HIP engine: flat parameter arenas, per-shape plans, autograd bridge.

PyTorch is used here for device memory, streams and autograd plumbing only; every number is
produced by libcilrs_hip.so (see include/cilrs_hip.h).

Memory layout (all fp32, on one device):
  * ``params``   flat arena in ``nn.Module.parameters()`` order (cilrs_param_info); each
                 ``nn.Parameter.data`` is a VIEW into it.  Conv weights are stored OHWI and exposed
                 as logical-OIHW permuted views (= torch channels_last memory), so
                 ``state_dict()`` / ``load_state_dict()`` / any ``torch.optim`` work unchanged
                 (reference contract: model/autonomous_drive.py:496-497,
                 notebook/notebook.ipynb:533, 631-636).
  * ``grads``    same layout; backward writes it, Adam / clip / all-reduce read it.
  * ``bn``       running_mean / running_var arena + int64[36] num_batches_tracked.
  * workspace    one allocation per (batch, H, W) plan holding every saved activation.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L

import operator

_VERSION_OF = operator.attrgetter("_version")

SEG_NAMES = ("heads", "layer4", "layer3", "layer2", "layer1", "stem")


def _layout(variant=0):
    """(parameter layout, BatchNorm layout) of architecture `variant` (0: the reference's
    ResNet-34 network; 1: the ResNet-50 variant of BASELINE.json configs[3])."""
    lib = L.lib()
    params = []
    name = C.create_string_buffer(256)
    for i in range(lib.cilrs_variant_num_params(variant)):
        off, numel, ndim = L.sz(), L.sz(), L.i32()
        shape = (L.i32 * 4)()
        L.check(lib.cilrs_variant_param_info(variant, i, name, 256, C.byref(off), C.byref(numel),
                                             C.byref(ndim), shape))
        params.append((name.value.decode(), off.value, numel.value,
                       tuple(shape[k] for k in range(ndim.value))))
    bns = []
    for j in range(lib.cilrs_variant_num_bn(variant)):
        ch, rm, rv = L.i32(), L.sz(), L.sz()
        L.check(lib.cilrs_variant_bn_info(variant, j, name, 256, C.byref(ch), C.byref(rm),
                                          C.byref(rv)))
        bns.append((name.value.decode(), ch.value, rm.value, rv.value))
    return params, bns


def segment_ranges(variant=0):
    """[(begin, end)] float ranges of the gradient arena, in backward execution order."""
    lib = L.lib()
    out = []
    for s in range(6):
        b, e = L.sz(), L.sz()
        L.check(lib.cilrs_variant_segment_range(variant, s, C.byref(b), C.byref(e)))
        out.append((b.value, e.value))
    return out


def _arena_view(arena, off, numel, shape):
    flat = arena[off:off + numel]
    if len(shape) == 4:                       # OHWI storage, logical OIHW
        o, i, h, w = shape
        return flat.view(o, h, w, i).permute(0, 3, 1, 2)
    return flat.view(shape)


PLAN_BF16_TRAIN = 1       # include/cilrs_hip.h CILRS_PLAN_BF16_TRAIN


class Plan:
    """cilrs_net for one (batch, H, W) + its workspace."""

    def __init__(self, device, batch, h, w, variant=0, flags=0):
        lib = L.lib()
        handle = L.vp()
        L.check(lib.cilrs_net_create_ex(variant, batch, h, w, flags, C.byref(handle)))
        self.flags = flags
        self.handle = handle
        self.batch, self.h, self.w = batch, h, w
        nbytes = lib.cilrs_net_workspace_bytes(handle)
        self.workspace = torch.empty(nbytes, dtype=torch.uint8, device=device)
        assert self.workspace.data_ptr() % 256 == 0
        # int32[4] status words inside the workspace (word 0: a command outside 0..3 was seen by
        # a forward SINCE THE LAST check_status() -- the reference's torch.gather raises there,
        # autonomous_drive.py:397; kernels only ever set the words, so one read at the end of an
        # epoch covers every batch; word 1: a grid barrier of the persistent launch gave up)
        off = lib.cilrs_net_status_offset(handle)
        self.status = self.workspace[off:off + 16].view(torch.int32)
        self.status.zero_()
        self.generation = 0

    def __del__(self):
        try:
            if self.handle:
                L.lib().cilrs_net_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def take_status(self):
        """Synchronising read-and-clear of the status words: (bad_command, barrier_gave_up)."""
        dev = self.workspace.device
        torch.cuda.synchronize(dev)
        st = self.status.tolist()
        if st[0] != 0 or st[1] != 0:
            self.status.zero_()
            torch.cuda.synchronize(dev)
        return st[0] != 0, st[1] != 0

    def check_status(self):
        """Synchronising read of the status words of the last forward on this plan; raises what
        the reference's ``all_out.gather(0, idx)`` raises for an out-of-range command."""
        # forwards may have run on other streams than torch's current one (Predictor, inference
        # lanes): wait for the whole device before reading, and again after clearing, so that a
        # set racing the clear cannot be lost
        dev = self.workspace.device
        torch.cuda.synchronize(dev)
        st = self.status.tolist()
        if st[0] != 0 or st[1] != 0:
            self.status.zero_()           # the words are sticky: "since the last check"
            torch.cuda.synchronize(dev)
        if st[1] != 0:
            raise RuntimeError("CILRS persistent forward: a grid barrier gave up (another "
                               "persistent launch was holding the device); outputs are NaN")
        if st[0] != 0:
            raise RuntimeError("CILRS.forward: command index out of range (expected 0..num_commands-1); "
                               "torch.gather raises 'index out of bounds' here "
                               "(model/autonomous_drive.py:397-398)")

    # per-kernel hipEvent timing ------------------------------------------------------------
    def profile(self, on: bool):
        L.check(L.lib().cilrs_net_profile_enable(self.handle, 1 if on else 0))

    def profile_reset(self):
        L.check(L.lib().cilrs_net_profile_reset(self.handle))

    def wino_convs(self) -> int:
        """Convolutions of this plan's train step that run on the Winograd kernel (0: none)."""
        return int(L.lib().cilrs_net_wino_convs(self.handle))

    def profile_table(self):
        lib = L.lib()
        L.check(lib.cilrs_net_profile_collect(self.handle))
        rows = {}
        label = C.create_string_buffer(128)
        for i in range(lib.cilrs_net_profile_count(self.handle)):
            calls, ms, fl, by = C.c_longlong(), L.f64(), L.f64(), L.f64()
            L.check(lib.cilrs_net_profile_entry(self.handle, i, label, 128, C.byref(calls),
                                                C.byref(ms), C.byref(fl), C.byref(by)))
            rows[label.value.decode()] = dict(calls=calls.value, ms=ms.value, flops=fl.value,
                                              bytes=by.value)
        return rows


class Engine:
    def __init__(self, module, variant=0):
        lib = L.lib()                                    # raises if the extension is missing
        self.variant = variant
        named = list(module.named_parameters())
        if not named:
            raise RuntimeError("CILRS has no parameters")
        device = named[0][1].device
        if device.type != "cuda":
            raise RuntimeError(
                "CILRS.forward runs only on a ROCm device (model.to('cuda')): the MI355X HIP "
                "engine has no CPU fallback")
        self.device = device
        self.module = module
        self.params_layout, self.bn_layout = _layout(variant)
        names = [n for n, _ in named]
        want = [p[0] for p in self.params_layout]
        if names != want:
            raise RuntimeError("module parameter names differ from the engine layout")
        n_arena = lib.cilrs_variant_param_arena_floats(variant)
        self.n_arena = n_arena
        self.params = torch.zeros(n_arena, dtype=torch.float32, device=device)
        self.grads = torch.zeros(n_arena, dtype=torch.float32, device=device)
        self.bn = torch.zeros(lib.cilrs_variant_bn_arena_floats(variant), dtype=torch.float32,
                              device=device)
        self.nbt = torch.zeros(len(self.bn_layout), dtype=torch.int64, device=device)
        self.param_views, self.grad_views = [], []
        with torch.no_grad():
            for (name, p), (_, off, numel, shape) in zip(named, self.params_layout):
                if p.dtype != torch.float32 or tuple(p.shape) != shape:
                    raise RuntimeError(f"{name}: expected float32 {shape}, got {p.dtype} "
                                       f"{tuple(p.shape)}")
                v = _arena_view(self.params, off, numel, shape)
                v.copy_(p.data)
                old_grad = p.grad
                p.data = v
                g = _arena_view(self.grads, off, numel, shape)
                if old_grad is not None:
                    g.copy_(old_grad)
                    p.grad = g
                self.param_views.append(v)
                self.grad_views.append(g)
            mods = dict(module.named_modules())
            for j, (prefix, ch, rm, rv) in enumerate(self.bn_layout):
                bnm = mods[prefix]
                for attr, off in (("running_mean", rm), ("running_var", rv)):
                    v = self.bn[off:off + ch]
                    v.copy_(getattr(bnm, attr))
                    setattr(bnm, attr, v)          # registered buffer keeps its name
                v = self.nbt[j]
                v.copy_(bnm.num_batches_tracked)
                bnm.num_batches_tracked = v
        self._first_param = named[0][1]
        self._last_param = named[-1][1]
        self.plans = {}
        self.bufs = {}
        self.last_plan = None
        # "fp32" (the reference's arithmetic) or "bf16": train-mode trunk convolutions on the bf16
        # matrix pipe (include/cilrs_hip.h CILRS_PLAN_BF16_TRAIN); set through
        # Trainer(..., precision=) or directly before the first train-mode forward
        self.train_precision = "fp32"
        self._scratch_grads = None        # second gradient arena (autograd accumulation only)
        # False: loss.backward() hands autograd CLONES of the gradient arena (torch semantics: they
        # stay valid whatever runs next).  True: views of the arena, valid until the next backward.
        self.zero_copy_grads = False
        self.weights_epoch = 1            # bumped by every kernel-side write to params / BN buffers
        # torch-visible in-place edits (`with torch.no_grad(): p.mul_(2)`) bump the tensors' version
        # counters: the eval-mode `model(...)` call -- the reference's own inference call -- polls
        # them (poll_versions), the latency paths rely on the weights_changed() contract instead
        self._versioned = tuple(p for _n, p in named) + tuple(
            b for m in module.modules() if isinstance(m, torch.nn.BatchNorm2d)
            for b in (m.running_mean, m.running_var))
        self._version_sum = sum(map(_VERSION_OF, self._versioned))

    # ------------------------------------------------------------------------------------------
    def is_attached(self) -> bool:
        a, b = self._first_param, self._last_param
        return (a.data_ptr() == self.params.data_ptr() and a.device == self.device
                and b.device == self.device)

    def plan(self, batch, h, w, lane=0) -> Plan:
        """The cilrs_net + workspace for one (batch, H, W).  `lane` > 0 gives further,
        independent plans of the same geometry: concurrent inference streams (each on its own
        HIP stream) must not share a workspace."""
        flags = PLAN_BF16_TRAIN if self.train_precision == "bf16" else 0
        key = (batch, h, w) if lane == 0 else (batch, h, w, lane)
        if flags:
            key = key + ("flags", flags)
        pl = self.plans.get(key)
        if pl is None:
            pl = Plan(self.device, batch, h, w, self.variant, flags)
            self.plans[key] = pl
            pl.bufs = L.Buffers(self.params.data_ptr(), self.grads.data_ptr(),
                                self.bn.data_ptr(), self.nbt.data_ptr(),
                                pl.workspace.data_ptr())
            self.bufs[key] = pl.bufs
        return pl

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def weights_key(self) -> int:
        """Non-zero value that changes whenever the parameters or BatchNorm buffers may have
        changed.  O(1) -- it sits on the single-frame latency path.  `weights_epoch` is bumped by
        every kernel-side write (train-mode forward, fused Adam), by load_state_dict and by every
        train()/eval() switch of the module (model.py hooks); code that writes parameters in
        place by other means while the module stays in eval mode calls CILRS.weights_changed()."""
        return (self.weights_epoch * 0xD6E8FEB86659FD93) & 0xFFFFFFFFFFFFFFFF or 1

    def poll_versions(self) -> bool:
        """Bump the weights epoch if any parameter / BatchNorm buffer was modified in place through
        torch since the last poll (sum of the tensors' `_version` counters: ~20 us for the 250
        tensors, so only the eval-mode `model(...)` call does it on every use; Predictor and
        run_forward_u8 -- the latency paths -- rely on CILRS.weights_changed()).  Edits that torch
        cannot see (`p.data`, raw pointers) always need weights_changed()."""
        v = sum(map(_VERSION_OF, self._versioned))
        if v == self._version_sum:
            return False
        self._version_sum = v
        self.weights_epoch += 1
        return True

    def _announce_weights(self, pl):
        L.check(L.lib().cilrs_net_set_weights_key(pl.handle, self.weights_key()))

    # ------------------------------------------------------------------------------------------
    def _check_inputs(self, image, speed, command):
        if image.dim() != 4 or image.size(1) != 3:
            raise RuntimeError(f"image must be [B,3,H,W], got {tuple(image.shape)}")
        b = image.size(0)
        if image.dtype != torch.float32 or speed.dtype != torch.float32:
            raise RuntimeError("image and speed must be float32")
        if command.dtype != torch.int64:
            raise RuntimeError("command must be int64 (torch.long), as torch.gather requires")
        if tuple(speed.shape) != (b,) or tuple(command.shape) != (b,):
            raise RuntimeError("speed and command must have shape [B]")
        for t in (image, speed, command):
            if t.device != self.device:
                raise RuntimeError(f"input on {t.device}, model on {self.device}")
        return b

    def run_forward(self, image, speed, command, train, dropout_p, seed):
        """Enqueue the forward; returns (controls, pred_speed, plan)."""
        b = self._check_inputs(image, speed, command)
        pl = self.plan(b, image.size(2), image.size(3))
        speed = speed.contiguous()
        command = command.contiguous()
        controls = torch.empty(b, 3, dtype=torch.float32, device=self.device)
        pred_speed = torch.empty(b, dtype=torch.float32, device=self.device)
        sn, sc, sh, sw = image.stride()
        if train:
            self.weights_epoch += 1           # BN running statistics are about to move
        else:
            self.poll_versions()
        self._announce_weights(pl)
        L.check(L.lib().cilrs_net_forward(
            pl.handle, C.byref(pl.bufs), L.ptr(image), sn, sc, sh, sw,
            L.ptr(speed), L.ptr(command), 1 if train else 0, float(dropout_p), int(seed),
            L.ptr(controls), L.ptr(pred_speed), self._stream()))
        if train:
            pl.generation += 1
        self.last_plan = pl
        return controls, pred_speed, pl

    def check_status(self):
        """Raise if the last forward saw an out-of-range command (one device->host read)."""
        if self.last_plan is not None:
            self.last_plan.check_status()

    def run_forward_u8(self, frames_u8, speed, command, out=None, graph=False, half=False,
                       lane=0, persistent=False):
        """uint8 RGB HWC frames [B,H,W,3] -> eval forward with fused preprocessing.  With
        graph=True the launch sequence is replayed from a cached hipGraph (all tensors must keep
        their addresses; the current stream must not be the default stream).  half=True / "f16"
        runs the trunk in fp16, half="bf16" in bf16 (BatchNorm folded into 16-bit weights, fp32
        accumulation): batched serving.  Calls issued on different HIP streams at the same time
        must use different `lane`s (one workspace each).  persistent=True (one frame, fp32, the
        reference network): the whole forward is ONE launch whose workgroups stay resident and
        meet at in-launch grid barriers (csrc/infer_b1.hip) -- the control-loop path."""
        if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.size(3) != 3:
            raise RuntimeError("frames must be uint8 [B,H,W,3]")
        b = frames_u8.size(0)
        pl = self.plan(b, frames_u8.size(1), frames_u8.size(2), lane)
        frames_u8 = frames_u8.contiguous()
        if out is None:
            controls = torch.empty(b, 3, dtype=torch.float32, device=self.device)
            pred_speed = torch.empty(b, dtype=torch.float32, device=self.device)
        else:
            controls, pred_speed = out
        lib = L.lib()
        if persistent:
            if half or b != 1 or self.variant != 0:
                raise RuntimeError("persistent=True serves one fp32 frame of the reference network")
            fn = lib.cilrs_net_forward_u8_b1
        elif half == "bf16":
            fn = lib.cilrs_net_forward_u8_bf16_graph if graph else lib.cilrs_net_forward_u8_bf16
        elif half:
            fn = lib.cilrs_net_forward_u8_f16_graph if graph else lib.cilrs_net_forward_u8_f16
        else:
            fn = lib.cilrs_net_forward_u8_graph if graph else lib.cilrs_net_forward_u8
        self._announce_weights(pl)
        L.check(fn(pl.handle, C.byref(pl.bufs), L.ptr(frames_u8),
                   L.ptr(speed.contiguous()), L.ptr(command.contiguous()), L.ptr(controls),
                   L.ptr(pred_speed), self._stream()))
        self.last_plan = pl
        return controls, pred_speed

    def run_forward_camera(self, frames_u8, speed, command, height=88, width=200, out=None):
        """Raw camera frames uint8 [B,Hs,Ws,3|4] (device) -> eval forward with the whole of
        preprocess_image fused: bilinear resize to (height, width), /255, normalise."""
        if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.size(3) not in (3, 4):
            raise RuntimeError("camera frames must be uint8 [B,Hs,Ws,3 or 4]")
        b = frames_u8.size(0)
        pl = self.plan(b, height, width)
        frames_u8 = frames_u8.contiguous()
        if out is None:
            controls = torch.empty(b, 3, dtype=torch.float32, device=self.device)
            pred_speed = torch.empty(b, dtype=torch.float32, device=self.device)
        else:
            controls, pred_speed = out
        hs, ws, px = frames_u8.size(1), frames_u8.size(2), frames_u8.size(3)
        self._announce_weights(pl)
        L.check(L.lib().cilrs_net_forward_camera(
            pl.handle, C.byref(pl.bufs), L.ptr(frames_u8), hs, ws, px,
            ws * px, hs * ws * px, L.ptr(speed.contiguous()), L.ptr(command.contiguous()),
            L.ptr(controls), L.ptr(pred_speed), self._stream()))
        self.last_plan = pl
        return controls, pred_speed

    def run_backward(self, pl, dcontrols, dpred_speed, seg_begin=0, seg_end=6, into=None):
        """Writes the parameter gradients of segments [seg_begin, seg_end) into the gradient
        arena, or into `into` (another arena of the same layout)."""
        bufs = pl.bufs
        if into is not None:
            bufs = L.Buffers(self.params.data_ptr(), into.data_ptr(), self.bn.data_ptr(),
                             self.nbt.data_ptr(), pl.workspace.data_ptr())
        L.check(L.lib().cilrs_net_backward(
            pl.handle, C.byref(bufs), L.ptr(dcontrols), L.ptr(dpred_speed), seg_begin, seg_end,
            self._stream()))

    def run_backward_step(self, pl, dcontrols, dpred_speed, exp_avg, exp_avg_sq, lr, betas, eps,
                          weight_decay, step, grad_scale=1.0):
        """loss.backward() + Adam.step() in one library call (cilrs_net_backward_step): each
        segment's parameter range is updated as soon as its gradients are complete, under the
        remaining data gradients.  No gradient clipping on this path."""
        opt = L.AdamArgs(exp_avg.data_ptr(), exp_avg_sq.data_ptr(), float(lr), float(betas[0]),
                         float(betas[1]), float(eps), float(weight_decay), int(step),
                         float(grad_scale))
        self.weights_epoch += 1               # the update writes the parameter arena in place
        L.check(L.lib().cilrs_net_backward_step(
            pl.handle, C.byref(pl.bufs), L.ptr(dcontrols), L.ptr(dpred_speed), C.byref(opt),
            self._stream()))

    def grads_aliased(self) -> bool:
        """True when some parameter's .grad currently lives in the gradient arena (autograd kept
        a view handed out by an earlier backward)."""
        base = self.grads.untyped_storage().data_ptr()
        for p in self.module.parameters():
            g = p.grad
            if g is not None and g.untyped_storage().data_ptr() == base:
                return True
        return False

    # ------------------------------------------------------------------------------------------
    def forward(self, image, speed, command, training, dropout_p, seed):
        needs_graph = training and torch.is_grad_enabled() and any(
            p.requires_grad for p in self.module.parameters())
        if not needs_graph:
            c, s, _ = self.run_forward(image, speed, command, training, dropout_p, seed)
            return c, s
        return _CILRSFunction.apply(self, image, speed, command, float(dropout_p), int(seed),
                                    *self.module.parameters())


class _CILRSFunction(torch.autograd.Function):
    """Composable path: lets ``loss.backward()`` + any torch.optim drive the HIP engine
    (notebook/notebook.ipynb:549-555).  Gradients w.r.t. the image are not produced (the
    reference never asks for them)."""

    @staticmethod
    def forward(ctx, eng, image, speed, command, dropout_p, seed, *params):
        controls, pred_speed, pl = eng.run_forward(image, speed, command, True, dropout_p, seed)
        ctx.eng, ctx.pl, ctx.generation = eng, pl, pl.generation
        ctx.n_params = len(params)
        return controls, pred_speed

    @staticmethod
    def backward(ctx, dcontrols, dpred_speed):
        eng, pl = ctx.eng, ctx.pl
        if pl.generation != ctx.generation:
            raise RuntimeError(
                "CILRS backward: the saved activations of this forward were overwritten by a "
                "later train-mode forward with the same input shape (one graph per shape)")
        b = pl.batch
        if dcontrols is None:
            dcontrols = torch.zeros(b, 3, device=eng.device)
        if dpred_speed is None:
            dpred_speed = torch.zeros(b, device=eng.device)
        # torch semantics: what backward hands to autograd must stay valid for as long as anyone
        # holds it (autograd's own input buffers while several CILRS nodes of one graph are
        # summed, results of torch.autograd.grad, saved p.grad lists).  The kernels write into the
        # engine's gradient arena, which the NEXT backward overwrites -- so by default autograd
        # gets a clone (89.7 MB device copy, ~30 us).  `engine.zero_copy_grads = True` opts into
        # views of the arena for loops that consume the gradients before the next backward
        # (optimizer.step() + zero_grad(set_to_none=True)); when some p.grad still lives in the
        # arena that backward is written to a second arena instead, so accumulation stays correct.
        dst = eng.grads
        if eng.zero_copy_grads and eng.grads_aliased():
            if eng._scratch_grads is None:
                eng._scratch_grads = torch.zeros_like(eng.grads)
            dst = eng._scratch_grads
        eng.run_backward(pl, dcontrols.contiguous().float(), dpred_speed.contiguous().float(),
                         into=None if dst is eng.grads else dst)
        src = dst if eng.zero_copy_grads else dst.clone()
        grads = [_arena_view(src, off, numel, shape)
                 for (_, off, numel, shape) in eng.params_layout]
        return (None, None, None, None, None, None, *grads)
