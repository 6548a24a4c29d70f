#!/usr/bin/env python3
"""Generate tests/golden/* -- runs ONLY in the build container (needs /root/reference).

TEST INFRASTRUCTURE.  Reads the reference's two source files *as text*, ast-extracts the hot-path
definitions -- ``CILRS`` (model/autonomous_drive.py:361-399 and notebook/notebook.ipynb:440-477),
``CILRSLoss`` (nb:504-527), ``train_one_epoch`` (nb:541-561), ``validate`` (nb:563-585) -- and
execs them with the real ``torch`` and with ``oracle.cilrs_oracle.resnet34_trunk`` standing in for
the absent ``torchvision.models.resnet34``.  It then

  1. checks the known answers (22,421,453 parameters, 250 state_dict entries),
  2. checks that the oracle restatement is BIT-IDENTICAL to the extracted reference code on the
     same weights and inputs (forward eval/train, loss B, three optimiser steps), and
  3. writes small fixtures (outputs, checksums, loss dicts) -- data only, no reference source.

Weights come from oracle.cilrs_oracle.portable_state_dict (integer hash; never committed).
"""
import ast
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.optim as optim

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import cilrs_oracle as O  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _extract(src: str, names):
    tree = ast.parse(src)
    got = {}
    for node in tree.body:
        if isinstance(node, (ast.ClassDef, ast.FunctionDef)) and node.name in names:
            got[node.name] = ast.get_source_segment(src, node)
    missing = set(names) - set(got)
    assert not missing, missing
    return got


def load_reference_defs():
    models = types.SimpleNamespace(resnet34=O.resnet34_trunk,
                                   ResNet34_Weights=types.SimpleNamespace(DEFAULT=None))
    drive_src = open(os.path.join(REF, "model", "autonomous_drive.py")).read()
    ns_drive = dict(torch=torch, nn=nn, models=models, np=np)
    exec(_extract(drive_src, ["CILRS"])["CILRS"], ns_drive)

    nb = json.load(open(os.path.join(REF, "notebook", "notebook.ipynb")))
    cell = max((c for c in nb["cells"] if c["cell_type"] == "code"),
               key=lambda c: len(c["source"]))
    lines = [ln for ln in "".join(cell["source"]).split("\n")
             if not ln.lstrip().startswith(("!", "%"))]
    nb_src = "\n".join(lines)
    ns_nb = dict(torch=torch, nn=nn, optim=optim, models=models, np=np)
    for name, seg in _extract(nb_src, ["CILRS", "CILRSLoss", "train_one_epoch",
                                       "validate"]).items():
        exec(seg, ns_nb)
    return ns_drive["CILRS"], ns_nb


def tsum(t):
    return float(t.detach().double().sum())


def param_checks(model):
    out = {}
    for n, p in model.named_parameters():
        f = p.detach().flatten()
        idx = [0, f.numel() // 3, (2 * f.numel()) // 3, f.numel() - 1]
        out[n] = dict(sum=tsum(p), l2=float(p.detach().double().norm()),
                      samples=[float(f[i]) for i in idx])
    return out


def grad_checks(model):
    out = {}
    for n, p in model.named_parameters():
        f = p.grad.detach().flatten()
        idx = [0, f.numel() // 3, (2 * f.numel()) // 3, f.numel() - 1]
        out[n] = dict(l2=float(p.grad.double().norm()), sum=tsum(p.grad),
                      samples=[float(f[i]) for i in idx])
    return out


def buffer_checks(model):
    return {n: dict(sum=tsum(b), l2=float(b.double().norm()))
            for n, b in model.named_buffers()}


def same(a, b):
    return a.shape == b.shape and bool((a == b).all())


def evaluation_schema_fixture():
    """Key structure + published identities of the reference's own evaluation report (a data file,
    /root/reference/evaluation_report.json:1-73): nested key names / value types, the (MSE, RMSE)
    pairs and the per-command sample counts."""
    r = json.load(open(os.path.join(REF, "evaluation_report.json")))

    def keys(d):
        return {k: (keys(v) if isinstance(v, dict) else type(v).__name__) for k, v in d.items()}
    g = {"schema": keys(r), "val_samples": r["val_samples"],
         "mse_rmse": {k: [v["MSE"], v["RMSE"]] for k, v in r["overall_metrics"].items()},
         "per_command_n": {k: v["n"] for k, v in r["per_command_metrics"].items()},
         "source": "/root/reference/evaluation_report.json:1-73 (key structure and the published "
                   "MSE/RMSE, n values)"}
    json.dump(g, open(os.path.join(OUT, "evaluation_report_schema.json"), "w"), indent=1)


def camera_fixture(orc):
    """Whole preprocess_image incl. the resize (autonomous_drive.py:868-872, 897-902): a 600x800
    4-byte-per-pixel camera frame -> resized bytes (digest + samples) -> controls.  cv2 is absent,
    so the resize is the oracle's restatement of OpenCV's 8-bit INTER_LINEAR (parity unpinned)."""
    import hashlib
    cam = np.floor(O._hash_u01(4321, 9, 600 * 800 * 4) * 256).astype(np.uint8).reshape(600, 800, 4)
    small = O.resize_bilinear_u8(np.ascontiguousarray(cam[:, :, :3]))
    with torch.no_grad():
        x = O.preprocess_camera(cam)
        pc, ps = orc.eval()(x, torch.tensor([min(40.0 / 90.0, 1.0)]), torch.tensor([2]))
    json.dump(dict(frame_seed=4321, frame_stream=9, shape=[600, 800, 4],
                   resized_sha256=hashlib.sha256(small.tobytes()).hexdigest(),
                   resized_sum=int(small.astype(np.int64).sum()),
                   resized_row0=small[0, :8].reshape(-1).tolist(),
                   speed_kmh=40.0, command=2,
                   out=[pc[0, 0].item(), pc[0, 1].item(), pc[0, 2].item(), ps[0].item() * 90.0]),
              open(os.path.join(OUT, "camera_pipeline.json"), "w"))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    RefDrive, nb = load_reference_defs()
    RefTrain, RefLoss = nb["CILRS"], nb["CILRSLoss"]

    # ---- known answers ---------------------------------------------------------------
    ref = RefDrive(num_commands=4, dropout=0.0)
    n_params = sum(p.numel() for p in ref.parameters())
    assert n_params == O.N_PARAMS, n_params
    sd = ref.state_dict()
    assert len(sd) == 250 and len(list(ref.parameters())) == 142
    orc = O.CILRSOracle(4, 0.0)
    assert list(orc.state_dict().keys()) == list(sd.keys())
    keys = [dict(name=k, shape=list(v.shape), dtype=str(v.dtype).replace("torch.", ""))
            for k, v in sd.items()]
    json.dump(dict(n_params=n_params, n_entries=len(sd), n_param_tensors=142, entries=keys),
              open(os.path.join(OUT, "state_dict_keys.json"), "w"), indent=0)

    psd = O.portable_state_dict(sd, seed=0)
    ref.load_state_dict(psd, strict=True)
    orc.load_state_dict(psd, strict=True)
    wsum = {k: tsum(v) for k, v in psd.items() if v.dtype.is_floating_point}
    json.dump(dict(seed=0, total=float(sum(wsum.values())),
                   first=wsum["visual_encoder.0.weight"],
                   last=wsum["speed_predictor.5.bias"]),
              open(os.path.join(OUT, "portable_weights_check.json"), "w"))

    # ---- forward, eval mode, B=4 -----------------------------------------------------
    img, spd, cmd, tgt, u8 = O.synthetic_batch(4, seed=1)
    cmd = torch.tensor([0, 1, 2, 3])
    ref.eval(); orc.eval()
    with torch.no_grad():
        rc, rs = ref(img, spd, cmd)
        oc, os_ = orc(img, spd, cmd)
    assert same(rc, oc) and same(rs, os_), "oracle != reference (eval forward)"
    np.savez(os.path.join(OUT, "forward_eval_b4.npz"), seed=1, batch=4, command=cmd.numpy(),
             image_sum=tsum(img), speed=spd.numpy(), controls=rc.numpy(), pred_speed=rs.numpy())

    # ---- forward, train mode (batch stats), B=8, dropout 0 ----------------------------
    img8, spd8, cmd8, tgt8, _ = O.synthetic_batch(8, seed=2)
    ref_t = RefTrain(num_commands=4, dropout=0.0); ref_t.load_state_dict(psd)
    orc_t = O.CILRSOracle(4, 0.0); orc_t.load_state_dict(psd)
    ref_t.train(); orc_t.train()
    rc, rs = ref_t(img8, spd8, cmd8)
    oc, os_ = orc_t(img8, spd8, cmd8)
    assert same(rc, oc) and same(rs, os_), "oracle != reference (train forward)"
    np.savez(os.path.join(OUT, "forward_train_b8.npz"), seed=2, batch=8,
             command=cmd8.numpy(), controls=rc.detach().numpy(), pred_speed=rs.detach().numpy(),
             bn0_running_mean=ref_t.state_dict()["visual_encoder.1.running_mean"].numpy(),
             bn0_running_var=ref_t.state_dict()["visual_encoder.1.running_var"].numpy(),
             last_running_mean=ref_t.state_dict()["visual_encoder.7.2.bn2.running_mean"].numpy(),
             last_running_var=ref_t.state_dict()["visual_encoder.7.2.bn2.running_var"].numpy())
    json.dump(buffer_checks(ref_t), open(os.path.join(OUT, "forward_train_b8_buffers.json"), "w"))

    # ---- three optimiser steps, Config B (the executed notebook code) ------------------
    def run_reference_steps(cfg, steps=3):
        m = RefTrain(num_commands=4, dropout=0.0); m.load_state_dict(psd)
        w = cfg.loss_weights
        crit = RefLoss(steer_w=w[0], throttle_w=w[1], brake_w=w[2], speed_w=w[3])
        opt = optim.Adam(m.parameters(), lr=cfg.lr, weight_decay=cfg.weight_decay)
        rec = []
        for s in range(steps):
            batch = O.synthetic_batch(8, seed=10 + s)[:4]
            imgs, spds, cmds, tgts = batch
            ld = nb["train_one_epoch"](m, [(imgs, spds, cmds, tgts)], crit, opt,
                                       torch.device("cpu"), cfg.grad_clip)
            rec.append(dict(loss=ld, grads=grad_checks(m) if s == 0 else None,
                            params=param_checks(m) if s in (0, steps - 1) else None))
        return m, opt, rec

    def run_oracle_steps(cfg, steps=3):
        m = O.CILRSOracle(4, 0.0); m.load_state_dict(psd)
        opt = O.make_optimizer(m, cfg)
        rec = []
        for s in range(steps):
            imgs, spds, cmds, tgts = O.synthetic_batch(8, seed=10 + s)[:4]
            ld, gn = O.train_step(m, opt, cfg, imgs, spds, cmds, tgts)
            rec.append(dict(loss=ld, gnorm=gn, grads=grad_checks(m) if s == 0 else None,
                            params=param_checks(m) if s in (0, steps - 1) else None))
        return m, opt, rec

    mr, optr, rec_r = run_reference_steps(O.CONFIG_B)
    mo, opto, rec_o = run_oracle_steps(O.CONFIG_B)
    for (n, a), (_, b) in zip(mr.named_parameters(), mo.named_parameters()):
        assert same(a, b), f"oracle != reference after 3 Config-B steps: {n}"
    for a, b in zip(rec_r, rec_o):
        assert a["loss"] == b["loss"], (a["loss"], b["loss"])
    st = optr.state_dict()["state"]
    json.dump(dict(config="B", batch=8, seeds=[10, 11, 12], steps=rec_o,
                   buffers=buffer_checks(mo),
                   adam=dict(step=float(st[0]["step"]), exp_avg0_sum=tsum(st[0]["exp_avg"]),
                             exp_avg_sq0_sum=tsum(st[0]["exp_avg_sq"]),
                             exp_avg141_sum=tsum(st[141]["exp_avg"]))),
              open(os.path.join(OUT, "step_cfgB_b8.json"), "w"))

    # Config A has no loss code in the reference (documented only): the oracle's restatement
    # is the definition; fixtures record it so later rounds cannot drift.
    mo, opto, rec_o = run_oracle_steps(O.CONFIG_A)
    json.dump(dict(config="A", batch=8, seeds=[10, 11, 12], steps=rec_o,
                   buffers=buffer_checks(mo)),
              open(os.path.join(OUT, "step_cfgA_b8.json"), "w"))

    # ---- validate() aggregation (nb:563-585) -----------------------------------------
    w = O.CONFIG_B.loss_weights
    crit = RefLoss(steer_w=w[0], throttle_w=w[1], brake_w=w[2], speed_w=w[3])
    batches = [O.synthetic_batch(4, seed=20 + i)[:4] for i in range(2)]
    vl_r, cmd_r = nb["validate"](ref, batches, crit, torch.device("cpu"))
    vl_o, cmd_o = O.validate_batches(orc, O.CONFIG_B, batches)
    assert vl_r == vl_o, (vl_r, vl_o)
    for k in cmd_r:
        a, b = float(cmd_r[k]), cmd_o[k]
        assert (np.isnan(a) and np.isnan(b)) or a == b
    json.dump(dict(seeds=[20, 21], batch=4, losses=vl_o,
                   cmd_steer={k: (None if np.isnan(v) else v) for k, v in cmd_o.items()}),
              open(os.path.join(OUT, "validate_cfgB.json"), "w"))

    # ---- inference adapter (autonomous_drive.py:897-920, no cv2.resize) -----------------
    frame = np.floor(O._hash_u01(1234, 7, O.IMG_H * O.IMG_W * 3) * 256).astype(np.uint8)
    frame = frame.reshape(O.IMG_H, O.IMG_W, 3)
    cases = []
    for kmh, c in [(25.0, 0), (0.0, 1), (120.0, 2), (61.5, 3)]:
        cases.append(dict(speed_kmh=kmh, command=c,
                          out=list(O.predict_controls(orc, frame, kmh, c))))
    # cross-check against the reference class driven exactly as predict_controls does
    with torch.no_grad():
        x = O.preprocess_frame(frame)
        pc, ps = ref(x, torch.tensor([min(25.0 / 90.0, 1.0)]), torch.tensor([0]))
    assert [pc[0, 0].item(), pc[0, 1].item(), pc[0, 2].item(), ps[0].item() * 90.0] == \
        cases[0]["out"]
    json.dump(dict(frame_seed=1234, frame_stream=7, frame_sum=int(frame.astype(np.int64).sum()),
                   cases=cases), open(os.path.join(OUT, "infer_pipeline.json"), "w"))
    camera_fixture(orc)
    evaluation_schema_fixture()
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print(f"  {f}: {os.path.getsize(os.path.join(OUT, f))} B")


if __name__ == "__main__":
    main()
