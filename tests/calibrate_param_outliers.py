"""How many parameters differ from the CPU oracle's by more than 2e-5 after k fused Adam steps
(Adam moves an element whose gradient is within fp32 noise of zero by +-lr): the measured
fractions behind the gates of tests/test_model_gpu.py::_close_params.  Run on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402

import cilrs_oracle as O  # noqa: E402
from cilrs_mi355 import CILRS, CONFIG_A, CONFIG_B, TrainConfig, Trainer  # noqa: E402


def stats(m, orc, tag):
    tot = bad = 0
    worst = (0.0, "", 0, 0)
    small_bad = 0
    maxerr = 0.0
    for (n, a), (_, b) in zip(m.named_parameters(), orc.named_parameters()):
        e = (a.detach().cpu() - b.detach()).abs()
        k = int((e > 2e-5).sum())
        maxerr = max(maxerr, float(e.max()))
        tot += e.numel()
        bad += k
        if e.numel() >= 4096:
            f = k / e.numel()
            if f > worst[0]:
                worst = (f, n, k, e.numel())
        else:
            small_bad = max(small_bad, k)
    print(f"{tag}: {bad}/{tot} = {bad / tot:.3e} beyond 2e-5; worst big tensor {worst[1]} "
          f"{worst[2]}/{worst[3]} = {worst[0]:.3e}; worst count in a tensor < 4096 elems: "
          f"{small_bad}; max |diff| {maxerr:.3e}", flush=True)


def run(B, cfg, ocfg, steps, seeds, tag, hw=(88, 200)):
    m = CILRS(4, 0.0)
    m.load_state_dict(O.portable_state_dict(m.state_dict(), 0))
    m = m.cuda()
    tr = Trainer(m, cfg)
    orc = O.build_oracle(0)
    opt = O.make_optimizer(orc, ocfg)
    for s in range(steps):
        imgs, spds, cmds, tgts = O.synthetic_batch(B, seed=seeds[s], h=hw[0], w=hw[1])[:4]
        tr.train_step(imgs.cuda(), spds.cuda(), cmds.cuda(), tgts.cuda())
        O.train_step(orc, opt, ocfg, imgs, spds, cmds, tgts)
        torch.cuda.synchronize()
        stats(m, orc, f"{tag} B={B} step {s + 1}")


if __name__ == "__main__":
    from cilrs_mi355.hostinfo import describe, usable_cores
    torch.set_num_threads(usable_cores())
    print(describe(), flush=True)
    cb = TrainConfig(**{**CONFIG_B.__dict__, "dropout": 0.0})
    run(4, CONFIG_A, O.CONFIG_A, 1, [1], "cfgA")
    run(3, CONFIG_A, O.CONFIG_A, 1, [5], "cfgA 96x160", hw=(96, 160))
    run(5, CONFIG_A, O.CONFIG_A, 1, [6], "cfgA 64x64", hw=(64, 64))
    run(128, CONFIG_A, O.CONFIG_A, 2, [2024, 2025], "cfgA")
    run(128, cb, O.CONFIG_B, 1, [2024], "cfgB")
