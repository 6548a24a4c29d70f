"""Data-parallel train step with two real ranks on one MI355X.

RCCL refuses two ranks on the same device, so the two processes talk through gloo (device tensors
staged by the backend); everything else -- the HIP plan, the bucketed all-reduce issued between
backward segments, the 1/world gradient scale folded into Adam -- is the production path that
`bench.py --gpus N` runs over RCCL.  Expected values: the oracle's per-shard gradients (per-replica
BatchNorm statistics) averaged, then one Adam step.
"""
import os

import pytest
import torch

import cilrs_oracle as O

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q, out_dir, cfg_name, variant=0):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from cilrs_mi355 import CILRS, CILRSResNet50, CONFIG_A, CONFIG_B, TrainConfig, Trainer
        from cilrs_mi355.parallel import broadcast_parameters
        torch.cuda.set_device(0)
        m = (CILRSResNet50 if variant == 1 else CILRS)(4, dropout=0.0)
        # rank 1 starts from different weights: the broadcast must overwrite them
        m.load_state_dict(O.portable_state_dict(m.state_dict(), 0 if rank == 0 else 5), strict=True)
        m = m.cuda()
        cfg = CONFIG_A if cfg_name == "A" else TrainConfig(**{**CONFIG_B.__dict__, "dropout": 0.0})
        tr = Trainer(m, cfg, process_group=dist.group.WORLD)
        broadcast_parameters(tr.eng, dist.group.WORLD)
        losses = []
        extra = {}
        for step in range(2):
            imgs, spds, cmds, tgts = O.synthetic_batch(4, seed=60 + 10 * step + rank)[:4]
            tr.train_step(imgs.cuda(), spds.cuda(), cmds.cuda(), tgts.cuda())
            losses.append(tr.losses()["total"])
            if step == 0:
                # the gradient the optimiser consumed: the all-reduced arena times the scale the
                # trainer folds in (1/world) -- Adam is scale-invariant, so the parameters alone
                # cannot tell a summed gradient from an averaged one
                extra["grads"] = {n: (g.detach().cpu() * tr.arena_grad_scale).contiguous()
                                  for (n, _, _, _), g in zip(tr.eng.params_layout,
                                                             tr.eng.grad_views)}
                extra["gnorm"] = tr.grad_norm() if cfg.grad_clip > 0 else None
        torch.cuda.synchronize()
        path = os.path.join(out_dir, f"rank{rank}.pt")
        torch.save({k: v.detach().cpu() for k, v in m.state_dict().items()}, path)
        torch.save(extra, os.path.join(out_dir, f"rank{rank}_extra.pt"))
        q.put((rank, None, losses, path))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:                                   # surface the failure in the parent
        import traceback
        q.put((rank, traceback.format_exc() + str(e), None, None))


@pytest.mark.parametrize("cfg_name,variant", [("A", 0), ("B", 0), ("B", 1)])
def test_two_rank_train_step_matches_oracle_dp_semantics(tmp_path, cfg_name, variant):
    """variant 1: the ResNet-50 network through the same DP path (its own gradient segments)."""
    import torch.multiprocessing as mp
    if variant == 1:
        import resnet50_oracle as R
        build = R.build_oracle50
    else:
        build = O.build_oracle
    gtol = 3e-2 if variant == 1 else 1e-2     # per-tensor HIP-vs-fp32-oracle gradient distance
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000) + (7 if cfg_name == "B" else 0) + 13 * variant
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, str(tmp_path), cfg_name, variant))
             for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(120)
    for r in res:
        assert r[1] is None, r[1]
    sd0, sd1 = (torch.load(res[r][3], weights_only=True) for r in range(2))
    # replicas stay bit-identical in their parameters; BN buffers are per-replica
    for k in sd0:
        if "running_" in k or "num_batches" in k:
            continue
        assert torch.equal(sd0[k], sd1[k]), k
    assert not torch.equal(sd0["visual_encoder.1.running_mean"], sd1["visual_encoder.1.running_mean"])

    # oracle: per-shard gradients averaged, [clipped,] one Adam step -- twice
    ocfg = O.CONFIG_A if cfg_name == "A" else O.CONFIG_B
    lr = ocfg.lr
    reps = [build(0) for _ in range(2)]                      # per-rank BN buffers
    opt = O.make_optimizer(reps[0], ocfg)
    for step in range(2):
        grads, want_losses = [], []
        for rank in range(2):
            m = reps[rank].train()
            m.zero_grad()
            imgs, spds, cmds, tgts = O.synthetic_batch(4, seed=60 + 10 * step + rank)[:4]
            pc, ps = m(imgs, spds, cmds)
            loss, _ = O.compute_loss(ocfg, pc, tgts, ps, spds)
            loss.backward()
            want_losses.append(float(loss.detach()))
            grads.append([p.grad.clone() for p in m.parameters()])
        for p, a, b in zip(reps[0].parameters(), *grads):
            p.grad = (a + b) / 2
        if step == 0:
            # both ranks hold the same AVERAGED gradient (not the sum, not their own shard's)
            ex = [torch.load(os.path.join(str(tmp_path), f"rank{r}_extra.pt"), weights_only=True)
                  for r in range(2)]
            avg_sq = own_sq = 0.0
            for (n, p), a in zip(reps[0].named_parameters(), grads[0]):
                want = p.grad
                for r in range(2):
                    got = ex[r]["grads"][n]
                    nrm = max(float(want.norm()), 1e-12)
                    assert float((got - want).norm()) <= gtol * nrm, (n, r)
                assert torch.equal(ex[0]["grads"][n], ex[1]["grads"][n]), n
                avg_sq += float((want.double() ** 2).sum())
                own_sq += float((a.double() ** 2).sum())
            if ocfg.grad_clip > 0:
                # clip_grad_norm_ saw the averaged gradient: its norm is neither the sum's (2x)
                # nor one shard's
                for r in range(2):
                    assert abs(ex[r]["gnorm"] - avg_sq ** 0.5) <= 2e-3 * avg_sq ** 0.5
                assert abs(own_sq ** 0.5 - avg_sq ** 0.5) > 1e-2 * avg_sq ** 0.5
        if ocfg.grad_clip > 0:         # clip acts on the AVERAGED gradient (nb:553-554)
            torch.nn.utils.clip_grad_norm_(reps[0].parameters(), ocfg.grad_clip)
        opt.step()
        with torch.no_grad():
            for p0, p1 in zip(reps[0].parameters(), reps[1].parameters()):
                p1.copy_(p0)
        tol = 1e-4 if step == 0 else 1e-3
        for rank in range(2):
            assert abs(res[rank][2][step] - want_losses[rank]) <= tol * max(1.0, want_losses[rank])
    for (n, p) in reps[0].named_parameters():
        err = (sd0[n] - p.detach()).abs()
        assert float(err.max()) <= 2.2 * lr * 2 + 1e-6, n       # Adam's lr*sign(g) ambiguity
        # (two free-running steps in, only the hard bound is a check -- tests/test_model_gpu.py
        #  _close_params; the gradient itself is pinned above)
    # rank 1's BN statistics followed ITS shard
    want_rm1 = reps[1].state_dict()["visual_encoder.1.running_mean"]
    # (step 2 starts from parameters that differ by Adam's lr*sign(g) ambiguity: 1e-4, not 1e-5)
    assert (sd1["visual_encoder.1.running_mean"] - want_rm1).abs().max() <= 1e-4
    assert (sd0["visual_encoder.1.running_mean"] - want_rm1).abs().max() > 1e-3


def _fit_worker(rank, world, port, q, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from cilrs_mi355 import CILRS, CONFIG_A, Trainer
        from cilrs_mi355.loop import fit
        from cilrs_mi355.parallel import broadcast_parameters
        torch.cuda.set_device(0)
        m = CILRS(4, dropout=0.0)
        m.load_state_dict(O.portable_state_dict(m.state_dict(), 0), strict=True)
        m = m.cuda()
        tr = Trainer(m, CONFIG_A, process_group=dist.group.WORLD)
        broadcast_parameters(tr.eng, dist.group.WORLD)

        def batches(seed0, n):
            def gen():
                for i in range(n):
                    b = O.synthetic_batch(4, seed=seed0 + 2 * i + rank)[:4]
                    yield [t.cuda() for t in b]
            return gen
        # The two ranks' validation shards differ a lot (rank 1's targets are shifted), so a
        # rank-local validation loss would improve on one rank and not on the other: with
        # patience 1 one rank would stop alone and the other hang in the next all-reduce.
        def val_batches():
            for i in range(2):
                imgs, spds, cmds, tgts = O.synthetic_batch(4, seed=500 + 2 * i + rank)[:4]
                if rank == 1:
                    tgts = tgts + 3.0
                yield imgs.cuda(), spds.cuda(), cmds.cuda(), tgts.cuda()
        logs = []
        res = fit(tr, batches(100, 2), val_batches, epochs=4, patience=1, out_dir=out_dir,
                  log=logs.append)
        torch.cuda.synchronize()
        q.put((rank, None, [r["val_total"] for r in res["history"]], res["best_epoch"],
               len(res["history"])))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, traceback.format_exc() + str(e), None, None, None))


def test_two_rank_fit_takes_the_same_decisions_on_every_rank(tmp_path):
    """Epoch loop under data parallelism: validation sums are all-reduced, so both
    ranks see the SAME metrics, stop at the same epoch (no rank left hanging in an all-reduce),
    and only rank 0 writes checkpoint_best.pth / checkpoint_latest.pth / training_history.csv."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 2000) + 101
    procs = [ctx.Process(target=_fit_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in res:
        assert r[1] is None, r[1]
    assert res[0][2] == res[1][2]                       # identical validation history
    assert res[0][3] == res[1][3] and res[0][4] == res[1][4]
    files = sorted(os.listdir(tmp_path))
    assert files == ["checkpoint_best.pth", "checkpoint_latest.pth", "training_history.csv"], files
    # the metric is the mean over BOTH shards: rank 1's shifted targets are in it
    assert res[0][2][0] > 1.0


def test_two_rank_train_step_without_side_stream_overlap(tmp_path, monkeypatch):
    """The same two-rank step with CILRS_OVERLAP=0 (weight gradients on the caller's stream): the
    segment join (net.hip gbuf_join_all) is what orders the side-stream weight gradients in front
    of each bucket's all-reduce -- with and without the side stream the replicas end bit-identical
    and equal to each other's result."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = {}
    for overlap in ("1", "0"):
        monkeypatch.setenv("CILRS_OVERLAP", overlap)
        d = tmp_path / f"ov{overlap}"
        d.mkdir()
        q = ctx.Queue()
        port = 29600 + (os.getpid() % 2000) + 131 + int(overlap)
        procs = [ctx.Process(target=_worker, args=(r, 2, port, q, str(d), "A", 0)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
        for p in procs:
            p.join(120)
        for r in res:
            assert r[1] is None, r[1]
        sd = [torch.load(res[r][3], weights_only=True) for r in range(2)]
        for k in sd[0]:
            if "running_" not in k and "num_batches" not in k:
                assert torch.equal(sd[0][k], sd[1][k]), (overlap, k)
        out[overlap] = (sd[0], res[0][2])
    # same arithmetic either way (the side stream changes the schedule, not the sums)
    for k in out["1"][0]:
        assert torch.equal(out["1"][0][k], out["0"][0][k]), k
    assert out["1"][1] == out["0"][1]
