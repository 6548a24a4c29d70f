#!/usr/bin/env python3
"""Benchmark of the CILRS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch: forward + loss + backward +
[gradient all-reduce] + Adam for BASELINE.json configs[1] ("CILRS training batch=128, 200x88 RGB,
ResNet-34, Adam lr=2e-4, fp32"; Config A of SURVEY.md).  Inputs are resident in HBM before the
timed region.  N > 1 runs one rank per GPU (weak scaling: 128 frames per GPU, RCCL all-reduce of
the gradient arena overlapped with backward): either the driver starts the ranks with
torch.distributed.run (RANK / WORLD_SIZE in the environment), or -- `python bench.py --gpus N`
on its own -- this process starts them itself as FRESH child processes before it has made any GPU
call, relays rank 0's JSON line and exits non-zero if any rank fails (an RCCL error included).

Prints ONE JSON line (rank 0).  Besides the driver's contract it carries
  roofline      dominant kernel family (implicit-GEMM conv), hipEvent-timed on the launch stream
  cpu_baseline  the CPU oracle (torch fp32, the reference's own arithmetic) on this host's cores,
                run on the SAME batch and the SAME initial weights as the GPU: its outputs and
                first-step loss are the parity check of this very run (parity_max_abs_err)
  allreduce     per-bucket timing of the gradient all-reduce (N > 1, or --force-dp)
  infer_ms      single-frame inference latency (predict_controls path), median
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "cilrs-autonomous-driving-carla_amd"))
# RCCL between processes needs dmabuf IPC on this driver (see the environment notes); keep it set
# for every rank even when the launcher's environment lost it
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

TRAIN_GFLOP_PER_FRAME = 8.39      # BASELINE.md section 2 (fwd 2.798 x 3)
PEAK_F32_MATRIX_TFLOPS = 157.3    # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0
PEAK_F16_TFLOPS = 2500.0            # dense fp16 / bf16 matrix peak (not the 2:1-sparsity figure)


def synthetic_batch(batch, seed, device):
    """ImageNet-normalised U{0..255} frames, speed ~ U[0,1), command ~
    U{0..3}, targets steer ~ U[-1,1], throttle/brake ~ U[0,1]."""
    g = torch.Generator().manual_seed(seed)
    u8 = torch.randint(0, 256, (batch, 88, 200, 3), generator=g, dtype=torch.uint8)
    img = u8.float().div(255.0).permute(0, 3, 1, 2)
    mean = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    img = ((img - mean) / std).contiguous()
    speed = torch.rand(batch, generator=g)
    cmd = torch.randint(0, 4, (batch,), generator=g)
    tgt = torch.rand(batch, 3, generator=g)
    tgt[:, 0] = tgt[:, 0] * 2 - 1
    return [t.to(device) for t in (img, speed, cmd, tgt)], u8


def cpu_baseline(batch_cpu, state0, gpu_out, gpu_loss1, steps=3, forwards=50):
    """The oracle (oracle/cilrs_oracle.py) on this host's cores, fed the batch the GPU timed and
    the weights the GPU started from: (1) a train-mode forward and the first
    train step reproduce what the GPU did before its warm-up -- outputs and loss terms must agree
    within 1e-4 (BASELINE.json north_star), which is the parity check of THIS run; (2) `steps`
    further Config-A train steps are timed; (3) `forwards` single-frame eval forwards are timed
    after 5 warm-ups."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cilrs_oracle as O
    from cilrs_mi355 import hostinfo
    # every core this job may keep busy: the affinity mask of a shared GPU box lists the whole
    # machine, the job's CPU share is 16 cores per GPU (hostinfo.py) -- more threads than that
    # oversubscribe the share and run an order of magnitude slower
    cores = hostinfo.usable_cores()
    torch.set_num_threads(cores)
    imgs, spds, cmds, tgts = batch_cpu
    batch = imgs.size(0)
    m = O.CILRSOracle(4, 0.0)
    m.load_state_dict(state0, strict=True)
    opt = O.make_optimizer(m, O.CONFIG_A)
    m.train()
    with torch.no_grad():
        oc, osp = m(imgs, spds, cmds)
    err_out = max(float((oc - gpu_out[0]).abs().max()), float((osp - gpu_out[1]).abs().max()))
    ld, _ = O.train_step(m, opt, O.CONFIG_A, imgs, spds, cmds, tgts)      # also the warm-up
    err_loss = max(abs(ld[k] - gpu_loss1[k]) / max(1.0, abs(ld[k])) for k in ld)
    t0 = time.perf_counter()
    for _ in range(steps):
        O.train_step(m, opt, O.CONFIG_A, imgs, spds, cmds, tgts)
    dt = time.perf_counter() - t0
    m.eval()
    with torch.no_grad():
        for _ in range(5):
            m(imgs[:1], spds[:1], cmds[:1])
        t1 = time.perf_counter()
        for _ in range(forwards):
            m(imgs[:1], spds[:1], cmds[:1])
        infer_ms = (time.perf_counter() - t1) / forwards * 1e3
    base = dict(value=round(batch * steps / dt, 2), unit="frames/s", cores=cores, kind="port",
                sample=f"{steps} Config-A train steps at B={batch} on the GPU run's own batch and "
                       f"initial weights (after the parity step as warm-up), "
                       f"torch.set_num_threads({cores}) [{hostinfo.describe()}]; "
                       f"infer = mean of {forwards} B=1 eval forwards after 5 warm-ups",
                infer_ms=round(infer_ms, 3))
    return base, err_out, err_loss


def free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N fresh ranks with
    torch.distributed.run BEFORE this process touches the GPU (it never will), forward rank 0's
    JSON line to stdout and everything else to stderr, return the children's exit status."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "2")
    print(f"[bench] starting {n} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line_out = None
    for line in proc.stdout:
        t = line.strip()
        if line_out is None and t.startswith("{") and '"metric"' in t:
            line_out = t
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if rc != 0:
        print(f"[bench] a rank failed (exit status {rc})", file=sys.stderr, flush=True)
        return rc
    if line_out is None:
        print("[bench] the ranks exited cleanly but rank 0 printed no result line",
              file=sys.stderr, flush=True)
        return 1
    print(line_out, flush=True)
    return 0


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def rehearse(args, rank, world, real_stdout):
    """--rehearse: the multi-rank plumbing of this file on CPU ranks (gloo) -- the launcher, the
    rendezvous, the bucketed gradient all-reduce over a stand-in gradient arena, the barrier /
    max-over-ranks timing and the one-line JSON relay -- WITHOUT the HIP engine.  It measures
    nothing: tests/test_host.py uses it to cover `bench.py --gpus 2` on a box without GPUs."""
    from cilrs_mi355 import _lib as L
    from cilrs_mi355.engine import segment_ranges
    from cilrs_mi355.parallel import BucketedAllReduce, bucket_plan
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = L.lib().cilrs_param_arena_floats()
    flat = torch.full((n,), float(rank + 1))
    red = BucketedAllReduce(flat, dist.group.WORLD, buckets=bucket_plan(segment_ranges()))

    class NoEngine:
        def run_backward(self, *a):
            pass
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        flat.fill_(float(rank + 1))
        red.backward_and_reduce(NoEngine(), None, None, None)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    want = float(world * (world + 1) // 2)
    ok = bool((flat == want).all())
    seen = dist.get_world_size()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(f"rank {rank}: all-reduce result wrong")
    if rank == 0:
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps({"metric": "REHEARSAL ONLY (gloo, no GPU work, nothing measured)",
                          "value": None, "unit": "frames/s", "n_gpus": world,
                          "n_ranks_seen": seen, "steps": args.steps, "warmup": args.warmup,
                          "rehearsal": True, "allreduce_sum_ok": ok,
                          "wall_s": round(float(t.item()), 4)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=128, help="frames per GPU")
    ap.add_argument("--config", default="A", choices=["A", "B"])
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-infer", action="store_true")
    ap.add_argument("--no-loader", action="store_true",
                    help="skip the input-pipeline leg (JPEG decode pool -> augmentation -> train step)")
    ap.add_argument("--fused-step", action="store_true",
                    help="cilrs_net_backward_step (per-segment Adam on the weight-gradient stream) instead of "
                         "backward + one Adam launch: same numbers, 0.3 %% slower (A/B)")
    ap.add_argument("--two-call-step", action="store_true", help="(the default since round 4; kept for scripts)")
    ap.add_argument("--force-dp", action="store_true",
                    help="use the bucketed all-reduce path even with one rank (rehearsal)")
    ap.add_argument("--rehearse", action="store_true",
                    help="CPU ranks over gloo: exercises the launcher / all-reduce / JSON relay "
                         "only, measures nothing (tests)")
    args = ap.parse_args()

    # ---- `python bench.py --gpus N` on its own: become the launcher.  Nothing above or below
    # this point has touched the GPU in this process (importing torch does not), and the ranks
    # are fresh child processes -- never an exec of a process that initialised HIP.
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # stdout carries exactly ONE JSON line: everything else that writes to file descriptor 1 (this
    # pool exports NCCL_DEBUG=VERSION and RCCL prints its banner there with printf) goes to stderr
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        # N ranks share the node's CPUs: keep every rank's host thread pool to its share
        from cilrs_mi355 import hostinfo as _hi
        torch.set_num_threads(max(1, _hi.usable_cores() // world))
    if args.rehearse:
        return rehearse(args, rank, world, real_stdout)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    pg = None
    if world > 1 or args.force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # an RCCL failure must end the job with a non-zero status, not hang the other ranks
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")
        # (no device_id=: eager communicator binding makes every async all-reduce cost the
        # train step +1.6 ms on this stack -- tools/dp_ab.py, 13.6 vs 12.0 ms at world size 1)
        dist.init_process_group("nccl", rank=rank, world_size=world)
        pg = dist.group.WORLD

    from cilrs_mi355 import CILRS, CONFIG_A, CONFIG_B, Trainer, TrainConfig
    from cilrs_mi355.parallel import broadcast_parameters
    torch.manual_seed(0)
    cfg = CONFIG_A if args.config == "A" else TrainConfig(**{**CONFIG_B.__dict__})
    model = CILRS(4, dropout=cfg.dropout).to(dev)
    trainer = Trainer(model, cfg, process_group=pg)
    trainer.fuse_optimizer = bool(args.fused_step)
    if pg is not None:
        broadcast_parameters(trainer.eng, pg)
    batch, u8 = synthetic_batch(args.batch, 1 + rank, dev)

    # ---- parity of THIS run (rank 0, one GPU, dropout-free config): what the GPU computes from
    # its initial weights on the batch it is about to time -- a train-mode forward and the first
    # train step -- is recomputed by the CPU oracle in cpu_baseline() below and must agree
    want_parity = (world == 1 and rank == 0 and not args.no_cpu_baseline and cfg.dropout == 0
                   and args.config == "A")
    state0 = gpu_out = gpu_loss1 = None
    if want_parity:
        state0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        model.train()
        with torch.no_grad():
            c0, s0 = model(batch[0], batch[1], batch[2])
        gpu_out = (c0.cpu(), s0.cpu())
        trainer.train_step(*batch)
        gpu_loss1 = trainer.losses()

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier(device_ids=[local])
            torch.cuda.synchronize(dev)

    log(f"rank {rank}/{world}: model + trainer ready, warm-up {args.warmup} steps")
    for _ in range(args.warmup):
        trainer.train_step(*batch)
    sync()
    log("timed region")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        trainer.train_step(*batch)
    e1.record()
    sync()
    wall = time.perf_counter() - t0
    t = torch.tensor([wall], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = float(t.item())
    loss_total = trainer.losses()["total"]

    # ---- per-bucket timing of the gradient all-reduce, each bucket alone on an idle GPU (every
    # rank takes part; in the step the buckets overlap the remaining backward)
    ar = None
    if pg is not None:
        ar = []
        for (_, b, e) in trainer.reducer.buckets:
            view = trainer.eng.grads[b:e]
            for _ in range(2):
                dist.all_reduce(view, group=pg)
            sync()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(5):
                dist.all_reduce(view, group=pg)
            ev1.record()
            sync()
            ms = ev0.elapsed_time(ev1) / 5
            tm = torch.tensor([ms], dtype=torch.float64, device=dev)
            if world > 1:
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            ms = float(tm.item())
            ar.append({"floats": e - b, "mbytes": round((e - b) * 4 / 1e6, 2), "ms": round(ms, 4),
                       "algbw_gbs": round((e - b) * 4 / 1e6 / max(ms, 1e-9), 1)})
        trainer.eng.grads.zero_()
    n_ranks_seen = dist.get_world_size() if dist.is_initialized() else 1

    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return
    frames = args.batch * world * args.steps
    value = frames / wall
    out = {
        "metric": "frames/sec CILRS train batch=128",
        "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"CILRS ResNet-34 train step (fwd+loss+bwd+Adam), Config "
                               f"{args.config}, 200x88 RGB, B={args.batch}/GPU, fp32",
                   "global_batch": args.batch * world,
                   "parallelism": f"dp{world}" if world > 1 else "single"},
        "n_ranks_seen": n_ranks_seen,
        "device_ms_per_step": round(e0.elapsed_time(e1) / args.steps, 3),
        "final_loss": round(loss_total, 6),
        "step_tflops": round(value * TRAIN_GFLOP_PER_FRAME / 1e3 / world, 2),
        "step_frac_of_f32_matrix_peak": round(
            value * TRAIN_GFLOP_PER_FRAME / 1e3 / world / PEAK_F32_MATRIX_TFLOPS, 4),
    }
    if ar is not None:
        out["allreduce"] = {"backend": "nccl (RCCL)", "buckets": ar,
                            "total_ms": round(sum(x["ms"] for x in ar), 4),
                            "note": "each bucket timed alone (max over ranks); in the step they "
                                    "overlap the remaining backward"}

    log(f"{value:.1f} frames/s, {wall / args.steps * 1e3:.3f} ms/step")
    if world == 1 and args.profile_steps > 0:
        # ---- per-kernel hipEvent timing on the launch stream (extra steps, same step fn) ----
        pl = trainer.eng.plan(args.batch, 88, 200)
        pl.profile_reset()
        pl.profile(True)
        for _ in range(args.profile_steps):
            trainer.train_step(*batch)
        torch.cuda.synchronize(dev)
        table = pl.profile_table()
        pl.profile(False)
        fam = {}
        for label, r in table.items():
            f = label.split(".")[0]
            a = fam.setdefault(f, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            for k in a:
                a[k] += r[k]
        total_ms = sum(a["ms"] for a in fam.values())
        igemm = dict(calls=fam.get("conv_fwd", {}).get("calls", 0) + fam.get("conv_dgrad", {}).get("calls", 0),
                     ms=fam.get("conv_fwd", {}).get("ms", 0.0) + fam.get("conv_dgrad", {}).get("ms", 0.0),
                     flops=fam.get("conv_fwd", {}).get("flops", 0.0) + fam.get("conv_dgrad", {}).get("flops", 0.0))
        wg = fam.get("conv_wgrad", dict(calls=0, ms=0.0, flops=0.0))
        nwino = pl.wino_convs()
        fam_name = "conv fwd+dgrad family: conv_igemm_kernel" + (
            f" + conv_wino_kernel ({2 * nwino} of the launches)" if nwino else "")
        dom_name, dom = (fam_name, igemm) if igemm["ms"] >= wg["ms"] \
            else ("conv_wgrad_kernel (+reduce)", wg)
        ach = dom["flops"] / max(dom["ms"], 1e-9) / 1e9          # TFLOP/s
        out["roofline"] = {
            "kernel": dom_name, "bound": "mfma", "achieved": round(ach, 2),
            "peak": PEAK_F32_MATRIX_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_F32_MATRIX_TFLOPS, 4), "traffic": None, "traffic_from": None,
            "launches_per_step": dom["calls"] // max(args.profile_steps, 1),
            "avg_launch_us": round(dom["ms"] / max(dom["calls"], 1) * 1e3, 2),
            "flops_per_launch": round(dom["flops"] / max(dom["calls"], 1), 1),
            "share_of_step": round(dom["ms"] / max(total_ms, 1e-9), 4),
            # ALGORITHMIC flops = the direct convolution's 2*M*Cout*K*K*Cin for every launch; a
            # Winograd F(2x2,3x3) launch issues 2.25x fewer multiplies for them
            "flops_basis": "direct convolution",
            "winograd_convs": nwino,
        }
        # What that fraction is NOT: occupancy of the matrix pipe.  A Winograd launch issues 2.25x
        # fewer multiplies than the direct-convolution flops it is credited with, so the share of
        # the fp32 MFMA issue rate the family actually uses is lower: frac_issued prices the
        # Winograd launches' flops at 1 / 2.25.  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES over all
        # SIMD-cycles per kernel from the committed rocprofv3 PMC pass (tools/pmc_mfma.sh ->
        # profiles/r04_mfma_util.json), not measured in this run.
        if dom is igemm and nwino == 24:
            # the 24 Winograd convolutions of the reference network: the stride-1 3x3 layers of
            # layers 1-3 (6 + 7 + 11), forward and data gradient each
            B_ = args.batch
            wino_flops = 2.0 * sum(n * 2.0 * B_ * h * w * c * 9 * c
                                   for n, h, w, c in ((6, 22, 50, 64), (7, 11, 25, 128), (11, 6, 13, 256)))
            wino_flops *= args.profile_steps
            issued = dom["flops"] - wino_flops * (1.0 - 1.0 / 2.25)
            out["roofline"]["frac_issued"] = round(issued / max(dom["ms"], 1e-9) / 1e9 / PEAK_F32_MATRIX_TFLOPS, 4)
        mpath = os.path.join(ROOT, "profiles", "r04_mfma_util.json")
        if os.path.exists(mpath):
            mj = json.load(open(mpath))
            out["roofline"]["mfma_busy"] = {k: v.get("mfma_busy_frac") for k, v in mj.items()
                                            if isinstance(v, dict) and "mfma_busy_frac" in v}
            out["roofline"]["mfma_busy_from"] = "profiles/r04_mfma_util.json (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES; tools/pmc_mfma.sh)"
        # HBM traffic per launch of that kernel family is NOT measured in this run: it is the
        # committed summary of separate rocprofv3 PMC passes over this same command (FETCH_SIZE
        # doubled, WRITE_SIZE as is -- MI355X_MICROARCH.md, HBM section; tools/pmc_traffic.sh)
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            key = "igemm" if dom is igemm else "wgrad"
            if key in tj:
                out["roofline"]["traffic"] = tj[key]["bytes_per_launch"]
                out["roofline"]["traffic_from"] = "profiles/traffic.json (" + tj.get(
                    "source", "rocprofv3 --pmc passes") + "; measured on commit " + tj.get(
                    "commit", "of round 2") + ")"
        out["kernels"] = {
            f: {"calls_per_step": a["calls"] // max(args.profile_steps, 1),
                "ms_per_step": round(a["ms"] / args.profile_steps, 4),
                "tflops": round(a["flops"] / max(a["ms"], 1e-9) / 1e9, 2) if a["flops"] else None,
                "gbs": round(a["bytes"] / max(a["ms"], 1e-9) / 1e6, 1) if a["bytes"] else None}
            for f, a in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}
        def timed(fn, reps=20):
            fn()
            torch.cuda.synchronize(dev)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(reps):
                fn()
            ev1.record()
            torch.cuda.synchronize(dev)
            return ev0.elapsed_time(ev1) / reps
        eng_ = trainer.eng
        snap = (eng_.params.clone(), trainer.exp_avg.clone(), trainer.exp_avg_sq.clone(),
                trainer.step_count)
        arena_bytes = eng_.n_arena * 4
        ms_adam = timed(lambda: trainer.optimizer_step(1.0))
        eng_.params.copy_(snap[0]); trainer.exp_avg.copy_(snap[1]); trainer.exp_avg_sq.copy_(snap[2])
        trainer.step_count = snap[3]
        pc_ = torch.zeros(args.batch, 3, device=dev)
        ps_ = torch.zeros(args.batch, device=dev)
        ms_loss = timed(lambda: trainer.loss(pc_, batch[3], ps_, batch[1]))
        if "adam" in out["kernels"]:
            # the step took cilrs_net_backward_step: its six per-segment Adam launches are already
            # in the profile (serialised there; in the timed steps they run on the weight-gradient
            # stream under the data gradients); the one-launch form is reported beside them
            out["kernels"]["adam"]["one_launch_ms"] = round(ms_adam, 4)
        else:
            out["kernels"]["adam" + ("+sqnorm" if cfg.grad_clip > 0 else "")] = {
                "calls_per_step": 2 if cfg.grad_clip > 0 else 1, "ms_per_step": round(ms_adam, 4),
                "tflops": None,
                "gbs": round((8 if cfg.grad_clip > 0 else 7) * arena_bytes / max(ms_adam, 1e-9) / 1e6, 1)}
        out["kernels"]["loss"] = {"calls_per_step": 1, "ms_per_step": round(ms_loss, 4),
                                  "tflops": None, "gbs": None}
        out["kernels_sum_ms"] = round(sum(k["ms_per_step"] for k in out["kernels"].values()), 3)
        out["kernels_by_layer"] = {
            l: {"ms_per_step": round(r["ms"] / args.profile_steps, 4),
                "tflops": round(r["flops"] / max(r["ms"], 1e-9) / 1e9, 2) if r["flops"] else None}
            for l, r in sorted(table.items()) if l.startswith("conv_")}

        log("per-kernel profile done; inference latency")
    if world == 1 and not args.no_infer:
        # ---- single-frame inference latency (predict_controls path) ----
        from cilrs_mi355.predict import Predictor
        pr = Predictor(model)
        frame = u8[0].numpy()
        for _ in range(20):
            pr.predict_controls(frame, 25.0, 0)
        lat = []
        for _ in range(1000):            # median of 1,000, readback included
            t1 = time.perf_counter()
            pr.predict_controls(frame, 25.0, 0)
            lat.append((time.perf_counter() - t1) * 1e3)
        lat.sort()
        out["infer_ms"] = round(lat[len(lat) // 2], 4)
        out["infer_ms_p99"] = round(lat[int(len(lat) * 0.99)], 4)
        out["infer_path"] = ("ONE persistent launch (csrc/infer_b1.hip: 38 stages, in-launch grid "
                             "barriers), frame / outputs in pinned host memory, one library call "
                             "per tick" if pr.persistent else "per-layer launches")
        # the same tick through the per-layer launch path (round 2's path), for comparison
        pr_l = Predictor(model, persistent=False)
        for _ in range(20):
            pr_l.predict_controls(frame, 25.0, 0)
        lat_l = []
        for _ in range(300):
            t1 = time.perf_counter()
            pr_l.predict_controls(frame, 25.0, 0)
            lat_l.append((time.perf_counter() - t1) * 1e3)
        lat_l.sort()
        out["infer_ms_per_layer_launches"] = round(lat_l[len(lat_l) // 2], 4)
        # BASELINE config 5's shape: 5 streams x 64 frames through the B=64 eval forward (uint8
        # frames resident on the device, outputs left on the device).  Served (a) one stream
        # after the other through one plan, (b) CONCURRENTLY: five lanes (a plan + workspace
        # each) on five HIP streams, eager and replayed from hipGraphs -- one launch's tail fills
        # with another stream's blocks (layers 3-4 at B=64 launch only 168-312 blocks).
        u64 = torch.randint(0, 256, (5, 64, 88, 200, 3), dtype=torch.uint8, device=dev)
        spd64 = torch.rand(64, device=dev)
        cmd64 = torch.randint(0, 4, (64,), device=dev)
        model.eval()
        eng = trainer.eng
        lanes = [torch.cuda.Stream(device=dev) for _ in range(5)]
        outs = [(torch.empty(64, 3, device=dev), torch.empty(64, device=dev)) for _ in range(5)]

        def serve(half, mode, reps=8):
            def once():
                for i in range(5):
                    if mode == "sequential":
                        eng.run_forward_u8(u64[i], spd64, cmd64, half=half)
                    else:
                        with torch.cuda.stream(lanes[i]):
                            eng.run_forward_u8(u64[i], spd64, cmd64, out=outs[i],
                                               graph=(mode == "graph"), half=half, lane=i + 1)

            def sync():
                torch.cuda.synchronize(dev)
            for _ in range(3):
                once()
            sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                once()
            sync()
            return (time.perf_counter() - t0) / reps            # seconds per 320 frames

        for key, half, label in (("infer_batch64", False, "f32"),
                                 ("infer_batch64_f16", True, "f16 trunk")):
            t_seq, t_lane, t_graph = (serve(half, m) for m in ("sequential", "lanes", "graph"))
            best = min(t_lane, t_graph)
            out[key] = {"frames_per_s": round(320 / best, 1),
                        "ms_per_320_frames": round(best * 1e3, 3), "dtype": label, "frames": 320,
                        "serving": "5 concurrent lanes (one plan + HIP stream per 64-frame stream)",
                        "lanes_eager_frames_per_s": round(320 / t_lane, 1),
                        "lanes_graph_frames_per_s": round(320 / t_graph, 1),
                        "sequential_frames_per_s": round(320 / t_seq, 1),
                        "sequential_ms_per_batch": round(t_seq / 5 * 1e3, 3)}
            # the same 320 frames as ONE batch (what a batching server would submit)
            u320 = u64.view(320, 88, 200, 3)
            s320, c320 = spd64.repeat(5), cmd64.repeat(5)
            for _ in range(2):
                eng.run_forward_u8(u320, s320, c320, half=half)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(5):
                eng.run_forward_u8(u320, s320, c320, half=half)
            torch.cuda.synchronize(dev)
            out[key]["one_batch_of_320_frames_per_s"] = round(320 * 5 / (time.perf_counter() - t0), 1)
            if half:
                # roofline of the config-5 leg: trunk convolutions on the fp16 matrix pipe (33 of
                # the 36: 2.70 of the 2.798 GFLOP per frame; stem + heads fp32) against the dense
                # fp16 peak, and the algorithmic bytes (16-bit activations in + out per layer,
                # folded weights once per batch) against HBM
                pl64 = eng.plan(64, 88, 200)
                pl64.profile_reset(); pl64.profile(True)
                for _ in range(3):
                    eng.run_forward_u8(u64[0], spd64, cmd64, half=True)
                torch.cuda.synchronize(dev)
                t64 = pl64.profile_table(); pl64.profile(False)
                cv = [v for k, v in t64.items() if k.startswith("conv_fwd.") and k != "conv_fwd.stem"]
                cms, cfl, cby = (sum(v[q] for v in cv) / 3 for q in ("ms", "flops", "bytes"))
                out[key]["roofline"] = {
                    "kernel": "conv_f16_kernel<fp16> (35 launches per 64-frame forward)", "bound": "mfma",
                    "achieved": round(320 * TRAIN_GFLOP_PER_FRAME / 3.0 / 1e3 / best, 1),
                    "peak": PEAK_F16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(320 * TRAIN_GFLOP_PER_FRAME / 3.0 / 1e3 / best / PEAK_F16_TFLOPS, 4),
                    "basis": "whole-forward flops of 320 frames / wall time of the five concurrent lanes",
                    "conv_kernels_serial": {"ms_per_64_frames": round(cms, 4),
                                            "tflops": round(cfl / max(cms, 1e-9) / 1e9, 1),
                                            "gbs": round(cby / max(cms, 1e-9) / 1e6, 1),
                                            "frac_mfma": round(cfl / max(cms, 1e-9) / 1e9 / PEAK_F16_TFLOPS, 4),
                                            "frac_hbm": round(cby / max(cms, 1e-9) / 1e6 / PEAK_HBM_GBS, 4)},
                    "profile": "profiles/r04_infer_f16/kernel_stats.csv (rocprofv3 --kernel-trace --stats)"}
        # device-side breakdown of one B=1 forward (eager launches, hipEvent per kernel)
        pl1 = trainer.eng.plan(1, 88, 200)
        pl1.profile_reset()
        pl1.profile(True)
        for _ in range(10):
            pr.predict_controls(frame, 25.0, 0)
        torch.cuda.synchronize(dev)
        t1 = pl1.profile_table()
        pl1.profile(False)
        out["infer_device_us"] = {k: round(v["ms"] / 10 * 1e3, 1)
                                  for k, v in sorted(t1.items(), key=lambda kv: -kv[1]["ms"])}
        out["infer_device_us"]["total"] = round(sum(v["ms"] for v in t1.values()) / 10 * 1e3, 1)
        # roofline-style view of the single-frame forward: what the frame costs on paper
        # (2.798 GFLOP at the fp32 matrix peak + every weight once from HBM) against the device
        # time of the one launch
        dev_us = out["infer_device_us"].get("infer_b1", out["infer_device_us"]["total"])
        floor_us = (TRAIN_GFLOP_PER_FRAME / 3.0) / PEAK_F32_MATRIX_TFLOPS * 1e3 + \
            trainer.eng.n_arena * 4 / (PEAK_HBM_GBS * 1e9) * 1e6
        out["infer_roofline"] = {
            "device_us": dev_us, "launches": 1 if pr.persistent else None, "stages": 38,
            "floor_us": round(floor_us, 1), "frac": round(floor_us / max(dev_us, 1e-9), 4),
            "floor": "2.798 GFLOP / 157.3 TFLOP/s + 89.7 MB of weights / 8 TB/s; the launch is "
                     f"bound by 37 dependent grid-wide hand-offs ({dev_us / 38:.1f} us per stage: "
                     "barrier ~2 us + the stage's dependent memory-side round trips), not by either"}
        model.train()

        # ---- the input pipeline in front of the step (the reference's 2-worker
        # JPEG DataLoader, notebook.ipynb cell lines 387-431, would starve this consumer): decode
        # pool -> pinned batch -> H2D -> fused augmentation kernel, alone and feeding the train
        # step, on synthetic 200x88 JPEGs the leg writes itself (bounded: 2,048 frames)
        if not args.no_loader:
            try:
                import importlib.util
                spec = importlib.util.spec_from_file_location(
                    "loader_bench", os.path.join(ROOT, "tools", "loader_bench.py"))
                lb = importlib.util.module_from_spec(spec)
                spec.loader.exec_module(lb)
                from cilrs_mi355.hostinfo import usable_cores
                res = lb.measure(frames=2048, batch=args.batch, workers=usable_cores(),
                                 trainer=trainer, log=log)
                out["loader"] = res
                out["loader_frames_per_s"] = res["loader_frames_per_s"]
            except Exception as e:                      # the headline must not depend on this leg
                out["loader"] = {"error": repr(e)}

        # ---- BASELINE configs[3] on this GPU's share: ResNet-50 variant, 176x400 frames, trunk
        # on the bf16 matrix pipe (BatchNorm folded), batched inference at 64 frames per call.
        # The reference has no such model; random-init weights, synthetic uint8 frames resident
        # in HBM.  Roofline: algorithmic conv FLOPs and bytes of the bf16 trunk per forward.
        from cilrs_mi355 import CILRSResNet50
        torch.manual_seed(0)
        m50 = CILRSResNet50(4, 0.0).to(dev).eval()
        eng50 = m50.engine()
        b50 = 64
        u50 = torch.randint(0, 256, (b50, 176, 400, 3), dtype=torch.uint8, device=dev)
        s50 = torch.rand(b50, device=dev)
        c50 = torch.randint(0, 4, (b50,), device=dev)
        for _ in range(3):
            eng50.run_forward_u8(u50, s50, c50, half="bf16")
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(10):
            eng50.run_forward_u8(u50, s50, c50, half="bf16")
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t1) / 10
        pl50 = eng50.plan(b50, 176, 400)
        pl50.profile_reset()
        pl50.profile(True)
        for _ in range(3):
            eng50.run_forward_u8(u50, s50, c50, half="bf16")
        torch.cuda.synchronize(dev)
        t50 = pl50.profile_table()
        pl50.profile(False)
        conv = [v for k, v in t50.items() if k.startswith("conv_fwd") and k != "conv_fwd.stem"]
        cms = sum(v["ms"] for v in conv) / 3
        cfl = sum(v["flops"] for v in conv) / 3
        cby = sum(v["bytes"] for v in conv) / 3
        ncalls = sum(v["calls"] for v in conv) // 3
        tf, gbs = cfl / max(cms, 1e-9) / 1e9, cby / max(cms, 1e-9) / 1e6
        out["resnet50_bf16"] = {
            "workload": "CILRS ResNet-50 variant eval forward, 176x400 RGB, B=64, bf16 trunk "
                        "(fp32 stem + heads), random-init weights, synthetic frames",
            "frames_per_s": round(b50 / dt, 1), "ms_per_batch": round(dt * 1e3, 3),
            "dtype": "bf16", "n_gpus": 1,
            "roofline": {"kernel": "conv_f16_kernel<bf16> (52 launches per forward)",
                         "bound": "mfma" if tf / 2500.0 >= gbs / PEAK_HBM_GBS else "hbm",
                         "achieved_tflops": round(tf, 1), "peak_tflops": 2500.0,
                         "frac_mfma": round(tf / 2500.0, 4),
                         "achieved_gbs": round(gbs, 1), "peak_gbs": PEAK_HBM_GBS,
                         "frac_hbm": round(gbs / PEAK_HBM_GBS, 4),
                         "launches_per_forward": ncalls,
                         "avg_launch_us": round(cms / max(ncalls, 1) * 1e3, 2),
                         "gflop_per_frame_trunk": round(cfl / b50 / 1e9, 3)},
            "device_ms_by_layer": {k: round(v["ms"] / 3, 4) for k, v in sorted(t50.items())}}
        del pl50, u50
        # ... and the same variant TRAINING in fp32 (Bottleneck chains through the fp32 kernels
        # of the headline path; no bf16 training kernels exist): Config A step at 176x400
        from cilrs_mi355 import CONFIG_A as _CA
        b50t = int(os.environ.get("CILRS_BENCH_R50_TRAIN_B", "64"))
        tr50 = Trainer(m50, _CA)
        g50 = torch.Generator(device="cpu").manual_seed(5)
        batch50 = (torch.randn(b50t, 3, 176, 400, generator=g50).to(dev),
                   torch.rand(b50t, generator=g50).to(dev),
                   torch.randint(0, 4, (b50t,), generator=g50).to(dev),
                   torch.rand(b50t, 3, generator=g50).to(dev))
        # (four warm-up steps: with two, one-off costs of this leg's first steps -- plan build,
        #  allocator growth after the previous legs' empty_cache -- leaked into a five-step timing
        #  on some runs: 29.7 ms where the steady state is 27.3)
        for _ in range(4):
            tr50.train_step(*batch50)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(8):
            tr50.train_step(*batch50)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t1) / 8
        l50 = tr50.losses()
        plt = eng50.plan(b50t, 176, 400)
        plt.profile_reset()
        plt.profile(True)
        for _ in range(2):
            tr50.train_step(*batch50)
        torch.cuda.synchronize(dev)
        tt = plt.profile_table()
        plt.profile(False)
        conv = [v for k, v in tt.items() if k.startswith("conv_")]
        cms = sum(v["ms"] for v in conv) / 2
        cfl = sum(v["flops"] for v in conv) / 2
        out["resnet50_train_f32"] = {
            "workload": f"CILRS ResNet-50 variant train step (fwd+loss+bwd+Adam), Config A, "
                        f"176x400 RGB, B={b50t}, fp32, random-init weights, synthetic batch",
            "frames_per_s": round(b50t / dt, 1), "ms_per_step": round(dt * 1e3, 3),
            "dtype": "f32", "n_gpus": 1, "final_loss": round(l50["total"], 6),
            "conv_tflops": round(cfl / max(cms, 1e-9) / 1e9, 1),
            "conv_frac_of_f32_matrix_peak": round(cfl / max(cms, 1e-9) / 1e9 / PEAK_F32_MATRIX_TFLOPS, 4),
            "gflop_per_frame": round(cfl / b50t / 1e9, 2),
            "device_ms_by_kernel": {k: round(v["ms"] / 2, 3) for k, v in
                                    sorted(tt.items(), key=lambda kv: -kv[1]["ms"])[:12]}}
        del tr50, plt
        torch.cuda.empty_cache()
        # ... and on the bf16 matrix pipe (Trainer(precision="bf16"): trunk convolutions of the
        # step multiply bf16 operands, fp32 accumulation; BatchNorm / heads / Adam / master
        # weights fp32) -- BASELINE configs[3] as stated, one GPU's share
        def bf16_leg(model, batch_t, tag, steps=10):
            trb = Trainer(model, _CA, precision="bf16")
            for _ in range(4):
                trb.train_step(*batch_t)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(steps):
                trb.train_step(*batch_t)
            torch.cuda.synchronize(dev)
            dtb = (time.perf_counter() - t0) / steps
            lb = trb.losses()
            b, _, hh, ww = batch_t[0].shape
            plb = trb.eng.plan(b, hh, ww)
            plb.profile_reset()
            plb.profile(True)
            for _ in range(2):
                trb.train_step(*batch_t)
            torch.cuda.synchronize(dev)
            tb = plb.profile_table()
            plb.profile(False)
            convb = [v for k, v in tb.items() if k.startswith("conv_") and not k.endswith(".stem")]
            cmsb = sum(v["ms"] for v in convb) / 2
            cflb = sum(v["flops"] for v in convb) / 2
            return {"workload": tag, "frames_per_s": round(b / dtb, 1),
                    "ms_per_step": round(dtb * 1e3, 3), "dtype": "bf16 operands, f32 accumulate",
                    "final_loss": round(lb["total"], 6),
                    "roofline": {"kernel": "conv_f16_kernel<bf16, TRAIN> / conv16p_kernel<bf16> + "
                                           "wgrad_f16_kernel<bf16> (forward, data gradient, weight "
                                           "gradient of the trunk; 16-bit tensors end to end)",
                                 "bound": "mfma", "achieved": round(cflb / max(cmsb, 1e-9) / 1e9, 1),
                                 "peak": 2500.0, "unit": "TFLOP/s",
                                 "frac": round(cflb / max(cmsb, 1e-9) / 1e9 / 2500.0, 4),
                                 "launches_per_step": int(sum(v["calls"] for v in convb) // 2),
                                 "note": "hipEvent brackets of the serialised step; the BatchNorm "
                                         "passes around the GEMMs (bf16 tensors, at the HBM / "
                                         "Infinity-Cache rate) are listed in device_ms_by_kernel"},
                    "device_ms_by_kernel": {k: round(v["ms"] / 2, 3) for k, v in
                                            sorted(tb.items(), key=lambda kv: -kv[1]["ms"])[:12]}}
        torch.manual_seed(0)
        m50b = CILRSResNet50(4, 0.0).to(dev)
        out["resnet50_train_bf16"] = bf16_leg(
            m50b, batch50, f"CILRS ResNet-50 variant train step, Config A, 176x400 RGB, "
                           f"B={b50t}, bf16 matrix pipe, random-init weights, synthetic batch")
        del m50b, m50, eng50, batch50
        torch.cuda.empty_cache()
        torch.manual_seed(0)
        m34b = CILRS(4, dropout=0.0).to(dev)
        out["train_bf16"] = bf16_leg(
            m34b, batch, f"CILRS ResNet-34 train step, Config A, 200x88 RGB, B={args.batch}, bf16 "
                         f"matrix pipe (NOT the headline: the reference's arithmetic is fp32)")
        del m34b
        torch.cuda.empty_cache()

    if want_parity:
        log("CPU oracle: parity of this run's batch + host baseline")
        batch_cpu = [t.cpu() for t in batch]
        base, err_out, err_loss = cpu_baseline(batch_cpu, state0, gpu_out, gpu_loss1)
        out["cpu_baseline"] = base
        out["parity_max_abs_err"] = float(f"{err_out:.3e}")
        out["parity_loss_err"] = float(f"{err_loss:.3e}")
        out["parity"] = ("train-mode forward outputs and first-step loss terms of this run's "
                         "batch vs the CPU oracle from the same initial weights; gate 1e-4")
        if not (err_out <= 1e-4 and err_loss <= 1e-4):
            log(f"PARITY FAILURE: outputs {err_out:.3e}, loss {err_loss:.3e} (gate 1e-4)")
            sys.stdout.flush()
            os.dup2(real_stdout, 1)
            print(json.dumps(out), flush=True)
            raise SystemExit(3)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
