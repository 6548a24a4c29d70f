"""Input pipeline for real driving data (reference notebook/notebook.ipynb:353-431,
on-disk format model/collect_data.py:545-564, 683-716, model/prepare_dataset.py:47-61).

On disk:  <data_dir>/sessionN/measurements.csv  (14 columns; this loader uses image_filename, steer,
throttle, brake, speed_normalized, command_name) and  <data_dir>/sessionN/images/frame_%08d.jpg
(200x88 JPEG).  The reference decodes and augments each frame on two CPU DataLoader workers; here
the host only parses the CSVs, draws the class-balanced sample indices and each sample's random
augmentation parameters, and decodes JPEGs on a thread pool into a pinned uint8 batch -- the
augmentation itself, /255 and Normalize run as one HIP kernel (`cilrs_augment_u8`).
"""
from __future__ import annotations

import csv
import ctypes as C
import os
import queue
import threading

import numpy as np
import torch

from . import _lib as L

COMMAND_MAP = {"LANEFOLLOW": 0, "LEFT": 1, "RIGHT": 2, "STRAIGHT": 3}     # notebook.ipynb:358
IMG_HEIGHT, IMG_WIDTH = 88, 200

# numpy mirror of `cilrs_aug_params` (include/cilrs_hip.h), 112 bytes
AUG_DTYPE = np.dtype([
    ("noise_seed", np.uint64), ("rbc_on", np.int32), ("alpha", np.float32),
    ("beta255", np.float32), ("hsv_on", np.int32), ("hue", np.float32), ("sat", np.float32),
    ("val", np.float32), ("blur_k", np.int32), ("blur_w", np.float32, 3),
    ("noise_std255", np.float32), ("nholes", np.int32), ("hole_y0", np.int32, 3),
    ("hole_x0", np.int32, 3), ("hole_y1", np.int32, 3), ("hole_x1", np.int32, 3),
    ("reserved", np.int32)], align=True)
assert AUG_DTYPE.itemsize == 112


def identity_params(batch: int) -> np.ndarray:
    """Every augmentation off (validation: only /255 + Normalize)."""
    p = np.zeros(batch, dtype=AUG_DTYPE)
    p["alpha"] = 1.0
    return p


def gaussian_taps(ksize: int, sigma: float) -> np.ndarray:
    """cv2.getGaussianKernel(ksize, sigma) as (centre, +-1, +-2) float32 weights."""
    r = ksize // 2
    x = np.arange(-r, r + 1, dtype=np.float64)
    w = np.exp(-(x * x) / (2.0 * sigma * sigma))
    w /= w.sum()
    out = np.zeros(3, dtype=np.float32)
    out[:r + 1] = w[r:].astype(np.float32)
    return out


def draw_aug_params(rng: np.random.Generator, batch: int, height: int = IMG_HEIGHT,
                    width: int = IMG_WIDTH) -> np.ndarray:
    """Per-sample parameters of the reference's train_augmentation (notebook.ipynb:387-394):
    RandomBrightnessContrast(0.2, 0.2, p=.5), HueSaturationValue(10, 20, 15, p=.3),
    GaussianBlur(blur_limit=(3,5), p=.2), GaussNoise(std_range=(.02,.06), p=.3),
    CoarseDropout(1-3 holes, h 4-10, w 8-20, fill 0, p=.2)."""
    p = identity_params(batch)
    on = rng.random((5, batch))
    m = on[0] < 0.5
    p["rbc_on"] = m
    p["alpha"] = np.where(m, 1.0 + rng.uniform(-0.2, 0.2, batch), 1.0)
    p["beta255"] = np.where(m, rng.uniform(-0.2, 0.2, batch) * 255.0, 0.0)
    m = on[1] < 0.3
    p["hsv_on"] = m
    p["hue"] = np.where(m, rng.uniform(-10, 10, batch), 0.0)
    p["sat"] = np.where(m, rng.uniform(-20, 20, batch), 0.0)
    p["val"] = np.where(m, rng.uniform(-15, 15, batch), 0.0)
    m = on[2] < 0.2
    ks = rng.choice((3, 5), batch)
    sig = rng.uniform(0.5, 3.0, batch)
    for i in np.nonzero(m)[0]:
        p["blur_k"][i] = ks[i]
        p["blur_w"][i] = gaussian_taps(int(ks[i]), float(sig[i]))
    m = on[3] < 0.3
    p["noise_std255"] = np.where(m, rng.uniform(0.02, 0.06, batch) * 255.0, 0.0)
    p["noise_seed"] = np.where(m, rng.integers(0, 1 << 63, batch, dtype=np.uint64), np.uint64(0))
    m = on[4] < 0.2
    nh = rng.integers(1, 4, batch)
    hh = rng.integers(4, 11, (batch, 3))
    ww = rng.integers(8, 21, (batch, 3))
    y0 = (rng.random((batch, 3)) * (height - hh + 1)).astype(np.int64)
    x0 = (rng.random((batch, 3)) * (width - ww + 1)).astype(np.int64)
    live = m[:, None] & (np.arange(3)[None, :] < nh[:, None])
    p["nholes"] = np.where(m, nh, 0)
    p["hole_y0"] = np.where(live, y0, 0)
    p["hole_x0"] = np.where(live, x0, 0)
    p["hole_y1"] = np.where(live, y0 + hh, 0)
    p["hole_x1"] = np.where(live, x0 + ww, 0)
    return p


def check_params(p: np.ndarray, height: int, width: int):
    if p.dtype != AUG_DTYPE:
        raise RuntimeError("augmentation parameters must use data.AUG_DTYPE")
    if not np.isin(p["blur_k"], (0, 1, 3, 5)).all():
        raise RuntimeError("blur_k must be 0, 1, 3 or 5")
    if (p["nholes"] < 0).any() or (p["nholes"] > 3).any():
        raise RuntimeError("nholes must be 0..3")
    if (p["hole_y1"] > height).any() or (p["hole_x1"] > width).any() or \
            (p["hole_y0"] < 0).any() or (p["hole_x0"] < 0).any():
        raise RuntimeError("dropout hole outside the frame")


def augment_u8(frames_u8: torch.Tensor, params: np.ndarray, want_u8: bool = False,
               params_dev: torch.Tensor = None):
    """frames uint8 [B,H,W,3] on the device + per-sample parameters -> normalised image as the
    logical NCHW float tensor `CILRS.forward` takes (a permuted view of the NHWC result, like the
    reference's permute at notebook.ipynb:413) [, augmented uint8 frames]."""
    if frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.size(3) != 3 \
            or frames_u8.device.type != "cuda":
        raise RuntimeError("augment_u8: frames must be uint8 [B,H,W,3] on the GPU (no CPU fallback)")
    b, h, w = frames_u8.size(0), frames_u8.size(1), frames_u8.size(2)
    if params.shape != (b,):
        raise RuntimeError("augment_u8: one parameter record per frame")
    check_params(params, h, w)
    frames_u8 = frames_u8.contiguous()
    # params_dev: the same records already on the device (uint8 [B, 112]; the loader uploads them
    # from pinned memory on its copy stream).  Without it they go up from pageable memory, which
    # makes the host wait for everything queued on this stream -- fine for one-off calls, a full
    # host-device synchronisation per step inside a training loop.
    if params_dev is not None:
        if params_dev.dtype != torch.uint8 or tuple(params_dev.shape) != (b, AUG_DTYPE.itemsize) \
                or params_dev.device != frames_u8.device:
            raise RuntimeError("augment_u8: params_dev must be uint8 [B, %d] on the frames' device"
                               % AUG_DTYPE.itemsize)
        pdev = params_dev.contiguous()
    else:
        pdev = torch.from_numpy(params.view(np.uint8).reshape(b, AUG_DTYPE.itemsize)).to(
            frames_u8.device, non_blocking=False)
    out = torch.empty(b, h, w, 3, dtype=torch.float32, device=frames_u8.device)
    out8 = torch.empty_like(frames_u8) if want_u8 else None
    stream = torch.cuda.current_stream(frames_u8.device).cuda_stream
    L.check(L.lib().cilrs_augment_u8(L.ptr(frames_u8), L.ptr(pdev), b, h, w, L.ptr(out),
                                     L.ptr(out8), C.c_void_p(stream)))
    img = out.permute(0, 3, 1, 2)
    return (img, out8) if want_u8 else img


# ---- dataset on disk -------------------------------------------------------------------------
class Sessions:
    """All `sessionN/measurements.csv` under data_dir, concatenated (notebook.ipynb:360-372)."""

    def __init__(self, data_dir: str):
        names = sorted(d for d in os.listdir(data_dir)
                       if os.path.isdir(os.path.join(data_dir, d)) and "session" in d)
        if not names:
            raise RuntimeError(f"no session directories under {data_dir}")
        paths, cols = [], {k: [] for k in ("steer", "throttle", "brake", "speed_normalized")}
        cmds = []
        for s in names:
            with open(os.path.join(data_dir, s, "measurements.csv"), newline="") as f:
                for row in csv.DictReader(f):
                    paths.append(os.path.join(data_dir, s, "images", row["image_filename"]))
                    for k in cols:
                        cols[k].append(float(row[k]))
                    if row["command_name"] not in COMMAND_MAP:
                        raise RuntimeError(f"unknown command_name {row['command_name']!r} in {s}")
                    cmds.append(COMMAND_MAP[row["command_name"]])
        self.paths = np.array(paths, dtype=object)
        self.targets = np.stack([np.array(cols[k], dtype=np.float32)
                                 for k in ("steer", "throttle", "brake")], axis=1)
        self.speed = np.array(cols["speed_normalized"], dtype=np.float32)
        self.command = np.array(cmds, dtype=np.int64)

    def __len__(self):
        return len(self.paths)

    def split(self, test_size=0.15, random_state=42):
        """The reference's stratified split (notebook.ipynb:376-377) -> (train_idx, val_idx)."""
        from sklearn.model_selection import train_test_split
        idx = np.arange(len(self))
        tr, va = train_test_split(idx, test_size=test_size, random_state=random_state,
                                  stratify=self.command)
        return tr, va


def class_balanced_weights(command: np.ndarray) -> np.ndarray:
    """Per-sample weight len / (4 * count[command]) (notebook.ipynb:383-384, 421)."""
    counts = np.bincount(command, minlength=4).astype(np.float64)
    cw = len(command) / (4.0 * np.where(counts > 0, counts, 1.0))
    return cw[command]


def weighted_indices(weights: np.ndarray, num_samples: int, generator=None) -> torch.Tensor:
    """WeightedRandomSampler(weights, num_samples, replacement=True) (notebook.ipynb:422-423):
    torch.multinomial over the double weights."""
    return torch.multinomial(torch.as_tensor(weights, dtype=torch.double), num_samples, True,
                             generator=generator)


def decode_jpeg(path: str) -> np.ndarray:
    """cv2.imread + BGR2RGB (notebook.ipynb:408-409) -> uint8 RGB [H,W,3]."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


class BatchLoader:
    """Iterates (image, speed, command, targets) device batches like the reference's DataLoader
    (notebook.ipynb:428-431).  train=True: class-balanced sampling with replacement, device-side
    augmentation, drop_last; train=False: sequential, no augmentation, last partial batch kept.
    JPEGs are decoded by `workers` threads into pinned memory one batch ahead of the consumer."""

    def __init__(self, sessions: Sessions, indices, batch_size: int, device, train: bool,
                 seed: int = 0, workers: int = 8, height: int = IMG_HEIGHT, width: int = IMG_WIDTH,
                 processes: bool = False, rank: int = 0, world_size: int = 1):
        """processes=True decodes in `workers` spawned processes (no GIL contention: what a
        B=128 / 14 ms training step needs); the default thread pool starts instantly.

        Data parallel (rank, world_size): every rank draws the SAME epoch order from the shared
        sampler stream (same `seed`) and keeps every world_size-th index starting at `rank`
        (SURVEY.md 8e "rank-strided sampling"): the ranks' index sets are disjoint and their
        union is exactly the single-process draw, so N ranks x batch B consume what one process
        at batch N*B would.  The order is cut to a multiple of world_size * batch_size when
        training (drop_last on the GLOBAL batch) so every rank runs the same number of steps --
        a rank with one batch more would hang the others' all-reduce.  Augmentation parameters
        come from a rank-offset stream."""
        if not (0 <= rank < world_size):
            raise ValueError(f"rank {rank} outside world of {world_size}")
        self.s, self.idx = sessions, np.asarray(indices)
        self.bs, self.device, self.train = batch_size, torch.device(device), train
        self.h, self.w = height, width
        self.rank, self.world = rank, world_size
        self.rng = np.random.default_rng([seed, rank])
        self.gen = torch.Generator().manual_seed(seed)
        self.workers = max(1, workers)
        self.processes = processes
        self.pool = None
        self._slots = None
        self.weights = class_balanced_weights(sessions.command[self.idx]) if train else None

    def _ensure_pool(self):
        if self.pool is None:
            if self.processes:
                import multiprocessing as mp
                self.pool = mp.get_context("spawn").Pool(self.workers)
            else:
                from multiprocessing.pool import ThreadPool
                self.pool = ThreadPool(self.workers)
        return self.pool

    def close(self):
        if self.pool is not None:
            self.pool.terminate()
            self.pool.join()
            self.pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _shard_len(self):
        n = len(self.idx)
        if self.train:
            return n // (self.bs * self.world) * self.bs
        return len(range(self.rank, n, self.world))

    def __len__(self):
        n = self._shard_len()
        return n // self.bs if self.train else (n + self.bs - 1) // self.bs

    def _order(self):
        """This rank's sample order for one epoch (see __init__ for the sharding rule)."""
        if self.train:
            pick = weighted_indices(self.weights, len(self.idx), self.gen).numpy()
            order = self.idx[pick]
            order = order[:len(order) // (self.bs * self.world) * self.bs * self.world]
        else:
            order = self.idx
        return order[self.rank::self.world]

    def __iter__(self):
        try:
            from cilrs_jpeg_worker import decode_chunk
        except ImportError:          # the worker module sits next to the package directory
            import sys
            sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
            from cilrs_jpeg_worker import decode_chunk
        order = self._order()
        nb = len(self)
        q: queue.Queue = queue.Queue(maxsize=3)
        pool = self._ensure_pool()
        chunk = max(4, -(-self.bs // self.workers))
        batches = [order[b * self.bs:(b + 1) * self.bs] for b in range(nb)]

        def tasks():              # every batch cut into per-worker chunks, in order
            for ids in batches:
                for c in range(0, len(ids), chunk):
                    yield (list(self.s.paths[ids[c:c + chunk]]), self.h, self.w)

        # ring of pinned staging buffers, allocated (and touched) once: a fresh pinned allocation
        # costs ~15 ms and its first DMA another ~17 ms on this platform (tools/loader_probe.py)
        if self._slots is None:
            self._slots = []
            for _ in range(4):
                t = torch.zeros(self.bs, self.h, self.w, 3, dtype=torch.uint8).pin_memory()
                meta = (torch.zeros(self.bs, dtype=torch.float32).pin_memory(),
                        torch.zeros(self.bs, dtype=torch.int64).pin_memory(),
                        torch.zeros(self.bs, 3, dtype=torch.float32).pin_memory(),
                        torch.zeros(self.bs, AUG_DTYPE.itemsize, dtype=torch.uint8).pin_memory())
                self._slots.append((t, t.numpy(), meta))
            # host-to-device copies run on a stream of their own: the wait that frees a staging
            # slot then waits for the COPY, not for the train step the copy would otherwise queue
            # behind (with everything on one stream the host could never run ahead of the device:
            # 12.6 k frames/s feeding a 13.9 k frames/s step)
            self._copy_stream = torch.cuda.Stream(device=self.device) \
                if torch.device(self.device).type == "cuda" else None
        free: queue.Queue = queue.Queue()
        for k in range(len(self._slots)):
            free.put(k)

        def producer():
            try:
                results = pool.imap(decode_chunk, tasks())       # ordered, workers run ahead
                for ids in batches:
                    slot = free.get()
                    view = self._slots[slot][1]
                    k = 0
                    while k < len(ids):
                        part = next(results)
                        view[k:k + len(part)] = part
                        k += len(part)
                    params = draw_aug_params(self.rng, len(ids), self.h, self.w) if self.train \
                        else identity_params(len(ids))
                    q.put((slot, params, ids))
                q.put(None)
            except Exception as e:          # surface decode errors in the consumer
                q.put(e)
        threading.Thread(target=producer, daemon=True).start()
        while True:
            item = q.get()
            if item is None:
                return
            if isinstance(item, Exception):
                raise item
            slot, params, ids = item
            n = len(ids)
            pinned, _view, (sp_p, cm_p, tg_p, par_p) = self._slots[slot]
            par_p[:n] = torch.from_numpy(params.view(np.uint8).reshape(n, AUG_DTYPE.itemsize))
            sp_p[:n] = torch.from_numpy(np.ascontiguousarray(self.s.speed[ids], dtype=np.float32))
            cm_p[:n] = torch.from_numpy(np.ascontiguousarray(self.s.command[ids], dtype=np.int64))
            tg_p[:n] = torch.from_numpy(np.ascontiguousarray(self.s.targets[ids], dtype=np.float32))
            cs = self._copy_stream
            main = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(cs):
                frames = torch.empty(n, self.h, self.w, 3, dtype=torch.uint8, device=self.device)
                frames.copy_(pinned[:n], non_blocking=True)
                speeds = sp_p[:n].to(self.device, non_blocking=True)
                cmds = cm_p[:n].to(self.device, non_blocking=True)
                tgts = tg_p[:n].to(self.device, non_blocking=True)
                pars = par_p[:n].to(self.device, non_blocking=True)
                copied = torch.cuda.Event()
                copied.record(cs)
            main.wait_event(copied)          # device side: the consumers run behind the copies
            for t in (frames, speeds, cmds, tgts, pars):
                t.record_stream(main)        # (allocated under the copy stream, used on `main`)
            img = augment_u8(frames, params, params_dev=pars)
            batch = (img, speeds, cmds, tgts)
            copied.synchronize()            # the staging slot may be refilled now
            free.put(slot)
            yield batch
