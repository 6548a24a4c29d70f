"""Device input pipeline: cilrs_augment_u8 against its numpy restatement, and the
loader end to end on a dataset written in the reference's on-disk format.  albumentations / cv2 are
absent from the image, so the augmentation semantics themselves are parity-unpinned (see
oracle/augment_oracle.py); what is pinned here is HIP kernel == restatement."""
import os

import numpy as np
import pytest
import torch

import augment_oracle as AO
import cilrs_oracle as O
from test_host import make_sessions

pytestmark = pytest.mark.gpu


def _recs(p):
    out = []
    for r in p:
        out.append({k: (r[k].tolist() if hasattr(r[k], "tolist") else r[k]) for k in p.dtype.names})
    return out


def test_augment_kernel_matches_oracle_stage_by_stage():
    from cilrs_mi355 import data as D
    rng = np.random.default_rng(1)
    B, H, W = 6, 88, 200
    frames = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    frames[0, :10] = 0                      # black / grey / saturated corner cases for HSV
    frames[0, 10:20] = 255
    frames[0, 20:30] = 128
    p = D.identity_params(B)
    p["rbc_on"][0], p["alpha"][0], p["beta255"][0] = 1, 1.17, -31.5
    p["hsv_on"][1], p["hue"][1], p["sat"][1], p["val"][1] = 1, -7.3, 14.2, -9.9
    p["blur_k"][2], p["blur_w"][2] = 5, D.gaussian_taps(5, 1.1)
    p["blur_k"][3], p["blur_w"][3] = 3, D.gaussian_taps(3, 2.5)
    p["nholes"][4] = 2
    p["hole_y0"][4, :2], p["hole_x0"][4, :2] = (0, 80), (0, 185)
    p["hole_y1"][4, :2], p["hole_x1"][4, :2] = (7, 88), (15, 200)
    p["hsv_on"][5], p["hue"][5] = 1, 9.0                       # everything at once
    p["rbc_on"][5], p["alpha"][5], p["beta255"][5] = 1, 0.85, 20.0
    p["blur_k"][5], p["blur_w"][5] = 3, D.gaussian_taps(3, 0.9)
    p["nholes"][5] = 1
    p["hole_y0"][5, 0], p["hole_x0"][5, 0], p["hole_y1"][5, 0], p["hole_x1"][5, 0] = 40, 100, 50, 120
    img, out8 = D.augment_u8(torch.from_numpy(frames).cuda(), p, want_u8=True)
    want8, wantf = AO.augment_batch(frames, _recs(p))
    got8 = out8.cpu().numpy()
    for b in range(B):
        assert np.array_equal(got8[b], want8[b]), f"sample {b}"
    assert img.shape == (B, 3, H, W) and not img.is_contiguous()
    assert np.array_equal(img.permute(0, 2, 3, 1).cpu().numpy(), wantf)


def test_augment_noise_and_random_compose():
    from cilrs_mi355 import data as D
    rng = np.random.default_rng(4)
    B, H, W = 32, 88, 200
    frames = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    p = D.draw_aug_params(np.random.default_rng(8), B)
    p["noise_std255"][0], p["noise_seed"][0] = 10.2, 123456789
    assert (p["noise_std255"] > 0).sum() >= 4 and (p["blur_k"] > 1).sum() >= 2
    _, out8 = D.augment_u8(torch.from_numpy(frames).cuda(), p, want_u8=True)
    want8, _ = AO.augment_batch(frames, _recs(p))
    diff = np.abs(out8.cpu().numpy().astype(int) - want8.astype(int))
    # logf / cosf differ from numpy's by an ulp: a rounding flip of one grey level is allowed on
    # noisy samples only, and rarely
    assert diff.max() <= 1
    for b in range(B):
        if p["noise_std255"][b] == 0:
            assert diff[b].max() == 0, b
        else:
            assert (diff[b] > 0).mean() <= 2e-3, b
    # the noise really is ~N(0, std)
    q = D.identity_params(1)
    q["noise_std255"][0], q["noise_seed"][0] = 12.0, 42
    grey = torch.full((1, H, W, 3), 128, dtype=torch.uint8).cuda()
    _, n8 = D.augment_u8(grey, q, want_u8=True)
    n = n8.cpu().numpy().astype(np.float64) - 128
    assert abs(n.mean()) < 0.2 and abs(n.std() - 12.0) < 0.3


def test_loader_end_to_end_and_train_step(tmp_path):
    from cilrs_mi355 import CILRS, CONFIG_A, Trainer
    from cilrs_mi355 import data as D
    make_sessions(str(tmp_path), (14, 13))
    s = D.Sessions(str(tmp_path))
    tr_idx, va_idx = s.split()
    dev = torch.device("cuda")
    val = D.BatchLoader(s, va_idx, batch_size=2, device=dev, train=False)
    seen = 0
    for bi, (img, spd, cmd, tgt) in enumerate(val):
        ids = va_idx[bi * 2:(bi + 1) * 2]
        assert img.shape[1:] == (3, 88, 200) and cmd.dtype == torch.int64
        for k, i in enumerate(ids):       # un-augmented: == the reference's /255 + Normalize
            want = O.preprocess_frame(D.decode_jpeg(s.paths[i]))[0]
            assert torch.equal(img[k].cpu(), want)
            assert float(spd[k]) == float(s.speed[i]) and int(cmd[k]) == int(s.command[i])
            assert torch.equal(tgt[k].cpu(), torch.from_numpy(s.targets[i]))
        seen += len(ids)
    assert seen == len(va_idx) == 5 and len(val) == 3
    train = D.BatchLoader(s, tr_idx, batch_size=4, device=dev, train=True, seed=5)
    assert len(train) == len(tr_idx) // 4
    m = CILRS(4, dropout=0.0)
    m.load_state_dict(O.portable_state_dict(m.state_dict(), 0), strict=True)
    trainer = Trainer(m.cuda(), CONFIG_A)
    nb = 0
    for img, spd, cmd, tgt in train:
        assert img.shape == (4, 3, 88, 200)
        trainer.train_step(img, spd, cmd, tgt)
        assert np.isfinite(trainer.losses()["total"])
        nb += 1
    assert nb == len(train)
    # process-pool decoding (the production setting) yields the same batches as the thread pool
    with D.BatchLoader(s, va_idx, 2, dev, False, workers=2, processes=True) as pv:
        for (ia, _, ca, _), (ib, _, cb, _) in zip(pv, D.BatchLoader(s, va_idx, 2, dev, False)):
            assert torch.equal(ia, ib) and torch.equal(ca, cb)
    # same seed -> same sample order and augmentation parameters
    a = [c.cpu() for _, _, c, _ in D.BatchLoader(s, tr_idx, 4, dev, True, seed=5)]
    b = [c.cpu() for _, _, c, _ in D.BatchLoader(s, tr_idx, 4, dev, True, seed=5)]
    assert all(torch.equal(x, y) for x, y in zip(a, b))
