"""CPU restatement of the device input pipeline (TEST INFRASTRUCTURE ONLY -- never imported by the
product path).

Follows notebook/notebook.ipynb:387-394 (albumentations Compose: RandomBrightnessContrast,
HueSaturationValue, GaussianBlur, GaussNoise, CoarseDropout) and :412-414 (/255, Normalize).
PARITY UNPINNED against albumentations / cv2 (absent from the build image): every step restates
the libraries' published behaviour -- float32 brightness/contrast LUT truncated to uint8; 8-bit HSV
with H in half-degrees, shifted through three LUTs; Gaussian taps exp(-x^2 / 2 sigma^2) normalised,
reflect-101 border; per-channel additive Gaussian noise; zero-filled rectangles.  All arithmetic is
float32, one rounding per operation, in the order csrc/augment.hip performs it.
"""
from __future__ import annotations

import numpy as np

F = np.float32
IMG_MEAN = (0.485, 0.456, 0.406)
IMG_STD = (0.229, 0.224, 0.225)


def _clip255(x):
    return np.minimum(np.maximum(x, F(0)), F(255))


def _colour_stages(img_f, p):
    """img_f float32 [H,W,3] holding uint8 values -> same after brightness/contrast and HSV."""
    c = img_f.copy()
    if p["rbc_on"]:
        t = c * F(p["alpha"])
        t = t + F(p["beta255"])
        c = np.trunc(_clip255(t))
    if p["hsv_on"]:
        r, g, b = c[..., 0], c[..., 1], c[..., 2]
        v = np.maximum(r, np.maximum(g, b))
        mn = np.minimum(r, np.minimum(g, b))
        d = v - mn
        with np.errstate(divide="ignore", invalid="ignore"):
            s = np.where(v > 0, d * F(255) / v, F(0)).astype(F)
            hr = F(60) * (g - b) / d
            hg = F(120) + F(60) * (b - r) / d
            hb = F(240) + F(60) * (r - g) / d
        h = np.where(v == r, hr, np.where(v == g, hg, hb)).astype(F)
        h = np.where(d > 0, h, F(0)).astype(F)
        h = np.where(h < 0, h + F(360), h).astype(F)
        H = np.rint(h * F(0.5))
        H = np.where(H >= 180, H - F(180), H).astype(F)
        S = np.rint(s)
        H2 = np.fmod(H + F(p["hue"]), F(180))
        H2 = np.trunc(np.where(H2 < 0, H2 + F(180), H2).astype(F))
        S2 = np.trunc(_clip255(S + F(p["sat"])))
        V2 = np.trunc(_clip255(v + F(p["val"])))
        hh = H2 * F(2) / F(60)
        sec = np.floor(hh)
        f = hh - sec
        sn = S2 / F(255)
        pp = V2 * (F(1) - sn)
        qq = V2 * (F(1) - sn * f)
        tt = V2 * (F(1) - sn * (F(1) - f))
        isec = sec.astype(np.int32)
        R = np.choose(np.clip(isec, 0, 5), [V2, qq, pp, pp, tt, V2])
        G = np.choose(np.clip(isec, 0, 5), [tt, V2, V2, qq, pp, pp])
        B = np.choose(np.clip(isec, 0, 5), [pp, pp, tt, V2, V2, qq])
        c = np.stack([np.rint(_clip255(R)), np.rint(_clip255(G)), np.rint(_clip255(B))], -1).astype(F)
    return c


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def gaussian_noise(seed: int, n_elems: int):
    """Standard normal float32 samples, element e = (y*W + x)*3 + channel."""
    with np.errstate(over="ignore"):
        e = np.arange(n_elems, dtype=np.uint64)
        h = _splitmix64(np.uint64(seed) + e * np.uint64(0xD1B54A32D192ED03))
    u1 = ((h >> np.uint64(40)).astype(np.uint32).astype(F) + F(1)) * F(1.0 / 16777216.0)
    u2 = ((h >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(np.uint32).astype(F) * F(1.0 / 16777216.0)
    rad = np.sqrt(F(-2) * np.log(u1))
    return (rad * np.cos(F(6.2831853071795864) * u2)).astype(F)


def augment_one(frame_u8, p):
    """uint8 [H,W,3] + one sample's parameter dict -> augmented uint8 [H,W,3]."""
    H, W = frame_u8.shape[:2]
    c = _colour_stages(frame_u8.astype(F), p)
    if p["blur_k"] > 1:
        r = min(p["blur_k"] // 2, 2)
        w = [F(x) for x in p["blur_w"]]
        pad = np.pad(c, ((r, r), (r, r), (0, 0)), mode="reflect")
        acc = np.zeros_like(c)
        for dy in range(-r, r + 1):
            row = np.zeros_like(c)
            for dx in range(-r, r + 1):
                row = row + w[abs(dx)] * pad[r + dy:r + dy + H, r + dx:r + dx + W]
            acc = acc + w[abs(dy)] * row
        c = np.rint(_clip255(acc))
    if p["noise_std255"] > 0:
        n = gaussian_noise(p["noise_seed"], H * W * 3).reshape(H, W, 3)
        c = np.rint(_clip255(c + n * F(p["noise_std255"])))
    for k in range(min(p["nholes"], 3)):
        c[p["hole_y0"][k]:p["hole_y1"][k], p["hole_x0"][k]:p["hole_x1"][k]] = 0
    return c.astype(np.uint8)


def normalize(u8):
    """/255 and Normalize(mean, std) in float32 (notebook.ipynb:412-414) -> [..., 3] float32."""
    x = u8.astype(F) / F(255)
    return (x - np.array(IMG_MEAN, dtype=F)) / np.array(IMG_STD, dtype=F)


def augment_batch(frames_u8, params):
    """frames uint8 [B,H,W,3]; params: list of dicts (cilrs_aug_params fields)."""
    out = np.stack([augment_one(f, p) for f, p in zip(frames_u8, params)])
    return out, normalize(out)
