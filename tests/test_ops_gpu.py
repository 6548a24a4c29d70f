"""Op-level parity of the HIP kernels (through the C-ABI) against torch fp32 CPU ops -- the ops
the reference's path dispatches.  Tolerances are stated per test; matrix work is
exact-f32 MFMA (an fmaf chain), so differences are summation-order only."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from cilrs_mi355 import _lib as L
    return L


def dev(t):
    return t.contiguous().cuda()


def nhwc(t):      # NCHW cpu -> NHWC gpu
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def ohwi(w):      # OIHW cpu -> OHWI gpu
    return w.permute(0, 2, 3, 1).contiguous().cuda()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# (N, H, W, Cin, Cout, k, stride, pad) -- the real trunk shapes at small batch + odd tails
CONV_CASES = [
    (3, 22, 50, 64, 64, 3, 1, 1),      # layer1
    (2, 22, 50, 64, 128, 3, 2, 1),     # layer2.0.conv1 (stride 2)
    (2, 22, 50, 64, 128, 1, 2, 0),     # layer2 downsample
    (3, 11, 25, 128, 128, 3, 1, 1),    # layer2
    (2, 11, 25, 128, 256, 3, 2, 1),    # layer3.0.conv1
    (5, 6, 13, 256, 256, 3, 1, 1),     # layer3
    (2, 6, 13, 256, 512, 1, 2, 0),     # layer4 downsample
    (7, 3, 7, 512, 512, 3, 1, 1),      # layer4
    (1, 3, 7, 512, 512, 3, 1, 1),      # single frame (M = 21)
]


def _tol(ref, scale=2e-5):
    return scale * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("cfg,splitk", [(-1, 0), (0, 1), (1, 1), (2, 1), (1, 3), (2, 2),
                                        (3, 1), (4, 1), (5, 1), (5, 2), (4, 3)])
def test_conv_fwd(case, cfg, splitk):
    L = _lib()
    lib = L.lib()
    N, H, W, Cin, Cout, k, s, p = case
    if cfg in (0, 3) and Cout % 128:
        pytest.skip("128-wide tile needs Cout % 128 == 0")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (k * k * Cin) ** 0.5
    ref = F.conv2d(x, w, None, s, p)
    Ho, Wo = ref.shape[2], ref.shape[3]
    y = torch.full((N, Ho, Wo, Cout), float("nan"), device="cuda")
    scratch = torch.empty(8 * y.numel(), device="cuda")
    xd, wd = nhwc(x), ohwi(w)
    L.check(lib.cilrs_conv2d_fwd(L.ptr(xd), L.ptr(wd), L.ptr(y), N, H, W, Cin, Cout, k, k, s, p,
                                 cfg, splitk, L.ptr(scratch), scratch.numel(), stream()))
    torch.cuda.synchronize()
    got = y.cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() <= _tol(ref)


def test_conv_fwd_stem():
    """7x7/s2/p3, Cin = 3 padded to 4 (generic-tap path)."""
    L = _lib()
    lib = L.lib()
    N, H, W = 2, 88, 200
    g = torch.Generator().manual_seed(2)
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    ref = F.conv2d(x, w, None, 2, 3)
    x4 = torch.zeros(N, H, W, 4)
    x4[..., :3] = x.permute(0, 2, 3, 1)
    w4 = torch.zeros(64, 7, 7, 4)
    w4[..., :3] = w.permute(0, 2, 3, 1)
    y = torch.full((N, 44, 100, 64), float("nan"), device="cuda")
    x4d, w4d = dev(x4), dev(w4)          # keep alive: ptr() does not hold a reference
    for cfg in (-1, 1, 2, 4, 5):
        y.fill_(float("nan"))
        L.check(lib.cilrs_conv2d_fwd(L.ptr(x4d), L.ptr(w4d), L.ptr(y), N, H, W, 4, 64, 7, 7,
                                     2, 3, cfg, 0, None, 0, stream()))
        torch.cuda.synchronize()
        got = y.cpu().permute(0, 3, 1, 2)
        assert (got - ref).abs().max() <= _tol(ref), cfg


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("cfg,splitk,with_addend", [(-1, 0, False), (1, 1, True), (2, 2, True),
                                                    (4, 1, True), (5, 2, False), (5, 1, True)])
def test_conv_dgrad(case, cfg, splitk, with_addend):
    L = _lib()
    lib = L.lib()
    N, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, Cin, H, W, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (k * k * Cin) ** 0.5
    y = F.conv2d(x, w, None, s, p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    ref = x.grad
    add = torch.randn(N, H, W, Cin, generator=g) if with_addend else None
    dx = torch.full((N, H, W, Cin), float("nan"), device="cuda")
    scratch = torch.empty(8 * dx.numel(), device="cuda")
    addd = dev(add) if with_addend else None
    dyd, wd = nhwc(dy), ohwi(w)
    L.check(lib.cilrs_conv2d_dgrad(L.ptr(dyd), L.ptr(wd), L.ptr(dx), L.ptr(addd), N, H,
                                   W, Cin, Cout, k, k, s, p, cfg, splitk, L.ptr(scratch),
                                   scratch.numel(), stream()))
    torch.cuda.synchronize()
    got = dx.cpu()
    want = ref.permute(0, 2, 3, 1)
    if with_addend:
        want = want + add
    assert torch.isfinite(got).all()
    assert (got - want).abs().max() <= _tol(want)


@pytest.mark.parametrize("shape", [(5, 88, 200), (2, 40, 120), (3, 176, 400), (2, 30, 70), (1, 88, 200),
                                   (3, 9, 253), (2, 61, 445)])
def test_stem_conv_fwd_with_batch_statistics(shape):
    """cilrs_stem_conv_fwd (csrc/stem_f32.hip: conv 7x7 / stride 2 / pad 3 of the training step, the
    weights in registers, k = 7 x 22) against torch's conv2d on the CPU, and its per-tile column
    partials against the sums of its own output: the reference's frame size, the ResNet-50 variant's,
    odd sizes with partial last tiles (one right after the reference size that needs MORE LDS from
    the same kernel instantiation), rows that end inside / at the edge of a 64-pixel DMA segment and
    the widest rows either row pitch serves."""
    L = _lib()
    lib = L.lib()
    N, H, W = shape
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5
    ref = F.conv2d(x.double(), w.double(), None, 2, 3)
    Ho, Wo = ref.shape[2], ref.shape[3]
    x4 = torch.zeros(N, H, W, 4)
    x4[..., :3] = x.permute(0, 2, 3, 1)
    x4[..., 3] = 7.0                                 # (the pad channel must never reach the result)
    x4 = x4.cuda()
    y = torch.full((N, Ho, Wo, 64), float("nan"), device="cuda")
    rows = C.c_int(0)
    part = torch.full((2 * 64 * (N * Ho * Wo // 64 + 64),), float("nan"), device="cuda")
    L.check(lib.cilrs_stem_conv_fwd(L.ptr(x4), L.ptr(ohwi(w)), L.ptr(y), L.ptr(part), N, H, W,
                                    C.byref(rows), stream()))
    torch.cuda.synchronize()
    got = y.cpu().permute(0, 3, 1, 2).double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() <= _tol(ref, 1e-5)
    r = rows.value
    assert 0 < r <= N * Ho * Wo // 64 + 64
    pt = part[:2 * 64 * r].view(2, 64, r).double().sum(2).cpu()
    yd = y.double().view(-1, 64)
    s1, s2 = yd.sum(0).cpu(), (yd * yd).sum(0).cpu()
    assert (pt[0] - s1).abs().max() <= 1e-4 * max(1.0, float(s1.abs().max()))
    assert (pt[1] - s2).abs().max() <= 1e-5 * float(s2.abs().max())
    # without partials (NULL): same tensor
    y2 = torch.empty_like(y)
    L.check(lib.cilrs_stem_conv_fwd(L.ptr(x4), L.ptr(ohwi(w)), L.ptr(y2), None, N, H, W, None, stream()))
    assert torch.equal(y, y2)


@pytest.mark.parametrize("shape", [(3, 88, 200), (2, 176, 400), (1, 88, 200), (9, 86, 199), (2, 26, 400)])
def test_stem_conv_wgrad(shape):
    """cilrs_stem_conv_wgrad (csrc/stem_f32.hip: the reduction over output pixels on the matrix
    pipe, every wave the whole 64 x 147 product for its share of the pixels) against the float64
    gradient of torch's conv2d: the reference's frame size and the variant's, one frame, an odd
    height / width whose last tile is partial (43 rows = 10 tiles of 4 + 3; 199 pixels: a zero
    column on the right), and fewer tiles than workgroups."""
    L = _lib()
    lib = L.lib()
    N, H, W = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, 3, H, W, generator=g).double().requires_grad_(False)
    w = (torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5).double().requires_grad_(True)
    y = F.conv2d(x, w, None, 2, 3)
    dy = torch.randn(y.shape, generator=g).double()
    (ref,) = torch.autograd.grad(y, w, dy)
    Ho, Wo = y.shape[2], y.shape[3]
    need = lib.cilrs_stem_conv_wgrad_scratch_floats(N, H, W)
    assert need > 0, "geometry must be served"
    x4 = torch.zeros(N, H, W, 4)
    x4[..., :3] = x.float().permute(0, 2, 3, 1)
    x4[..., 3] = 5.0
    x4 = x4.cuda()
    dyd = dy.float().permute(0, 2, 3, 1).contiguous().cuda()
    dw = torch.full((64, 7, 7, 3), float("nan"), device="cuda")
    scratch = torch.full((need,), float("nan"), device="cuda")
    L.check(lib.cilrs_stem_conv_wgrad(L.ptr(x4), L.ptr(dyd), L.ptr(dw), L.ptr(scratch), need, N, H, W,
                                      stream()))
    torch.cuda.synchronize()
    got = dw.cpu().permute(0, 3, 1, 2).double()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() <= 2e-5 * max(1.0, float(ref.abs().max()))
    # deterministic: a second launch gives the same bits
    dw2 = torch.empty_like(dw)
    L.check(lib.cilrs_stem_conv_wgrad(L.ptr(x4), L.ptr(dyd), L.ptr(dw2), L.ptr(scratch), need, N, H, W,
                                      stream()))
    assert torch.equal(dw, dw2)
    assert lib.cilrs_stem_conv_wgrad_scratch_floats(2, 30, 70) == 0      # (served by cilrs_conv2d_wgrad)


@pytest.mark.parametrize("case", [(2, 22, 50, 64, 128, 1, 2, 0), (3, 11, 25, 128, 256, 1, 2, 0),
                                  (2, 22, 50, 64, 128, 3, 2, 1), (2, 7, 9, 64, 128, 1, 2, 0)])
def test_conv_dgrad_stride2_in_place(case):
    """dx += dgrad(dy) IN PLACE (addend == dx), the form the down-sample branch's data gradient
    takes in the train step: parity classes no filter tap reaches (three of four for a 1x1 / s2
    filter) get no blocks at all -- their pixels must keep the value they had."""
    L = _lib()
    lib = L.lib()
    N, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(31)
    x = torch.randn(N, Cin, H, W, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (k * k * Cin) ** 0.5
    y = F.conv2d(x, w, None, s, p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    add = torch.randn(N, H, W, Cin, generator=g)
    want = x.grad.permute(0, 2, 3, 1) + add
    dx = dev(add.clone())
    dyd, wd = nhwc(dy), ohwi(w)
    scratch = torch.empty(8 * dx.numel(), device="cuda")
    L.check(lib.cilrs_conv2d_dgrad(L.ptr(dyd), L.ptr(wd), L.ptr(dx), L.ptr(dx), N, H, W, Cin, Cout,
                                   k, k, s, p, -1, 0, L.ptr(scratch), scratch.numel(), stream()))
    torch.cuda.synchronize()
    assert (dx.cpu() - want).abs().max() <= _tol(want)


@pytest.mark.parametrize("case", CONV_CASES + [(4, 22, 50, 64, 64, 3, 1, 1)])
def test_conv_wgrad(case):
    L = _lib()
    lib = L.lib()
    N, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(4)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g, requires_grad=True)
    y = F.conv2d(x, w, None, s, p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    ref = w.grad.permute(0, 2, 3, 1)
    nsc = lib.cilrs_conv2d_wgrad_scratch_floats(N, H, W, Cin, Cout, k, k, s, p)
    scratch = torch.empty(nsc, device="cuda")
    dw = torch.full((Cout, k, k, Cin), float("nan"), device="cuda")
    xd, dyd = nhwc(x), nhwc(dy)
    L.check(lib.cilrs_conv2d_wgrad(L.ptr(xd), L.ptr(dyd), L.ptr(dw), L.ptr(scratch), N,
                                   H, W, Cin, Cout, k, k, s, p, Cin, stream()))
    torch.cuda.synchronize()
    got = dw.cpu()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max() <= _tol(ref, 3e-5)


def test_conv_wgrad_stem():
    L = _lib()
    lib = L.lib()
    N, H, W = 2, 88, 200
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g, requires_grad=True)
    y = F.conv2d(x, w, None, 2, 3)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    ref = w.grad.permute(0, 2, 3, 1)
    x4 = torch.zeros(N, H, W, 4)
    x4[..., :3] = x.permute(0, 2, 3, 1)
    nsc = lib.cilrs_conv2d_wgrad_scratch_floats(N, H, W, 4, 64, 7, 7, 2, 3)
    scratch = torch.empty(nsc, device="cuda")
    dw = torch.full((64, 7, 7, 3), float("nan"), device="cuda")
    x4d, dyd = dev(x4), nhwc(dy)
    L.check(lib.cilrs_conv2d_wgrad(L.ptr(x4d), L.ptr(dyd), L.ptr(dw), L.ptr(scratch), N,
                                   H, W, 4, 64, 7, 7, 2, 3, 3, stream()))
    torch.cuda.synchronize()
    got = dw.cpu()
    assert (got - ref).abs().max() <= _tol(ref, 3e-5)


@pytest.mark.parametrize("M,Cc", [(3 * 22 * 50, 64), (5 * 275, 128), (2 * 78, 256), (21, 512),
                                  (8 * 4400, 64)])
@pytest.mark.parametrize("relu,res", [(1, False), (1, True), (0, False)])
def test_bn_train_fwd_bwd(M, Cc, relu, res):
    L = _lib()
    lib = L.lib()
    g = torch.Generator().manual_seed(6)
    y = (torch.randn(M, Cc, generator=g) * 1.7 + 0.3).requires_grad_(True)
    gamma = (torch.rand(Cc, generator=g) + 0.5).requires_grad_(True)
    beta = (torch.rand(Cc, generator=g) - 0.5).requires_grad_(True)
    rm = torch.rand(Cc, generator=g) - 0.5
    rv = torch.rand(Cc, generator=g) + 0.5
    resid = torch.randn(M, Cc, generator=g) if res else None
    rm_ref, rv_ref = rm.clone(), rv.clone()
    # torch reference on [M,C,1,1]
    out = F.batch_norm(y.view(M, Cc, 1, 1), rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    out = out.view(M, Cc)
    if res:
        out = out + resid
    if relu:
        out = F.relu(out)
    dz = torch.randn(M, Cc, generator=g)
    out.backward(dz)

    yd, gd, bd, rmd, rvd = dev(y.detach()), dev(gamma.detach()), dev(beta.detach()), dev(rm), dev(rv)
    nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
    stats = torch.empty(4 * Cc, device="cuda")
    part = torch.empty(lib.cilrs_bn_partial_floats(Cc), device="cuda")
    z = torch.empty(M, Cc, device="cuda")
    resd = dev(resid) if res else None
    L.check(lib.cilrs_bn_train_fwd(L.ptr(yd), M, Cc, L.ptr(gd), L.ptr(bd), L.ptr(rmd), L.ptr(rvd),
                                   L.ptr(nbt), 0.1, 1e-5, L.ptr(resd), relu, L.ptr(stats),
                                   L.ptr(part), L.ptr(z), stream()))
    torch.cuda.synchronize()
    assert (z.cpu() - out.detach()).abs().max() <= 2e-5 * max(1.0, float(out.abs().max()))
    assert (rmd.cpu() - rm_ref).abs().max() <= 1e-6
    assert (rvd.cpu() - rv_ref).abs().max() <= 1e-5
    assert int(nbt.item()) == 1

    dgamma = torch.empty(Cc, device="cuda")
    dbeta = torch.empty(Cc, device="cuda")
    coef = torch.empty(3 * Cc, device="cuda")
    dy = torch.empty(M, Cc, device="cuda")
    gout = torch.empty(M, Cc, device="cuda")
    dzd = dev(dz)
    L.check(lib.cilrs_bn_bwd(L.ptr(dzd), L.ptr(z), L.ptr(yd), M, Cc, L.ptr(gd), L.ptr(stats),
                             relu, L.ptr(dgamma), L.ptr(dbeta), L.ptr(coef), L.ptr(part),
                             L.ptr(dy), L.ptr(gout), stream()))
    torch.cuda.synchronize()
    sc = max(1.0, float(y.grad.abs().max()))
    assert (dy.cpu() - y.grad).abs().max() <= 5e-5 * sc
    gs = max(1.0, float(gamma.grad.abs().max()))
    assert (dgamma.cpu() - gamma.grad).abs().max() <= 5e-5 * gs
    assert (dbeta.cpu() - beta.grad).abs().max() <= 5e-5 * max(1.0, float(beta.grad.abs().max()))
    mask = (out.detach() > 0).float() if relu else torch.ones(M, Cc)
    assert (gout.cpu() - dz * mask).abs().max() == 0.0


def test_bn_eval_fwd():
    L = _lib()
    lib = L.lib()
    M, Cc = 1100, 64
    g = torch.Generator().manual_seed(7)
    y = torch.randn(M, Cc, generator=g)
    gamma, beta = torch.rand(Cc, generator=g) + 0.5, torch.rand(Cc, generator=g) - 0.5
    rm, rv = torch.rand(Cc, generator=g) - 0.5, torch.rand(Cc, generator=g) + 0.5
    ref = F.relu(F.batch_norm(y.view(M, Cc, 1, 1), rm, rv, gamma, beta, False, 0.1, 1e-5)).view(M, Cc)
    stats = torch.empty(4 * Cc, device="cuda")
    z = torch.empty(M, Cc, device="cuda")
    yd, gd, bd, rmd, rvd = dev(y), dev(gamma), dev(beta), dev(rm), dev(rv)
    L.check(lib.cilrs_bn_eval_fwd(L.ptr(yd), M, Cc, L.ptr(gd), L.ptr(bd),
                                  L.ptr(rmd), L.ptr(rvd), 1e-5, None, 1, L.ptr(stats),
                                  L.ptr(z), stream()))
    torch.cuda.synchronize()
    assert (z.cpu() - ref).abs().max() <= 1e-5


@pytest.mark.parametrize("N,H,W", [(2, 44, 100), (1, 7, 9), (3, 8, 8)])
def test_maxpool(N, H, W):
    L = _lib()
    lib = L.lib()
    Cc = 64
    g = torch.Generator().manual_seed(8)
    x = F.relu(torch.randn(N, Cc, H, W, generator=g)).requires_grad_(True)   # many exact-zero ties
    out = F.max_pool2d(x, 3, 2, 1)
    dout = torch.randn(out.shape, generator=g)
    out.backward(dout)
    Ho, Wo = out.shape[2], out.shape[3]
    xd = nhwc(x.detach())
    o = torch.empty(N, Ho, Wo, Cc, device="cuda")
    am = torch.empty(N, Ho, Wo, Cc, dtype=torch.uint8, device="cuda")
    L.check(lib.cilrs_maxpool_fwd(L.ptr(xd), L.ptr(o), L.ptr(am), N, H, W, Cc, stream()))
    dx = torch.empty(N, H, W, Cc, device="cuda")
    doutd = nhwc(dout)
    L.check(lib.cilrs_maxpool_bwd(L.ptr(doutd), L.ptr(am), L.ptr(dx), N, H, W, Cc, stream()))
    torch.cuda.synchronize()
    assert torch.equal(o.cpu().permute(0, 3, 1, 2), out.detach())
    # ties only occur at relu zeros, where the later ReLU mask kills the gradient anyway:
    # compare where the input is positive
    pos = (x.detach() > 0)
    got = dx.cpu().permute(0, 3, 1, 2)
    assert (got[pos] - x.grad[pos]).abs().max() <= 1e-6
    assert abs(float(got.sum()) - float(x.grad.sum())) <= 1e-2


# (batch, in, out): the heads' layer shapes plus ragged sizes (tiles are 32x32, k-groups of 8)
LINEAR_CASES = [(128, 640, 256), (5, 640, 256), (128, 512, 256), (33, 256, 256), (128, 256, 3),
                (7, 256, 1), (128, 1, 128), (3, 128, 128), (1, 640, 256), (70, 100, 45)]


@pytest.mark.parametrize("B,fin,fout", LINEAR_CASES)
@pytest.mark.parametrize("relu", [0, 1])
def test_linear_fwd_bwd(B, fin, fout, relu):
    """nn.Linear forward / backward as the heads launch it (grouped MFMA GEMMs, one group) vs
    torch fp32 on the CPU; x and dx live inside wider rows (pitch in + 4) like `combined`."""
    L = _lib()
    lib = L.lib()
    g = torch.Generator().manual_seed(B * 1000 + fin + fout + relu)
    ld = fin + 4
    xw = torch.randn(B, ld, generator=g)
    x = xw[:, :fin]
    w = torch.randn(fout, fin, generator=g) / max(1.0, fin ** 0.5)
    b = torch.randn(fout, generator=g)
    ref = x @ w.t() + b
    if relu:
        ref = ref.clamp_min(0)
    dxw, wd, bd = dev(xw), dev(w), dev(b)
    y = torch.empty(B, fout, device="cuda")
    L.check(lib.cilrs_linear_fwd(L.ptr(dxw), L.ptr(wd), L.ptr(bd), L.ptr(y), B, fin, fout, ld,
                                 fout, relu, stream()))
    assert (y.cpu() - ref).abs().max() <= _tol(ref)
    # backward: dy -> dw, db, and dx masked by an activation (the previous layer's ReLU) x 2
    dy = torch.randn(B, fout, generator=g)
    act = torch.randn(B, fin, generator=g)
    ref_dw = dy.t() @ x
    ref_db = dy.sum(0)
    ref_dx = (dy @ w) * (act > 0) * 2.0
    dxo = torch.full((B, ld), 7.0, device="cuda")
    dw = torch.empty(fout, fin, device="cuda")
    db = torch.empty(fout, device="cuda")
    dyd, actd = dev(dy), dev(act)                        # keep the device copies alive
    L.check(lib.cilrs_linear_bwd(L.ptr(dyd), L.ptr(dxw), L.ptr(wd), L.ptr(actd), 2.0,
                                 L.ptr(dxo), L.ptr(dw), L.ptr(db), B, fin, fout, fout, ld, ld, fin,
                                 stream()))
    assert (dw.cpu() - ref_dw).abs().max() <= _tol(ref_dw)
    assert (db.cpu() - ref_db).abs().max() <= _tol(ref_db)
    assert (dxo.cpu()[:, :fin] - ref_dx).abs().max() <= _tol(ref_dx)
    assert (dxo.cpu()[:, fin:] == 7.0).all()            # the row padding is untouched


# ---- round 2: the plans that the benchmark actually runs (N = 128) ------------------------------
# The ten distinct convolution shapes of the ResNet-34 trunk at the benchmark batch (BASELINE.json
# configs[1]: B = 128).  At this size the launchers pick other tiles, split-K factors and K-slab
# counts than at the small batches above (conv_wgrad<128> with grid.z slabs, split-K reduces that
# emit BatchNorm partials, five 64x64 blocks per CU, ...), so every op is checked here with the
# AUTO plan against torch's CPU convolution and its autograd.
TRUNK_SHAPES = [
    (22, 50, 64, 64, 3, 1, 1),       # layer1 3x3
    (22, 50, 64, 128, 3, 2, 1),      # layer2.0.conv1
    (22, 50, 64, 128, 1, 2, 0),      # layer2.0.downsample
    (11, 25, 128, 128, 3, 1, 1),     # layer2 3x3
    (11, 25, 128, 256, 3, 2, 1),     # layer3.0.conv1
    (11, 25, 128, 256, 1, 2, 0),     # layer3.0.downsample
    (6, 13, 256, 256, 3, 1, 1),      # layer3 3x3
    (6, 13, 256, 512, 3, 2, 1),      # layer4.0.conv1
    (6, 13, 256, 512, 1, 2, 0),      # layer4.0.downsample
    (3, 7, 512, 512, 3, 1, 1),       # layer4 3x3
]
# split-K scratch the plan hands every convolution at B = 128 (net.hip: 8 x the largest tensor of
# at most 6 M floats = layer2's 35,200 x 128 output) -- the cost model's choices depend on it
PLAN_KSPLIT_FLOATS = 8 * 128 * 11 * 25 * 128


@pytest.mark.parametrize("shape", TRUNK_SHAPES)
def test_conv_auto_plan_at_benchmark_batch(shape):
    L = _lib()
    lib = L.lib()
    N = 128
    H, W, Cin, Cout, k, s, p = shape
    from cilrs_mi355.hostinfo import usable_cores
    torch.set_num_threads(usable_cores())
    g = torch.Generator().manual_seed(100 + Cin + Cout + k)
    x = torch.randn(N, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / (k * k * Cin) ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, s, p)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    Ho, Wo = y.shape[2], y.shape[3]
    xd, wd, dyd = nhwc(x.detach()), ohwi(w.detach()), nhwc(dy)
    scratch = torch.empty(PLAN_KSPLIT_FLOATS, device="cuda")
    # forward
    yd = torch.full((N, Ho, Wo, Cout), float("nan"), device="cuda")
    L.check(lib.cilrs_conv2d_fwd(L.ptr(xd), L.ptr(wd), L.ptr(yd), N, H, W, Cin, Cout, k, k, s, p,
                                 -1, 0, L.ptr(scratch), scratch.numel(), stream()))
    # data gradient, plain and with the residual gradient added in the epilogue
    dxd = torch.full((N, H, W, Cin), float("nan"), device="cuda")
    L.check(lib.cilrs_conv2d_dgrad(L.ptr(dyd), L.ptr(wd), L.ptr(dxd), None, N, H, W, Cin, Cout, k,
                                   k, s, p, -1, 0, L.ptr(scratch), scratch.numel(), stream()))
    add = torch.randn(N, H, W, Cin, generator=g)
    addd = dev(add)
    dxa = torch.full((N, H, W, Cin), float("nan"), device="cuda")
    L.check(lib.cilrs_conv2d_dgrad(L.ptr(dyd), L.ptr(wd), L.ptr(dxa), L.ptr(addd), N, H, W, Cin,
                                   Cout, k, k, s, p, -1, 0, L.ptr(scratch), scratch.numel(),
                                   stream()))
    # weight gradient
    nsc = lib.cilrs_conv2d_wgrad_scratch_floats(N, H, W, Cin, Cout, k, k, s, p)
    slabs = torch.empty(nsc, device="cuda")
    dwd = torch.full((Cout, k, k, Cin), float("nan"), device="cuda")
    L.check(lib.cilrs_conv2d_wgrad(L.ptr(xd), L.ptr(dyd), L.ptr(dwd), L.ptr(slabs), N, H, W, Cin,
                                   Cout, k, k, s, p, Cin, stream()))
    torch.cuda.synchronize()
    got_y = yd.cpu().permute(0, 3, 1, 2)
    ref_y = y.detach()
    assert torch.isfinite(got_y).all()
    assert (got_y - ref_y).abs().max() <= _tol(ref_y)
    want_dx = x.grad.permute(0, 2, 3, 1)
    got_dx = dxd.cpu()
    assert torch.isfinite(got_dx).all()
    assert (got_dx - want_dx).abs().max() <= _tol(want_dx)
    assert (dxa.cpu() - (want_dx + add)).abs().max() <= _tol(want_dx + add)
    # K = N*Ho*Wo products per element (up to 140,800): both fp32 results carry summation-order
    # noise ~ sqrt(K) ulp, so the weight gradient is budgeted against a float64 reference
    want_dw = w.grad.permute(0, 2, 3, 1)
    got_dw = dwd.cpu()
    assert torch.isfinite(got_dw).all()
    err = (got_dw - want_dw).abs().max()
    assert err <= _tol(want_dw, 5e-5), float(err / want_dw.abs().max())
    _wgrad_vs_f64(x.detach(), dy, (Cout, Cin, k, k), s, p, got_dw, want_dw)


def _wgrad_vs_f64(x, dy, wshape, stride, pad, got_ohwi, cpu32_ohwi):
    """relative-L2 error of the HIP weight gradient against a float64 computation; it may not
    exceed twice the error torch's own fp32 CPU path makes (floor 1e-6)."""
    w64 = torch.zeros(wshape, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.double(), w64, None, stride, pad).backward(dy.double())
    ref = w64.grad.permute(0, 2, 3, 1)
    nrm = float(ref.norm())
    e_hip = float((got_ohwi.double() - ref).norm()) / nrm
    e_cpu = float((cpu32_ohwi.double() - ref).norm()) / nrm
    assert e_hip <= max(2.0 * e_cpu, 1e-6), (e_hip, e_cpu)


def test_stem_auto_plan_at_benchmark_batch():
    """conv 7x7 / s2 / p3 with Cin padded 3 -> 4 at N = 128: forward and weight gradient."""
    L = _lib()
    lib = L.lib()
    N, H, W = 128, 88, 200
    g = torch.Generator().manual_seed(77)
    x = torch.randn(N, 3, H, W, generator=g)
    w = (torch.randn(64, 3, 7, 7, generator=g) / 147 ** 0.5).requires_grad_(True)
    y = F.conv2d(x, w, None, 2, 3)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy)
    x4 = torch.zeros(N, H, W, 4)
    x4[..., :3] = x.permute(0, 2, 3, 1)
    w4 = torch.zeros(64, 7, 7, 4)
    w4[..., :3] = w.detach().permute(0, 2, 3, 1)
    x4d, w4d, dyd = dev(x4), dev(w4), nhwc(dy)
    scratch = torch.empty(PLAN_KSPLIT_FLOATS, device="cuda")
    yd = torch.full((N, 44, 100, 64), float("nan"), device="cuda")
    L.check(lib.cilrs_conv2d_fwd(L.ptr(x4d), L.ptr(w4d), L.ptr(yd), N, H, W, 4, 64, 7, 7, 2, 3, -1,
                                 0, L.ptr(scratch), scratch.numel(), stream()))
    nsc = lib.cilrs_conv2d_wgrad_scratch_floats(N, H, W, 4, 64, 7, 7, 2, 3)
    slabs = torch.empty(nsc, device="cuda")
    dwd = torch.full((64, 7, 7, 3), float("nan"), device="cuda")
    L.check(lib.cilrs_conv2d_wgrad(L.ptr(x4d), L.ptr(dyd), L.ptr(dwd), L.ptr(slabs), N, H, W, 4,
                                   64, 7, 7, 2, 3, 3, stream()))
    torch.cuda.synchronize()
    ref = y.detach()
    assert (yd.cpu().permute(0, 3, 1, 2) - ref).abs().max() <= _tol(ref)
    want_dw = w.grad.permute(0, 2, 3, 1)
    got = dwd.cpu()
    assert torch.isfinite(got).all()
    assert (got - want_dw).abs().max() <= _tol(want_dw, 5e-5)
    _wgrad_vs_f64(x, dy, (64, 3, 7, 7), 2, 3, got, want_dw)


@pytest.mark.parametrize("clip,gscale", [(False, 1.0), (True, 0.5)])
def test_adam_step_matches_torch_adam(clip, gscale):
    """cilrs_adam_step vs torch.optim.Adam (coupled L2 weight decay, bias correction, eps outside
    the sqrt -- notebook/notebook.ipynb:533-534, 555) element by element over five steps with
    fresh gradients each step, so the first/second-moment history, the bias corrections and the
    weight decay all matter (step 1 alone is lr*sign(g) whatever the hyper-parameters are).  With
    clip: the device coefficient of cilrs_grad_sqnorm scales the gradient, and grad_scale (the
    1/world_size of data-parallel sums) multiplies it too."""
    L = _lib()
    lib = L.lib()
    n = 100_003 * 4            # the arenas are float4-granular
    lr, b1, b2, eps, wd, max_norm = 2e-4, 0.9, 0.999, 1e-8, 1e-4, 1.0
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(n, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=lr, betas=(b1, b2), eps=eps, weight_decay=wd)
    pd = p0.cuda()
    m = torch.zeros(n, device="cuda")
    v = torch.zeros(n, device="cuda")
    clip_out = torch.zeros(2, device="cuda")
    scr = torch.empty(lib.cilrs_sqnorm_scratch_bytes(), dtype=torch.uint8, device="cuda")
    for step in range(1, 6):
        # gradients of very different magnitudes, some exactly zero, rescaled every step
        gr = torch.randn(n, generator=g) * (10.0 ** torch.randint(-6, 1, (n,), generator=g).float())
        gr[::97] = 0.0
        gr *= 3.0 / step
        gd = gr.cuda()
        clip_ptr = None
        g_eff = gr * gscale
        if clip:
            # clip acts on the gradient the optimiser sees (already averaged over ranks)
            L.check(lib.cilrs_scale(L.ptr(gd), n, None, gscale, stream()))
            L.check(lib.cilrs_grad_sqnorm(L.ptr(gd), n, max_norm, L.ptr(scr), L.ptr(clip_out),
                                          stream()))
            clip_ptr = L.ptr(clip_out)
            p_ref.grad = g_eff.clone()
            tn = torch.nn.utils.clip_grad_norm_([p_ref], max_norm)
            L.check(lib.cilrs_adam_step(L.ptr(pd), L.ptr(gd), L.ptr(m), L.ptr(v), n, lr, b1, b2,
                                        eps, wd, step, clip_ptr, 1.0, stream()))
            torch.cuda.synchronize()
            assert abs(float(clip_out[0]) - float(tn)) <= 1e-4 * float(tn)    # fp32 vs double sums
            assert abs(float(clip_out[1]) - min(1.0, max_norm / (float(tn) + 1e-6))) <= 1e-6
            # element-wise comparison with the coefficient the kernel itself used: where g*coef
            # nearly cancels wd*p, the 1e-5 relative difference between the two norms above would
            # otherwise decide the sign of a ~1e-9 gradient, i.e. a full +-lr step
            p_ref.grad = g_eff * float(clip_out[1])
        else:
            p_ref.grad = g_eff.clone()
            L.check(lib.cilrs_adam_step(L.ptr(pd), L.ptr(gd), L.ptr(m), L.ptr(v), n, lr, b1, b2,
                                        eps, wd, step, None, gscale, stream()))
        opt.step()
        torch.cuda.synchronize()
        st = opt.state[p_ref]
        assert (pd.cpu() - p_ref.detach()).abs().max() <= 1e-6, step
        assert (m.cpu() - st["exp_avg"]).abs().max() <= 1e-6 * max(1.0, float(st["exp_avg"].abs().max()))
        assert (v.cpu() - st["exp_avg_sq"]).abs().max() <= 1e-6 * max(1.0, float(st["exp_avg_sq"].abs().max()))
    # and the trajectory really moved by more than the tolerance
    assert (p_ref.detach() - p0).abs().max() >= 4 * lr


# ---- the 16-bit matrix pipe (bf16 training mode), op by op ---------------------------------------
# Operands are rounded to bf16 / fp16 by the entry points; the reference is the SAME product of the
# rounded operands in float64 (so only the accumulation order differs: fp32 MFMA accumulation over
# up to 4,608 terms -> 2e-5 of max|ref| like the fp32 kernels, relaxed to 5e-5).
CONV16_CASES = [  # N, H, W, Cin, Cout, K, stride, pad
    (3, 9, 14, 64, 64, 3, 1, 1),
    (2, 12, 10, 64, 128, 3, 2, 1),      # stride-2 3x3 (BasicBlock / Bottleneck down-sampling conv)
    (2, 11, 13, 128, 64, 1, 1, 0),      # Bottleneck 1x1
    (2, 11, 13, 64, 256, 1, 2, 0),      # 1x1 stride-2 down-sample branch, odd size
    (1, 6, 13, 256, 256, 3, 1, 1),
    (5, 7, 7, 64, 64, 3, 2, 1),         # odd size, stride 2
    (2, 12, 10, 128, 128, 3, 2, 1),     # 128-wide weight-gradient tile, stride 2
    (3, 5, 9, 256, 128, 1, 1, 0),       # 128-wide tile, 1x1, ragged pixel count
]


def _round16(t, bf16):
    return t.to(torch.bfloat16 if bf16 else torch.float16).double()


@pytest.mark.parametrize("case", CONV16_CASES)
@pytest.mark.parametrize("bf16", [1, 0])
def test_conv16_fwd_dgrad_wgrad(case, bf16):
    L = _lib()
    lib = L.lib()
    N, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / (k * k * Cin) ** 0.5
    xr = _round16(x, bf16).requires_grad_(True)
    wr = _round16(w, bf16).requires_grad_(True)
    ref = F.conv2d(xr, wr, None, s, p)
    Ho, Wo = ref.shape[2], ref.shape[3]
    dy = torch.randn(N, Cout, Ho, Wo, generator=g)
    dyr = _round16(dy, bf16)
    gx, gw = torch.autograd.grad(ref, (xr, wr), dyr)
    n16 = lib.cilrs_conv2d_16_scratch_halfs(N, H, W, Cin, Cout, k, s, p)
    s16 = torch.zeros(n16, dtype=torch.int16, device="cuda")
    xd, wd, dyd = nhwc(x), ohwi(w), nhwc(dy)
    M = N * Ho * Wo
    # forward (+ BatchNorm column partials)
    y = torch.full((N, Ho, Wo, Cout), float("nan"), device="cuda")
    nmt = (M + 63) // 64
    part = torch.full((2 * Cout * nmt,), float("nan"), device="cuda")
    L.check(lib.cilrs_conv2d_fwd_16(L.ptr(xd), L.ptr(wd), L.ptr(y), L.ptr(part), N, H, W, Cin,
                                    Cout, k, s, p, bf16, L.ptr(s16), stream()))
    torch.cuda.synchronize()
    got = y.cpu().permute(0, 3, 1, 2).double()
    tol = 5e-5 * float(ref.detach().abs().max())
    assert torch.isfinite(got).all()
    assert (got - ref.detach()).abs().max() <= tol
    pp = part.cpu().double().view(2, Cout, nmt).sum(-1)
    flat = ref.detach().permute(0, 2, 3, 1).reshape(M, Cout)
    assert (pp[0] - flat.sum(0)).abs().max() <= 1e-4 * max(1.0, float(flat.sum(0).abs().max()))
    assert (pp[1] - (flat ** 2).sum(0)).abs().max() <= 1e-4 * float((flat ** 2).sum(0).max())
    # data gradient, with and without the fp32 addend
    for with_add in (False, True):
        add = torch.randn(N, H, W, Cin, generator=g) if with_add else None
        addd = dev(add) if with_add else None
        dx = torch.full((N, H, W, Cin), float("nan"), device="cuda")
        L.check(lib.cilrs_conv2d_dgrad_16(L.ptr(dyd), L.ptr(wd), L.ptr(dx), L.ptr(addd), N, H, W,
                                          Cin, Cout, k, s, p, bf16, L.ptr(s16), stream()))
        torch.cuda.synchronize()
        want = gx.permute(0, 2, 3, 1) + (add.double() if with_add else 0.0)
        assert torch.isfinite(dx).all()
        assert (dx.cpu().double() - want).abs().max() <= 5e-5 * float(want.abs().max()), with_add
    # weight gradient
    nsl = lib.cilrs_conv2d_wgrad_16_scratch_floats(N, H, W, Cin, Cout, k, s, p)
    sl = torch.empty(max(nsl, 4), device="cuda")
    dw = torch.full((Cout, k, k, Cin), float("nan"), device="cuda")
    L.check(lib.cilrs_conv2d_wgrad_16(L.ptr(xd), L.ptr(dyd), L.ptr(dw), L.ptr(sl), N, H, W, Cin,
                                      Cout, k, s, p, bf16, L.ptr(s16), stream()))
    torch.cuda.synchronize()
    want = gw.permute(0, 2, 3, 1)
    assert torch.isfinite(dw).all()
    assert (dw.cpu().double() - want).abs().max() <= 5e-5 * float(want.abs().max())


def test_conv_wino_channel_split_of_an_underfilled_launch():
    """layer3 at the benchmark batch: 168 64-tile x 64-channel blocks on 256 CUs.  The train step's
    launch plan cuts the reduction over the 256 input channels into three parts (504 blocks, two
    rounds of a third of the work), every part writes a slab, and a fixed-order reduce sums the
    slabs, adds the addend and emits the BatchNorm column partials.  Same contract as the one-part
    launch: 5e-5 of max|ref| against torch's direct convolution; partials = column sums of the
    result."""
    L = _lib()
    lib = L.lib()
    N, H, W, Cc = 128, 6, 13, 256
    g = torch.Generator().manual_seed(9)
    x = torch.randn(N, Cc, H, W, generator=g)
    w = torch.randn(Cc, Cc, 3, 3, generator=g) / (9 * Cc) ** 0.5
    add = torch.randn(N, H, W, Cc, generator=g)
    torch.set_num_threads(16)
    ref = F.conv2d(x, w, None, 1, 1).permute(0, 2, 3, 1) + add
    xd, wd, addd = nhwc(x), ohwi(w), dev(add)
    U = torch.empty(lib.cilrs_conv2d_wino_scratch_floats(Cc, Cc), device="cuda")
    L.check(lib.cilrs_wino_filter_transform(L.ptr(wd), L.ptr(U), Cc, Cc, 0, stream()))
    M = N * H * W
    slabs = torch.full((4 * M * Cc,), float("nan"), device="cuda")
    part = torch.full((2 * Cc * 1024,), float("nan"), device="cuda")
    y = torch.full((N, H, W, Cc), float("nan"), device="cuda")
    cs, rows = C.c_int(0), C.c_int(0)
    L.check(lib.cilrs_conv2d_wino_split(L.ptr(xd), L.ptr(U), L.ptr(y), L.ptr(addd), L.ptr(part), N, H,
                                        W, Cc, Cc, L.ptr(slabs), slabs.numel(), C.byref(cs),
                                        C.byref(rows), stream()))
    torch.cuda.synchronize()
    assert cs.value == 3, f"the launch plan did not split this launch in three ({cs.value})"
    got = y.cpu()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max()) <= _tol(ref, 5e-5)
    pp = part[:2 * Cc * rows.value].cpu().double().view(2, Cc, rows.value).sum(-1)
    flat = got.double().reshape(M, Cc)
    assert (pp[0] - flat.sum(0)).abs().max() <= 1e-4 * max(1.0, float(flat.sum(0).abs().max()))
    assert (pp[1] - (flat ** 2).sum(0)).abs().max() <= 1e-4 * float((flat ** 2).sum(0).max())
    # without scratch the same call runs the one-part launch: identical contract, bitwise its own
    y2 = torch.full((N, H, W, Cc), float("nan"), device="cuda")
    L.check(lib.cilrs_conv2d_wino_pre(L.ptr(xd), L.ptr(U), L.ptr(y2), L.ptr(addd), N, H, W, Cc, Cc,
                                      stream()))
    torch.cuda.synchronize()
    assert float((y2.cpu() - ref).abs().max()) <= _tol(ref, 5e-5)
    assert float((y2.cpu() - got).abs().max()) <= _tol(ref, 2e-5)     # (another summation order)


# ---- the bf16 training mode's operators on 16-bit tensors (round 4) ------------------------------
CONV16T_CASES = [
    # N, H, W, Cin, Cout, k, s, p                 tile the launch plan picks on a 256-CU device
    (2, 22, 50, 64, 64, 3, 1, 1),                 # 64x64 (small layer)
    (16, 44, 100, 64, 128, 3, 1, 1),              # 128x128, nine K-tiles per tile
    (16, 44, 100, 64, 64, 1, 1, 0),               # 128x64, ONE K-tile per tile (all epilogue)
    (33, 44, 50, 256, 128, 1, 1, 0),              # 128x128, ragged last tile (M % 128 = 24)
    (64, 44, 100, 64, 128, 3, 2, 1),              # 128x128, stride 2
    (5, 6, 13, 256, 256, 3, 1, 1),                # 64x64, 36 K-tiles
]


def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("case", CONV16T_CASES)
def test_conv16_train_ops_on_16bit_tensors(case):
    """cilrs_conv2d_train_16 (both tile families): bf16 in, result ROUNDED to bf16 (+ bf16 addend),
    BatchNorm statistics and BatchNorm-backward reductions taken from the STORED values.  Against
    torch's fp32 CPU convolution of the same bf16 operands: one bf16 ulp (2^-8 relative; the two
    fp32 sums differ in order, which can move a result across a rounding boundary); the column
    partials against sums of what the kernel stored: 1e-4."""
    L = _lib()
    lib = L.lib()
    N, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(23)
    x = _bf(torch.randn(N, Cin, H, W, generator=g))
    w = _bf(torch.randn(Cout, Cin, k, k, generator=g) / (k * k * Cin) ** 0.5)
    torch.set_num_threads(16)
    ref = F.conv2d(x.float(), w.float(), None, s, p)
    Ho, Wo = ref.shape[2], ref.shape[3]
    M = N * Ho * Wo
    ref = ref.permute(0, 2, 3, 1).reshape(M, Cout)
    xd = x.permute(0, 2, 3, 1).contiguous().cuda()
    wd = w.permute(0, 2, 3, 1).contiguous().cuda()
    rows = C.c_int(0)

    def run(y16, y32, add, part, bz, by, bstats, brelu, bpart):
        L.check(lib.cilrs_conv2d_train_16(L.ptr(xd), L.ptr(wd), L.ptr(y16), L.ptr(y32), L.ptr(add),
                                          L.ptr(part), L.ptr(bz), L.ptr(by), L.ptr(bstats), brelu,
                                          L.ptr(bpart), N, H, W, Cin, Ho, Wo, Cout, k, s, p, 0, 1,
                                          C.byref(rows), stream()))
        torch.cuda.synchronize()

    ulp = 2.0 ** -8
    # (1) forward form: rounded result + BatchNorm statistics of the stored tensor
    y = torch.full((M, Cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    part = torch.full((2 * Cout * ((M + 63) // 64),), float("nan"), device="cuda")
    run(y, None, None, part, None, None, None, 0, None)
    got = y.float().cpu()
    assert torch.isfinite(got).all()
    assert ((got - ref).abs() <= ulp * ref.abs() + 1e-6 * float(ref.abs().max())).all()
    nr = rows.value
    assert nr in ((M + 63) // 64, (M + 127) // 128)
    pp = part[:2 * Cout * nr].cpu().double().view(2, Cout, nr).sum(-1)
    s1, s2 = got.double().sum(0), (got.double() ** 2).sum(0)
    assert (pp[0] - s1).abs().max() <= 1e-4 * max(1.0, float(s1.abs().max()))
    assert (pp[1] - s2).abs().max() <= 1e-4 * float(s2.max())
    # (2) data-gradient form: + bf16 addend, BatchNorm-backward reductions of the stored gradient
    add = _bf(torch.randn(M, Cout, generator=g))
    bz = _bf(torch.randn(M, Cout, generator=g))
    by = _bf(torch.randn(M, Cout, generator=g))
    bstats = torch.cat([torch.randn(Cout, generator=g) * 0.1, torch.rand(Cout, generator=g) + 0.5])
    bpart = torch.full((2 * Cout * ((M + 63) // 64),), float("nan"), device="cuda")
    y.fill_(float("nan"))
    run(y, None, add.cuda(), None, bz.cuda(), by.cuda(), bstats.cuda(), 1, bpart)
    got = y.float().cpu()
    want = ref + add.float()
    assert ((got - want).abs() <= ulp * want.abs() + 1e-6 * float(want.abs().max())).all()
    gm = torch.where(bz.float() > 0, got, torch.zeros_like(got)).double()
    xh = (by.double() - bstats[:Cout].double()) * bstats[Cout:].double()
    nr = rows.value
    pp = bpart[:2 * Cout * nr].cpu().double().view(2, Cout, nr).sum(-1)
    t1, t2 = gm.sum(0), (gm * xh).sum(0)
    scale = max(1.0, float(gm.abs().sum(0).max()))
    assert (pp[0] - t1).abs().max() <= 1e-4 * scale
    assert (pp[1] - t2).abs().max() <= 1e-4 * max(1.0, float((gm * xh).abs().sum(0).max()))
    # (3) fp32 result with a bf16 addend (the gradient handed to the fp32 stem)
    y32 = torch.full((M, Cout), float("nan"), device="cuda")
    run(None, y32, add.cuda(), None, None, None, None, 0, None)
    assert (y32.cpu() - want).abs().max() <= 5e-5 * float(want.abs().max())


@pytest.mark.parametrize("M,Cch", [(2 * 22 * 50, 64), (8 * 11 * 25, 128), (3 * 6 * 13, 256),
                                   (77, 2048), (16 * 44 * 100, 256)])
def test_bn16_fwd_bwd_on_bf16_tensors(M, Cch):
    """BatchNorm2d (training) on bf16 NHWC tensors against torch's fp32 BatchNorm of the SAME bf16
    tensor: outputs / input gradients within one bf16 ulp, statistics / parameter gradients 1e-4,
    running statistics 1e-5, the residual path's masked gradient exact."""
    L = _lib()
    lib = L.lib()
    g = torch.Generator().manual_seed(7)
    y = _bf(torch.randn(M, Cch, generator=g) * 1.5 + 0.3)
    res = _bf(torch.randn(M, Cch, generator=g))
    gamma = torch.rand(Cch, generator=g) + 0.5
    beta = torch.randn(Cch, generator=g) * 0.2
    rm0, rv0 = torch.randn(Cch, generator=g) * 0.1, torch.rand(Cch, generator=g) + 0.5
    dz = _bf(torch.randn(M, Cch, generator=g))
    yf = y.float().requires_grad_(True)
    gp, bp = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = rm0.clone(), rv0.clone()
    t = F.batch_norm(yf.t().reshape(1, Cch, M, 1), rm, rv, gp, bp, True, 0.1, 1e-5)
    t = t.reshape(Cch, M).t()
    zref = F.relu(t + res.float())
    zq = _bf(zref).float()
    # backward as the mode defines it: the mask comes from the STORED z, dz is a stored bf16 tensor
    gmask = torch.where(zq > 0, dz.float(), torch.zeros(()))
    (t * gmask.detach()).sum().backward()
    yd, resd, dzd = y.cuda(), res.cuda(), dz.cuda()
    stats = torch.full((4 * Cch,), float("nan"), device="cuda")
    partial = torch.empty(lib.cilrs_bn_partial_floats(Cch), device="cuda")
    z = torch.full((M, Cch), float("nan"), dtype=torch.bfloat16, device="cuda")
    rmd, rvd = rm0.cuda(), rv0.cuda()
    nbt = torch.zeros(1, dtype=torch.int64, device="cuda")
    gd, bd = gamma.cuda(), beta.cuda()
    L.check(lib.cilrs_bn16_train_fwd(L.ptr(yd), M, Cch, L.ptr(gd), L.ptr(bd), L.ptr(rmd), L.ptr(rvd),
                                     L.ptr(nbt), 0.1, 1e-5, L.ptr(resd), 1, L.ptr(stats),
                                     L.ptr(partial), L.ptr(z), 0, stream()))
    torch.cuda.synchronize()
    ulp = 2.0 ** -8
    zg = z.float().cpu()
    assert ((zg - zref.detach()).abs() <= ulp * zref.detach().abs() + 1e-5).all()   # (fp32 BatchNorm sums of 70,400 rows differ in order)
    assert (rmd.cpu() - rm).abs().max() <= 1e-5 and (rvd.cpu() - rv).abs().max() <= 1e-5
    assert int(nbt) == 1
    # backward on the kernel's own z (the mask must be the stored one)
    dy = torch.full((M, Cch), float("nan"), dtype=torch.bfloat16, device="cuda")
    gout = torch.full((M, Cch), float("nan"), dtype=torch.bfloat16, device="cuda")
    dgam, dbet = torch.empty(Cch, device="cuda"), torch.empty(Cch, device="cuda")
    coef = torch.empty(3 * Cch, device="cuda")
    L.check(lib.cilrs_bn16_bwd(L.ptr(dzd), L.ptr(z), L.ptr(yd), M, Cch, L.ptr(gd), L.ptr(stats), 1,
                               L.ptr(dgam), L.ptr(dbet), L.ptr(coef), L.ptr(partial), L.ptr(dy),
                               L.ptr(gout), 0, stream()))
    torch.cuda.synchronize()
    same_mask = (zg > 0) == (zq > 0)
    assert same_mask.float().mean() > 0.9999       # (an element at rounding distance from zero may flip)
    if bool(same_mask.all()):
        assert torch.equal(gout.float().cpu(), gmask)
        want = yf.grad
        assert ((dy.float().cpu() - want).abs() <= ulp * want.abs() + 1e-5 * float(want.abs().max())).all()
        assert (dgam.cpu() - gp.grad).abs().max() <= 1e-4 * max(1.0, float(gp.grad.abs().max()))
        assert (dbet.cpu() - bp.grad).abs().max() <= 1e-4 * max(1.0, float(bp.grad.abs().max()))


# ---- Winograd F(2x2, 3x3) forms of the 3x3 / stride 1 / pad 1 convolution ------------------------
WINO_CASES = [
    (3, 22, 50, 64, 64),       # layer1
    (3, 11, 25, 128, 128),     # layer2 (odd height and width: half-filled edge tiles)
    (5, 6, 13, 256, 256),      # layer3
    (7, 3, 7, 512, 512),       # layer4
    (1, 5, 3, 8, 64),          # fewer tiles than a block, one reduction chunk
    (128, 22, 50, 64, 64),     # the benchmark batch, layer1
    (128, 11, 25, 128, 128),   # the benchmark batch, layer2
]


@pytest.mark.parametrize("case", WINO_CASES)
def test_conv_wino_fwd_and_dgrad(case):
    """Winograd F(2x2,3x3) against torch's fp32 CPU convolution and its autograd.  fp32 Winograd
    rounds at other points than the direct sum (filter transform with halves, 4x4 transform-domain
    sums): tolerance 5e-5 of max|ref| (the direct kernels sit at 2e-5; the contract is 1e-4)."""
    L = _lib()
    lib = L.lib()
    N, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (9 * Cin) ** 0.5
    x.requires_grad_(True)
    torch.set_num_threads(16)
    ref = F.conv2d(x, w, None, 1, 1)
    dy = torch.randn(ref.shape, generator=g)
    ref.backward(dy)
    add = torch.randn(N, Cin, H, W, generator=g)
    xd, wd = nhwc(x.detach()), ohwi(w)
    y = torch.full((N, H, W, Cout), float("nan"), device="cuda")
    scratch = torch.empty(lib.cilrs_conv2d_wino_scratch_floats(Cin, Cout), device="cuda")
    L.check(lib.cilrs_conv2d_wino_fwd(L.ptr(xd), L.ptr(wd), L.ptr(y), N, H, W, Cin, Cout,
                                      L.ptr(scratch), stream()))
    torch.cuda.synchronize()
    got = y.cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all()
    err = float((got - ref.detach()).abs().max())
    assert err <= _tol(ref.detach(), 5e-5), err
    if Cin % 64 == 0:
        dx = torch.full((N, H, W, Cin), float("nan"), device="cuda")
        dyd, addd = nhwc(dy), nhwc(add)              # (kept alive across the launch)
        L.check(lib.cilrs_conv2d_wino_dgrad(L.ptr(dyd), L.ptr(wd), L.ptr(dx), L.ptr(addd),
                                            N, H, W, Cin, Cout, L.ptr(scratch), stream()))
        torch.cuda.synchronize()
        gdx = dx.cpu().permute(0, 3, 1, 2)
        want = x.grad + add
        assert torch.isfinite(gdx).all()
        assert float((gdx - want).abs().max()) <= _tol(want, 5e-5)
        # weight gradient in the Winograd domain against a float64 computation of the same sum:
        # error <= 2x torch-fp32's own (the reduction is N x H x W long: up to 140,800 terms)
        x64, w64 = x.detach().double().requires_grad_(False), w.double().requires_grad_(True)
        F.conv2d(x64, w64, None, 1, 1).backward(dy.double())
        w32 = w.clone().requires_grad_(True)
        F.conv2d(x.detach(), w32, None, 1, 1).backward(dy)
        nsc = lib.cilrs_conv2d_wino_wgrad_scratch_floats(N, H, W, Cin, Cout)
        sc2 = torch.empty(nsc, device="cuda")
        dw = torch.full((Cout, 3, 3, Cin), float("nan"), device="cuda")
        L.check(lib.cilrs_conv2d_wino_wgrad(L.ptr(xd), L.ptr(dyd), L.ptr(dw), N, H, W, Cin, Cout,
                                            L.ptr(sc2), nsc, stream()))
        torch.cuda.synchronize()
        gdw = dw.cpu().permute(0, 3, 1, 2).double()
        assert torch.isfinite(gdw).all()
        e_hip = float((gdw - w64.grad).abs().max())
        e_cpu = float((w32.grad.double() - w64.grad).abs().max())
        assert e_hip <= max(2.0 * e_cpu, 2e-5 * float(w64.grad.abs().max())), (e_hip, e_cpu)
