"""GPU parity tests: every HIP kernel family, called through the C ABI (u2seg_amd._hip / layers.functional), against
the CPU oracle on the same seeded inputs.  Integer / index outputs must be bit-exact; floating-point outputs are
compared at the tolerance written next to each check (bf16 storage => 2^-8 relative rounding per stored value).

Run on the GPU box with:  python -m pytest tests -m gpu -q
"""
import itertools
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

from oracle import ops as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "configs", "COCO-PanopticSegmentation", "u2seg_R50_800.yaml")
DEV = "cuda:0"


@pytest.fixture(scope="module")
def F():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from u2seg_amd import _hip
    from u2seg_amd.layers import functional

    _hip.load()  # fails loudly if libu2seg_hip.so is absent
    return functional


@pytest.fixture(scope="module")
def G():
    return np.load(os.path.join(ROOT, "tests", "golden", "ops_golden.npz"))


def bf(x):
    return x.bfloat16().float()


def nhwc(x_nchw, cp=None):
    b, c, h, w = x_nchw.shape
    cp = cp or (c + 31) // 32 * 32
    out = torch.zeros((b, h, w, cp), dtype=torch.bfloat16, device=DEV)
    out[..., :c] = x_nchw.permute(0, 2, 3, 1).to(DEV)
    return out


def nchw(x_nhwc, c=None):
    c = c or x_nhwc.shape[3]
    return x_nhwc[..., :c].permute(0, 3, 1, 2).float().cpu()


def rel_err(a, b):
    """max |a - b| over max |b|.  Round 4: the kernel-level bounds below are 4e-3 ... 5e-3 - one bf16 step (2^-8) of the largest
    element, i.e. one rounding of the result plus accumulation-order noise; measured values are 2e-3 ... 3.5e-3 (scratch probe of
    every call site), the two looser bounds carry their reason next to the assert."""
    return float((a.detach() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-12))


# -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,k,stride,pad,bias,relu", [
    (64, 64, 1, 1, 0, False, False), (64, 256, 3, 1, 1, True, True), (128, 96, 3, 2, 1, False, False),
    (256, 40, 1, 2, 0, True, False), (32, 28, 1, 1, 0, True, False), (96, 64, 3, 1, 1, False, True),
])
def test_conv_fwd_bwd(F, cin, cout, k, stride, pad, bias, relu):
    """conv (+bias)(+relu) forward, dgrad, wgrad, bias grad vs F.conv2d on the same bf16-rounded operands.
    Tolerance: 4e-3 of the output range against the UNROUNDED fp32 reference (the HIP result is the correctly rounded fp32 sum
    up to accumulation-order noise: half a bf16 step, 2^-9 of the element); fp32 weight / bias gradients 2e-3."""
    g = torch.Generator().manual_seed(cin * 7 + cout)
    x = bf(torch.randn((2, cin, 19, 23), generator=g))
    w = (torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5).requires_grad_(True)
    b = (torch.randn(cout, generator=g) * 0.1).requires_grad_(True) if bias else None
    xr = x.clone().requires_grad_(True)
    yr = TF.conv2d(xr, bf(w), bf(b) if bias else None, stride, pad)  # autocast rounds the bias like the other operands
    if relu:
        yr = TF.relu(yr)
    gy = bf(torch.randn(yr.shape, generator=g))
    yr.backward(gy)
    xd = nhwc(x).requires_grad_(True)
    wd = w.detach().to(DEV).requires_grad_(True)
    bd = b.detach().to(DEV).requires_grad_(True) if bias else None
    y, stats = F._Conv2dFn.apply(xd, wd, bd, stride, pad, relu, not relu and not bias)
    assert rel_err(nchw(y, cout), yr.detach()) < 4e-3
    if stats is not None:  # BN statistics of the stored bf16 output
        yy = nchw(y, cout)
        assert torch.allclose(stats[0].cpu(), yy.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
        assert torch.allclose(stats[1].cpu(), (yy * yy).sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    if y.shape[3] != cout:
        assert float(y[..., cout:].abs().max()) == 0.0  # pad columns stay zero
    y.backward(nhwc(gy, y.shape[3]))
    assert rel_err(nchw(xd.grad, cin), xr.grad) < 4e-3
    assert rel_err(wd.grad.cpu(), w.grad) < 4e-3  # the reference gradient passes through bf(w)'s backward: rounded to bf16
    if bias:
        assert rel_err(bd.grad.cpu(), b.grad) < 4e-3


def test_stem_conv(F):
    """normalise + pad + 7x7/2 conv as im2col GEMM vs (x-mean)/std -> F.conv2d."""
    g = torch.Generator().manual_seed(3)
    imgs = [torch.randint(0, 256, (3, 50, 70), generator=g, dtype=torch.uint8), torch.randint(0, 256, (3, 64, 61), generator=g, dtype=torch.uint8)]
    mean, std = torch.tensor([123.675, 116.28, 103.53]), torch.tensor([58.395, 57.12, 57.375])
    w = (torch.randn((64, 3, 7, 7), generator=g) * 0.05).requires_grad_(True)
    canvas = torch.zeros((2, 3, 64, 96))
    for i, im in enumerate(imgs):
        canvas[i, :, : im.shape[1], : im.shape[2]] = (im.float() - mean[:, None, None]) / std[:, None, None]
    yr = bf(TF.conv2d(bf(canvas), bf(w), None, 2, 3))
    gy = bf(torch.randn(yr.shape, generator=g))
    yr.backward(gy)
    wd = w.detach().to(DEV).requires_grad_(True)
    y, stats = F.stem_conv(wd, [i.to(DEV) for i in imgs], mean.to(DEV), std.to(DEV), 64, 96)
    assert rel_err(nchw(y), yr.detach()) < 4e-3
    y.backward(nhwc(gy))
    assert rel_err(wd.grad.cpu(), w.grad) < 4e-3


def test_batch_norm_residual_relu(F):
    g = torch.Generator().manual_seed(5)
    x = bf(torch.randn((3, 64, 11, 13), generator=g) * 2 + 0.5)
    res = bf(torch.randn((3, 64, 11, 13), generator=g))
    gamma = (1 + 0.2 * torch.randn(64, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(64, generator=g)).requires_grad_(True)
    rm, rv = torch.zeros(64), torch.ones(64)
    xr, rr = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
    # backbone/resnet.py:204-209 under autocast: the norm output is a bf16 tensor, `out += shortcut` a bf16 add
    yr = bf(TF.relu(bf(TF.batch_norm(xr, rm, rv, gamma, beta, True, 0.1, 1e-5)) + rr))
    gy = bf(torch.randn(yr.shape, generator=g))
    yr.backward(gy)
    xd, rd = nhwc(x).requires_grad_(True), nhwc(res).requires_grad_(True)
    gd, bd = gamma.detach().to(DEV).requires_grad_(True), beta.detach().to(DEV).requires_grad_(True)
    rmd, rvd = torch.zeros(64, device=DEV), torch.ones(64, device=DEV)
    xf = nchw(xd)
    stats = torch.stack([xf.sum((0, 2, 3)), (xf * xf).sum((0, 2, 3))]).to(DEV)
    y = F.batch_norm_act(xd, stats, gd, bd, rmd, rvd, rd, True, 0.1, 1e-5)
    assert rel_err(nchw(y), yr.detach()) < 4e-3
    assert torch.allclose(rmd.cpu(), rm, atol=1e-4) and torch.allclose(rvd.cpu(), rv, atol=1e-3)  # running stats
    y.backward(nhwc(gy))
    assert rel_err(nchw(xd.grad), xr.grad) < 5e-3
    assert rel_err(nchw(rd.grad), rr.grad) < 4e-3
    assert rel_err(gd.grad.cpu(), gamma.grad) < 1e-4 and rel_err(bd.grad.cpu(), beta.grad) < 1e-4


@pytest.mark.parametrize("shape", [(2, 64, 23, 31), (1, 64, 40, 66), (3, 32, 7, 9), (2, 128, 12, 10)])
def test_stem_tail_norm_relu_maxpool_one_pass(F, shape):
    """Round 6: norm -> relu_ -> max_pool2d(3, 2, 1) of the stem (backbone/resnet.py:355-359) as one pass each way
    (`batch_norm_relu_max_pool`): pooled values bit-identical to `max_pool_3x3_s2(batch_norm_act(..., relu=True))`, the input
    gradient identical to that path's up to the summation order of the two column sums, and both against the reference formula
    (train-mode batch_norm under autocast rounding: the norm output is a bf16 tensor).  Odd and even map sizes (windows cut by
    the border on every side), negative gammas (the affine map is decreasing there: the pool cannot be commuted with it)."""
    b, c, h, w = shape
    g = torch.Generator().manual_seed(h * 10 + w)
    x = bf(torch.randn((b, c, h, w), generator=g) * 2 + 0.3)
    gamma = (1 + 0.2 * torch.randn(c, generator=g))
    gamma[::5] *= -1
    gamma = gamma.requires_grad_(True)
    beta = (0.1 * torch.randn(c, generator=g)).requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    xr = x.clone().requires_grad_(True)
    yr = TF.max_pool2d(bf(TF.relu(bf(TF.batch_norm(xr, rm, rv, gamma, beta, True, 0.1, 1e-5)))), 3, 2, 1)
    gy = bf(torch.randn(yr.shape, generator=g))
    yr.backward(gy)

    def run(fused):
        xd = nhwc(x).requires_grad_(True)
        gd, bd = gamma.detach().to(DEV).requires_grad_(True), beta.detach().to(DEV).requires_grad_(True)
        rmd, rvd = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
        xf = nchw(xd)
        stats = torch.stack([xf.sum((0, 2, 3)), (xf * xf).sum((0, 2, 3))]).to(DEV)
        if fused:
            y = F.batch_norm_relu_max_pool(xd, stats, gd, bd, rmd, rvd, 0.1, 1e-5)
        else:
            y = F.max_pool_3x3_s2(F.batch_norm_act(xd, stats, gd, bd, rmd, rvd, None, True, 0.1, 1e-5))
        y.backward(nhwc(gy))
        return y.detach(), xd.grad, gd.grad, bd.grad, rmd, rvd

    yf, dxf, dgf, dbf, rmf, rvf = run(True)
    yu, dxu, dgu, dbu, rmu, rvu = run(False)
    assert torch.equal(yf, yu)                                  # pooled activations: bit for bit
    assert torch.equal(rmf, rmu) and torch.equal(rvf, rvu)       # running statistics
    assert rel_err(dxf.float(), dxu.float()) < 4e-3              # one bf16 step: the column sums are summed in another order
    assert rel_err(dgf, dgu) < 1e-5 and rel_err(dbf, dbu) < 1e-5
    assert rel_err(nchw(yf), yr.detach()) < 4e-3
    assert rel_err(nchw(dxf), xr.grad) < 5e-3
    assert rel_err(dgf.cpu(), gamma.grad) < 1e-4 and rel_err(dbf.cpu(), beta.grad) < 1e-4
    # inference form: the same pooled map from scale / shift
    with torch.no_grad():
        xd = nhwc(x)
        sc = (1 + 0.1 * torch.randn(c, generator=g)).to(DEV)
        sh = (0.1 * torch.randn(c, generator=g)).to(DEV)
        assert torch.equal(F.affine_relu_max_pool(xd, sc, sh), F.max_pool_3x3_s2(F.affine_act(xd, sc, sh, None, True)))


def test_group_norm_relu(F):
    g = torch.Generator().manual_seed(6)
    x = bf(torch.randn((2, 128, 9, 14), generator=g) * 1.5)
    gamma = (1 + 0.2 * torch.randn(128, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(128, generator=g)).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    yr = bf(TF.relu(TF.group_norm(xr, 32, gamma, beta, 1e-5)))
    gy = bf(torch.randn(yr.shape, generator=g))
    yr.backward(gy)
    xd = nhwc(x).requires_grad_(True)
    gd, bd = gamma.detach().to(DEV).requires_grad_(True), beta.detach().to(DEV).requires_grad_(True)
    y = F.group_norm_act(xd, gd, bd, 32, True, 1e-5)
    assert rel_err(nchw(y), yr.detach()) < 4e-3
    y.backward(nhwc(gy))
    assert rel_err(nchw(xd.grad), xr.grad) < 5e-3
    assert rel_err(gd.grad.cpu(), gamma.grad) < 1e-4 and rel_err(bd.grad.cpu(), beta.grad) < 1e-4


def test_pool_and_resample(F):
    g = torch.Generator().manual_seed(8)
    x = bf(torch.randn((2, 64, 13, 18), generator=g))
    xr = x.clone().requires_grad_(True)
    yr = TF.max_pool2d(xr, 3, 2, 1)
    gy = bf(torch.randn(yr.shape, generator=g))
    yr.backward(gy)
    xd = nhwc(x).requires_grad_(True)
    y = F.max_pool_3x3_s2(xd)
    assert torch.equal(nchw(y), yr.detach())  # max of bf16 values is exact
    y.backward(nhwc(gy))
    assert rel_err(nchw(xd.grad), xr.grad) < 5e-3
    # FPN top-down: lateral + nearest x2
    lat, top = bf(torch.randn((2, 64, 12, 16), generator=g)), bf(torch.randn((2, 64, 6, 8), generator=g))
    lr_, tr_ = lat.clone().requires_grad_(True), top.clone().requires_grad_(True)
    yr = bf(lr_ + TF.interpolate(tr_, scale_factor=2.0, mode="nearest"))
    gy = bf(torch.randn(yr.shape, generator=g))
    yr.backward(gy)
    ld, td = nhwc(lat).requires_grad_(True), nhwc(top).requires_grad_(True)
    y = F.fpn_upsample_add(ld, td)
    assert rel_err(nchw(y), yr.detach()) < 5e-3
    y.backward(nhwc(gy))
    assert torch.equal(nchw(ld.grad), lr_.grad) and rel_err(nchw(td.grad), tr_.grad) < 5e-3
    # bilinear x2 (+ addend)
    a = bf(torch.randn((2, 32, 7, 9), generator=g))
    add = bf(torch.randn((2, 32, 14, 18), generator=g))
    ar = a.clone().requires_grad_(True)
    yr = bf(TF.interpolate(ar, scale_factor=2.0, mode="bilinear", align_corners=False) + add)
    gy = bf(torch.randn(yr.shape, generator=g))
    yr.backward(gy)
    ad, addd = nhwc(a).requires_grad_(True), nhwc(add).requires_grad_(True)
    y = F.bilinear_up2(ad, addd)
    assert rel_err(nchw(y), yr.detach()) < 5e-3
    y.backward(nhwc(gy))
    assert rel_err(nchw(ad.grad), ar.grad) < 5e-3 and torch.equal(nchw(addd.grad), gy)


@pytest.mark.parametrize("shape", [(2, 32, 7, 9), (1, 128, 25, 42), (3, 64, 1, 5), (2, 128, 12, 1)])
@pytest.mark.parametrize("with_addend", [False, True])
def test_bilinear_up2_blocked_equals_per_pixel(F, shape, with_addend):
    """Round 6: nn.Upsample(scale 2, bilinear) of the semantic head (meta_arch/semantic_seg.py:206-211) by 2 x 2 output blocks
    (four taps per four outputs) equals the per-pixel kernel bit for bit - odd sizes, one-row / one-column maps (both taps
    clamped onto the same source row), with and without the running sum - and the reference formula to one bf16 step."""
    import os
    b, c, h, w = shape
    g = torch.Generator().manual_seed(h * 50 + w)
    a = bf(torch.randn((b, c, h, w), generator=g))
    add = bf(torch.randn((b, c, 2 * h, 2 * w), generator=g)) if with_addend else None
    ad, addd = nhwc(a), (nhwc(add) if with_addend else None)
    with torch.no_grad():
        y_block = F.bilinear_up2(ad, addd)
        os.environ["U2_BILINEAR_PER_PIXEL"] = "1"
        try:
            y_pixel = F.bilinear_up2(ad, addd)
        finally:
            del os.environ["U2_BILINEAR_PER_PIXEL"]
    assert torch.equal(y_block, y_pixel)
    yr = TF.interpolate(a, scale_factor=2.0, mode="bilinear", align_corners=False)
    yr = bf(bf(yr) + add) if with_addend else bf(yr)
    assert rel_err(nchw(y_block), yr) < 5e-3


def test_sem_seg_loss(F):
    """bilinear x4 + CE(mean, ignore 255): loss 1e-4 relative; logit gradient 1e-2 of its range (bf16 output)."""
    g = torch.Generator().manual_seed(9)
    logits = bf(torch.randn((2, 28, 12, 20), generator=g) * 2)
    tgt = torch.randint(0, 28, (2, 48, 80), generator=g)
    tgt[torch.rand(tgt.shape, generator=g) < 0.1] = 255
    lr_ = logits.clone().requires_grad_(True)
    loss_r = TF.cross_entropy(TF.interpolate(lr_, scale_factor=4.0, mode="bilinear", align_corners=False), tgt, ignore_index=255)
    (loss_r * 0.5).backward()
    ld = nhwc(logits).requires_grad_(True)
    loss = F.sem_seg_loss(ld, tgt.to(torch.uint8).to(DEV), 28, 255)
    assert float(loss) == pytest.approx(float(loss_r), rel=1e-4)
    (loss * 0.5).backward()
    assert rel_err(nchw(ld.grad, 28), lr_.grad) < 4e-3
    assert float(ld.grad[..., 28:].abs().max()) == 0.0


def test_head_losses(F):
    g = torch.Generator().manual_seed(10)
    # softmax CE over K+1 = 801 classes
    z = bf(torch.randn((70, 801), generator=g) * 3)
    lab = torch.randint(0, 801, (70,), generator=g)
    zr = z.clone().requires_grad_(True)
    lr_ = TF.cross_entropy(zr, lab)
    lr_.backward()
    zd = torch.zeros((70, 832), dtype=torch.bfloat16, device=DEV)
    zd[:, :801] = z.to(DEV)
    zd.requires_grad_(True)
    loss = F.softmax_cross_entropy(zd, lab.to(DEV), 801)
    assert float(loss) == pytest.approx(float(lr_), rel=1e-4)
    loss.backward()
    assert rel_err(zd.grad[:, :801].float().cpu(), zr.grad) < 4e-3
    # class-agnostic box regression L1 over foreground rows
    prop = torch.rand((70, 2), generator=g) * 100
    prop = torch.cat([prop, prop + 10 + torch.rand((70, 2), generator=g) * 50], 1)
    gtb = prop + torch.randn((70, 4), generator=g) * 3
    cls = torch.randint(0, 801, (70,), generator=g)
    cls[:20] = 800
    pred = bf(torch.randn((70, 4), generator=g))
    pr = pred.clone().requires_grad_(True)
    fg = cls < 800
    wts = (10.0, 10.0, 5.0, 5.0)
    lref = (pr[fg] - O.get_deltas(prop[fg], gtb[fg], wts)).abs().sum() / 70
    lref.backward()
    pd = torch.zeros((70, 32), dtype=torch.bfloat16, device=DEV)
    pd[:, :4] = pred.to(DEV)
    pd.requires_grad_(True)
    loss = F.box_reg_l1_loss(pd, prop.to(DEV), gtb.to(DEV), cls.to(DEV), 800, wts, 70)
    assert float(loss) == pytest.approx(float(lref), rel=1e-4)
    loss.backward()
    assert torch.equal(pd.grad[:, :4].float().cpu(), bf(pr.grad))
    # mask head: predictor restricted to the gt class + BCE(mean)
    x = bf(torch.randn((6, 256, 28, 28), generator=g))
    w = (torch.randn((800, 256, 1, 1), generator=g) * 0.05).requires_grad_(True)
    b = (torch.randn(800, generator=g) * 0.1).requires_grad_(True)
    cls = torch.randint(0, 800, (6,), generator=g)
    cls[1] = cls[0]
    tgt = torch.rand((6, 28, 28), generator=g) > 0.5
    xr = x.clone().requires_grad_(True)
    logit = bf(TF.conv2d(xr, bf(w), bf(b)))[torch.arange(6), cls]
    lref = TF.binary_cross_entropy_with_logits(logit, tgt.float())
    lref.backward()
    xd = nhwc(x).requires_grad_(True)
    wd, bd = w.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    loss = F.mask_predict_bce_loss(xd, wd, bd, cls.to(DEV), tgt.to(torch.uint8).to(DEV))
    assert float(loss) == pytest.approx(float(lref), rel=2e-3)
    loss.backward()
    assert rel_err(nchw(xd.grad), xr.grad) < 8e-3  # measured 5.6e-3: the reference rounds the selected logit to bf16 BEFORE the loss (one extra half step of the gradient's range), the kernel differentiates the fp32 logit
    assert rel_err(wd.grad.cpu(), w.grad) < 6e-3 and rel_err(bd.grad.cpu(), b.grad) < 6e-3
    # the deconvolution's unshuffled phases [n, P, P, (dy, dx, C)] as input (what the mask head hands over): the same loss (its
    # terms summed in another order) and bit for bit the same input gradient, in the phased layout
    xs = nhwc(x)                                                                          # [6, 28, 28, 256]
    xp = xs.view(6, 14, 2, 14, 2, 256).permute(0, 1, 3, 2, 4, 5).reshape(6, 14, 14, 1024).contiguous().requires_grad_(True)
    wp, bp = w.detach().to(DEV).requires_grad_(True), b.detach().to(DEV).requires_grad_(True)
    loss_p = F.mask_predict_bce_loss(xp, wp, bp, cls.to(DEV), tgt.to(torch.uint8).to(DEV), True)
    loss_p.backward()
    assert float(loss_p) == pytest.approx(float(loss), rel=1e-6)
    gp = xp.grad.view(6, 14, 14, 2, 2, 256).permute(0, 1, 3, 2, 4, 5).reshape(6, 28, 28, 256)
    assert torch.equal(gp, xd.grad)
    assert torch.allclose(wp.grad, wd.grad, rtol=1e-5, atol=1e-7) and torch.allclose(bp.grad, bd.grad, rtol=1e-5, atol=1e-7)


def test_roi_align_fwd_bwd(F, G):
    """multi-level ROIAlign forward / both backward variants vs the C oracle (itself pinned to the vendored C++ op)."""
    import u2seg_amd.layers.functional as FF

    g = torch.Generator().manual_seed(12)
    shapes, scales = [(24, 32), (12, 16), (6, 8), (3, 4)], [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    feats = [bf(torch.randn((2, 64, h, w), generator=g)) for h, w in shapes]
    xy = torch.rand((40, 2), generator=g) * 90
    wh = 2 + torch.rand((40, 2), generator=g) ** 2 * 120
    boxes = torch.cat([xy, xy + wh], 1)
    boxes[0] = torch.tensor([-20.0, -10.0, 40.0, 30.0])
    boxes[1] = torch.tensor([50.0, 40.0, 50.0, 40.0])  # empty box -> zeros
    bidx = torch.randint(0, 2, (40,), generator=g).float()
    rois = torch.cat([bidx[:, None], boxes], 1)
    lv_ref = O.assign_boxes_to_levels(boxes, 2, 5)
    lv = F.assign_levels(boxes.to(DEV), 2, 5)
    assert torch.equal(lv.cpu().long(), lv_ref)  # bit exact level routing
    for ps in (7, 14):
        fr = [f.clone().requires_grad_(True) for f in feats]
        out_ref = torch.zeros((40, 64, ps, ps))
        for l in range(4):
            idx = torch.nonzero(lv_ref == l)[:, 0]
            if len(idx):
                out_ref = out_ref.index_put((idx,), O.roi_align(fr[l], rois[idx], ps, scales[l]))
        gy = bf(torch.randn(out_ref.shape, generator=g))
        out_ref.backward(gy)
        for atomic in (False, True):
            FF.ROI_ALIGN_BWD_ATOMIC = atomic
            fd = [nhwc(f).requires_grad_(True) for f in feats]
            y = F.roi_align(fd, rois.to(DEV), lv, ps, scales, grad_scale=1.0 / 3)
            assert rel_err(y.permute(0, 3, 1, 2).float().cpu(), out_ref.detach()) < 5e-3
            assert float(y[1].abs().max()) == 0.0
            y.backward(gy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV))
            for l in range(4):
                if fr[l].grad is None:  # no ROI routed to this level
                    assert float(fd[l].grad.abs().max()) == 0.0
                else:
                    assert rel_err(nchw(fd[l].grad), fr[l].grad / 3) < 5e-3, (ps, atomic, l)
        FF.ROI_ALIGN_BWD_ATOMIC = False
    # ground-truth mask crop (golden from the reference; exact except threshold ties of the rotated stand-in)
    out = F.mask_crop(torch.from_numpy(G["crop_masks"]).to(torch.uint8).to(DEV), torch.cat(
        [torch.arange(5.0)[:, None], torch.from_numpy(G["crop_boxes"])], 1).to(DEV), 28)
    ref = O.crop_and_resize_masks(torch.from_numpy(G["crop_masks"]), torch.from_numpy(G["crop_boxes"]), 28)
    assert torch.equal(out.cpu().bool(), ref)


def test_roi_align_elongated_boxes(F):
    """ROIAlign on elongated boxes (column footprints of 40 ... 160 pixels at their level, bins wider than they are tall) next to
    ordinary ones: the forward tables and the backward gather (bin rectangles that fit its 48-bin LDS stage and 14 x 14 grids
    that do not) vs the C oracle."""
    g = torch.Generator().manual_seed(31)
    shapes, scales = [(24, 160), (12, 80), (6, 40), (3, 20)], [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    feats = [bf(torch.randn((2, 64, h, w), generator=g)) for h, w in shapes]
    rows = []
    for w_, h_ in ((150.0, 12.0), (300.0, 20.0), (520.0, 24.0), (600.0, 90.0), (158.0, 60.0), (40.0, 30.0), (630.0, 8.0)):
        for x0, y0 in ((3.0, 2.0), (20.5, 30.25)):
            rows.append([x0, y0, min(x0 + w_, 655.0), min(y0 + h_, 99.0)])
    boxes = torch.tensor(rows)
    n = boxes.shape[0]
    bidx = (torch.arange(n) % 2).float()
    rois = torch.cat([bidx[:, None], boxes], 1)
    lv_ref = O.assign_boxes_to_levels(boxes, 2, 5)
    lv = F.assign_levels(boxes.to(DEV), 2, 5)
    assert torch.equal(lv.cpu().long(), lv_ref)
    assert int(((boxes[:, 2] - boxes[:, 0]) * torch.tensor(scales)[lv_ref] > 41).sum()) >= 4  # wide footprints are present
    for ps in (7, 14):
        fr = [f.clone().requires_grad_(True) for f in feats]
        out_ref = torch.zeros((n, 64, ps, ps))
        for l in range(4):
            idx = torch.nonzero(lv_ref == l)[:, 0]
            if len(idx):
                out_ref = out_ref.index_put((idx,), O.roi_align(fr[l], rois[idx], ps, scales[l]))
        gy = bf(torch.randn(out_ref.shape, generator=g))
        out_ref.backward(gy)
        fd = [nhwc(f).requires_grad_(True) for f in feats]
        y = F.roi_align(fd, rois.to(DEV), lv, ps, scales, grad_scale=1.0)
        assert rel_err(y.permute(0, 3, 1, 2).float().cpu(), out_ref.detach()) < 5e-3, ps
        y.backward(gy.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV))
        for l in range(4):
            if fr[l].grad is None:
                assert float(fd[l].grad.abs().max()) == 0.0
            else:
                assert rel_err(nchw(fd[l].grad), fr[l].grad) < 5e-3, (ps, l)


def test_roi_grad_tap_combines_poolers(F):
    """Three 7x7 poolers (gradient scale 1/3) and one 14x14 pooler on tapped FPN maps: the single deferred multi-set gather
    must give the sum of the four separate ROIAlign backward passes (what autograd forms without the tap)."""
    torch.manual_seed(3)
    shapes = [(2, 40, 56, 64), (2, 20, 28, 64), (2, 10, 14, 64), (2, 5, 7, 64)]
    scales = (0.25, 0.125, 0.0625, 0.03125)

    def boxes(n):
        xy = torch.rand(n, 2) * torch.tensor([150.0, 100.0])
        wh = 4 + torch.rand(n, 2) * torch.tensor([120.0, 90.0])
        img = torch.randint(0, 2, (n, 1)).float()
        return torch.cat([img, xy, xy + wh], 1).to(DEV)

    sets = [(boxes(60), 7, 1.0 / 3), (boxes(50), 7, 1.0 / 3), (boxes(40), 7, 1.0 / 3), (boxes(30), 14, 1.0)]
    base = [torch.randn(s, device=DEV).bfloat16() for s in shapes]
    douts = [torch.randn((r.shape[0], p, p, 64), device=DEV).bfloat16() for r, p, _ in sets]

    def run(tapped):
        feats = [f.clone().requires_grad_() for f in base]
        use = F.roi_grad_tap(feats) if tapped else feats
        total = 0.0
        for (rois, p, gs), d in zip(sets, douts):
            lv = F.assign_levels(rois[:, 1:].contiguous(), 2, 5)
            out = F.roi_align(use, rois, lv, p, scales, gs)
            total = total + (out.float() * d.float()).sum()
        total.backward()
        return [f.grad.float().cpu() for f in feats]

    ref, got = run(False), run(True)
    for l, (a, b) in enumerate(zip(got, ref)):
        assert rel_err(a, b) < 5e-3, l  # bf16 rounding of one sum instead of four partial maps


def test_roi_gather_takes_the_other_readers_gradients(F):
    """Round 6: the tapped FPN maps are fan_out handles (three readers: semantic head, RPN, ROI poolers).  The tap's backward
    then hands every level's _FanOutFn.backward a zero placeholder, and THAT node runs the level's gather with the two other
    gradients as addends (u2_roi_align_bwd_gather_sum: fp32 sum, one rounding) instead of gather + u2_add_n.  Against the
    unfolded path (F.ROI_SUM_FOLD False): the maps' gradients agree to the one bf16 rounding the fold saves, a level no ROI
    lands on gets exactly the sum of the other two, and nothing deferred is left behind."""
    torch.manual_seed(5)
    shapes = [(2, 40, 56, 64), (2, 20, 28, 64), (2, 10, 14, 64), (2, 5, 7, 64)]
    scales = (0.25, 0.125, 0.0625, 0.03125)

    def boxes(n, small=False):
        xy = torch.rand(n, 2) * torch.tensor([150.0, 100.0])
        wh = 4 + torch.rand(n, 2) * (torch.tensor([20.0, 16.0]) if small else torch.tensor([120.0, 90.0]))
        img = torch.randint(0, 2, (n, 1)).float()
        return torch.cat([img, xy, xy + wh], 1).to(DEV)

    # small boxes only: everything is assigned to the finest level, the coarser maps see no ROI at all
    for small in (False, True):
        sets = [(boxes(60, small), 7, 1.0 / 3), (boxes(50, small), 7, 1.0 / 3), (boxes(40, small), 7, 1.0 / 3), (boxes(30, small), 14, 1.0)]
        base = [torch.randn(s, device=DEV).bfloat16() for s in shapes]
        douts = [torch.randn((r.shape[0], p, p, 64), device=DEV).bfloat16() for r, p, _ in sets]
        w_sem = [torch.randn(s, device=DEV).bfloat16() for s in shapes]
        w_rpn = [torch.randn(s, device=DEV).bfloat16() for s in shapes]

        def run(fold):
            old = F.ROI_SUM_FOLD
            F.ROI_SUM_FOLD = fold
            try:
                feats = [f.clone().requires_grad_() for f in base]
                hs = [F.fan_out(f, 3) for f in feats]
                use = F.roi_grad_tap([h[2] for h in hs])
                total = 0.0
                for (rois, p, gs), d in zip(sets, douts):
                    lv = F.assign_levels(rois[:, 1:].contiguous(), 2, 5)
                    out = F.roi_align(use, rois, lv, p, scales, gs)
                    total = total + (out.float() * d.float()).sum()
                for h, a, b in zip(hs, w_sem, w_rpn):
                    total = total + (h[0] * a).float().sum() + (h[1] * b).float().sum()   # bf16 gradients a and b
                total.backward()
                F.assert_no_deferred_gradients()
                return [f.grad.float().cpu() for f in feats]
            finally:
                F.ROI_SUM_FOLD = old

        ref, got = run(False), run(True)
        for l, (a, b) in enumerate(zip(got, ref)):
            assert rel_err(a, b) < 8e-3, (small, l)     # one bf16 rounding of the sum instead of two (measured 4.6e-3)
            assert float(b.abs().max()) > 0
        if small:
            lv_all = torch.cat([F.assign_levels(r[:, 1:].contiguous(), 2, 5) for r, _, _ in sets])
            assert int(lv_all.max()) == 0
            for l in (1, 2, 3):   # no ROI: the gather writes the other readers' sum, exactly (fp32 sum of two bf16, one rounding)
                want = (w_sem[l].float() + w_rpn[l].float()).bfloat16().float().cpu()
                assert torch.equal(got[l], want), l


def test_index_bookkeeping_bit_exact(F, G):
    """levels, IoU matching (+low-quality), NMS keep lists: identical integers to the oracle and the reference goldens."""
    lv = F.assign_levels(torch.from_numpy(G["lvl_boxes"]).to(DEV), 2, 5)
    assert np.array_equal(lv.cpu().numpy().astype(np.int64), G["lvl_out"])
    gt, cand = torch.from_numpy(G["match_gt"]), torch.from_numpy(G["match_cand"])
    ngt = torch.tensor([4], dtype=torch.int32, device=DEV)
    m, lab, val = F.iou_match(cand.to(DEV), gt[None].to(DEV), ngt, 0.3, 0.7, True)
    assert np.array_equal(m[0].cpu().numpy(), G["match_rpn_idx"]) and np.array_equal(lab[0].cpu().numpy(), G["match_rpn_lab"])
    assert np.array_equal(val[0].cpu().numpy(), G["match_iou"].max(0))  # IoU values bit exact
    m, lab, _ = F.iou_match(cand[None].to(DEV), gt[None].to(DEV), ngt, 0.5, 0.5, False)
    assert np.array_equal(m[0].cpu().numpy(), G["match_roi_idx"]) and np.array_equal(lab[0].cpu().numpy(), G["match_roi_lab"])
    # batch of two images with different gt counts (second has none -> all labels 0)
    gt2 = torch.zeros((2, 4, 4))
    gt2[0] = gt
    m, lab, _ = F.iou_match(cand.to(DEV), gt2.to(DEV), torch.tensor([4, 0], dtype=torch.int32, device=DEV), 0.3, 0.7, True)
    assert np.array_equal(lab[0].cpu().numpy(), G["match_rpn_lab"]) and int(lab[1].abs().sum()) == 0 and int(m[1].abs().sum()) == 0
    # NMS
    b, s = torch.from_numpy(G["nms_boxes"]), torch.from_numpy(G["nms_scores"])
    order = torch.sort(s, descending=True, stable=True)[1]
    for thr in (0.5, 0.65):
        keep, nk = F.batched_nms(b[order][None].to(DEV), torch.zeros((1, 400), dtype=torch.int32, device=DEV),
                                 torch.tensor([400], dtype=torch.int32, device=DEV), thr, 400)
        got = order[keep[0, : int(nk[0])].long().cpu()]
        assert np.array_equal(got.numpy(), G["nms_keep_%02d" % int(thr * 100)])
    # grouped, 2 images, ragged counts, truncation to max_keep
    g = torch.Generator().manual_seed(4)
    n = 1500
    bb = torch.rand((2, n, 2), generator=g) * 300
    bb = torch.cat([bb, bb + 8 + torch.rand((2, n, 2), generator=g) * 90], 2)
    sc = torch.rand((2, n), generator=g)
    grp = torch.randint(0, 5, (2, n), generator=g)
    cnt = [1500, 777]
    sb, sg = torch.zeros_like(bb), torch.zeros_like(grp)
    orders = []
    for i in range(2):
        o = torch.sort(sc[i, : cnt[i]], descending=True, stable=True)[1]
        orders.append(o)
        sb[i, : cnt[i]], sg[i, : cnt[i]] = bb[i, o], grp[i, o]
    keep, nk = F.batched_nms(sb.to(DEV), sg.to(torch.int32).to(DEV), torch.tensor(cnt, dtype=torch.int32, device=DEV), 0.65, 300)
    for i in range(2):
        ref = O.nms(bb[i, : cnt[i]], sc[i, : cnt[i]], 0.65, grp[i, : cnt[i]])[:300]
        got = orders[i][keep[i, : int(nk[i])].long().cpu()]
        assert got.tolist() == ref.tolist()
    # decode: same operation order as the oracle; expf differs by <= 1 ulp => 1e-6 relative
    src, d = torch.from_numpy(G["b2b_src"]), torch.from_numpy(G["b2b_s0_noisy"])
    out = F.apply_deltas(src.to(DEV), d.to(DEV), (10.0, 10.0, 5.0, 5.0))
    np.testing.assert_allclose(out.cpu().numpy(), G["b2b_s0_applied"], rtol=1e-5, atol=1e-4)


def test_rpn_loss(F):
    """fused per-level RPN loss (BCE sum + L1 sum over positives, / (256*B)) vs the oracle's statement."""
    g = torch.Generator().manual_seed(14)
    grids, strides = [(12, 16), (6, 8), (3, 4)], [4, 8, 16]
    cells = [O.generate_cell_anchors([s * 8], (0.5, 1.0, 2.0)).float() for s in strides]
    anchors = O.grid_anchors(grids, strides, cells, 0.0)
    B, A = 2, 3
    atot = sum(a.shape[0] for a in anchors)
    objs = [bf(torch.randn((B, h, w, A), generator=g)) for h, w in grids]
    dlts = [bf(torch.randn((B, h, w, 4 * A), generator=g) * 0.5) for h, w in grids]
    gt = torch.tensor([[[5.0, 5, 40, 50], [20, 10, 60, 44]], [[1.0, 2, 30, 20], [0, 0, 0, 0]]])
    ngt = torch.tensor([2, 1], dtype=torch.int32)
    acat = torch.cat(anchors)
    labels, match = torch.zeros((B, atot), dtype=torch.int8), torch.zeros((B, atot), dtype=torch.int32)
    for b in range(B):
        iou = O.pairwise_iou(gt[b, : ngt[b]], acat)
        m, l = O.matcher(iou, [0.3, 0.7], [0, -1, 1], True)
        l[torch.rand(atot, generator=g) < 0.5] = -1
        labels[b], match[b] = l, m.int()
    # oracle statement
    ov = [o.clone().requires_grad_(True) for o in objs]
    dv = [d.clone().requires_grad_(True) for d in dlts]
    oc = torch.cat([o.reshape(B, -1) for o in ov], 1)
    dc = torch.cat([d.reshape(B, -1, 4) for d in dv], 1)
    pos = labels == 1
    tgt = torch.stack([O.get_deltas(acat, gt[b][match[b].long()], (1.0, 1.0, 1.0, 1.0)) for b in range(B)])
    loc = (dc[pos] - tgt[pos]).abs().sum() / (256 * B)
    valid = labels >= 0
    cls = TF.binary_cross_entropy_with_logits(oc[valid], labels[valid].float(), reduction="sum") / (256 * B)
    (cls + loc).backward()
    od = [torch.zeros((B, h, w, 32), dtype=torch.bfloat16, device=DEV) for h, w in grids]
    dd = [torch.zeros((B, h, w, 32), dtype=torch.bfloat16, device=DEV) for h, w in grids]
    for i in range(3):
        od[i][..., :A], dd[i][..., : 4 * A] = objs[i].to(DEV), dlts[i].to(DEV)
        od[i].requires_grad_(True)
        dd[i].requires_grad_(True)
    lc, ll = F.rpn_losses(labels.to(DEV), match.to(DEV), gt.to(DEV), [a.to(DEV) for a in anchors], A, 256 * B, od, dd)
    assert float(lc) == pytest.approx(float(cls), rel=1e-4) and float(ll) == pytest.approx(float(loc), rel=1e-4)
    (lc + ll).backward()
    for i in range(3):
        assert rel_err(od[i].grad[..., :A].float().cpu(), ov[i].grad) < 5e-3
        assert rel_err(dd[i].grad[..., : 4 * A].float().cpu(), dv[i].grad) < 5e-3


def test_rpn_fused_predictors_equal_separate(F):
    """StandardRPNHead runs the objectness and the anchor-delta 1x1 convs as ONE conv of 15 channels; the maps must be those of
    the two separate convs (the reference's form, rpn.py:170-176) bit for bit, the losses equal up to summation order, and the gradients of the shared
    3x3 conv's output, of both predictors and of the level inputs equal to one bf16 ulp of their scale (the separate form
    rounds two data gradients to bf16 and adds them, the fused form rounds their sum once)."""
    from u2seg_amd.modeling.rpn import StandardRPNHead

    g = torch.Generator().manual_seed(8)
    B, A, grids = 2, 3, [(24, 32), (12, 16), (6, 8)]
    head = StandardRPNHead(in_channels=64, num_anchors=A).to(DEV)
    with torch.no_grad():
        for m in (head.conv, head.objectness_logits, head.anchor_deltas):
            m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.05)
            m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    feats = [bf(torch.randn((B, h, w, 64), generator=g)) for h, w in grids]
    atot = sum(h * w * A for h, w in grids)
    labels = torch.randint(-1, 2, (B, atot), generator=g).to(torch.int8)
    match = torch.randint(0, 3, (B, atot), generator=g).to(torch.int32)
    gt = torch.rand((B, 3, 4), generator=g) * 50
    gt[..., 2:] += gt[..., :2] + 4
    anchors = []
    for h, w in grids:
        xy = torch.rand((h * w * A, 2), generator=g) * 80
        anchors.append(torch.cat([xy, xy + 8 + torch.rand((h * w * A, 2), generator=g) * 40], 1).to(DEV))
    res = {}
    for fused in (True, False):
        head.fuse_predictors = fused
        head.zero_grad()
        xs = [f.to(torch.bfloat16).to(DEV).requires_grad_(True) for f in feats]
        objs, dlts = head(xs)
        lc, ll = F.rpn_losses(labels.to(DEV), match.to(DEV), gt.to(DEV), anchors, A, 256.0 * B, objs, dlts)
        (lc + 2.0 * ll).backward()
        res[fused] = dict(objs=[o[..., :A].float().cpu() for o in objs], dlts=[d[..., : 4 * A].float().cpu() for d in dlts],
                          losses=(float(lc), float(ll)), gx=[x.grad.float().cpu() for x in xs],
                          gw=[p.grad.float().cpu().clone() for p in head.parameters()])
    a, b = res[True], res[False]
    for u, v in zip(a["objs"] + a["dlts"], b["objs"] + b["dlts"]):
        assert torch.equal(u, v)
    for u, v in zip(a["losses"], b["losses"]):  # sums of per-block partial sums by float atomics: equal up to their order
        assert abs(u - v) <= 1e-5 * abs(v)
    for u, v in zip(a["gx"] + a["gw"], b["gx"] + b["gw"]):
        assert rel_err(u, v) < 8e-3


def test_arena_direct_grads_and_cached_layouts(F):
    """Under solver.FlatSGD the 1x1-conv / linear weight gradients and the BN affine gradients are accumulated by the
    kernels straight into the optimizer's arena, and the bf16 weight layouts are cached and rewritten in one launch per
    step.  Both must be indistinguishable from the plain autograd path (no optimizer) on the same inputs."""
    import copy

    from u2seg_amd.layers.modules import BatchNorm2d, Conv2d, Linear
    from u2seg_amd.solver import FlatSGD

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.c1 = Conv2d(64, 96, 1, bias=False, norm=BatchNorm2d(96), activation="relu")
            self.c2 = Conv2d(96, 64, 3, padding=1, bias=False, norm=BatchNorm2d(64), activation="relu")
            self.c3 = Conv2d(64, 40, 1, bias=True)
            self.fc = Linear(64, 70)
            self.fc1 = Linear(64 * 3 * 5, 48)  # applied as a 3x5 "fully connected" conv like FastRCNNConvFCHead.fc1

        def forward(self, x):
            y = self.c3(self.c2(self.c1(x)))
            z = self.fc(x.reshape(-1, 64))
            u = F.conv2d(x[:, :3, :5].contiguous(), self.fc1.weight.view(48, 64, 3, 5), self.fc1.bias, 1, 0, relu=True,
                         param=self.fc1.weight)
            return y.float().square().mean() + z.float().square().mean() + u.float().square().mean()

    torch.manual_seed(0)
    plain = Net().to(DEV).train()
    arena = copy.deepcopy(plain)
    opt = FlatSGD(arena, lr=0.05, momentum=0.9, weight_decay=1e-4, weight_decay_norm=1e-4, clip_value=0.0)
    xs = [nhwc(torch.randn(2, 64, 12, 20)).requires_grad_() for _ in range(3)]  # data gradients too (dgrad layouts)
    popt = torch.optim.SGD(plain.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    for it, x in enumerate(xs):
        opt.zero_grad()
        popt.zero_grad()
        la, lp = arena(x), plain(x)
        # cached layouts == layouts built on the fly (fp32-atomic BN statistics may flip a few bf16 roundings)
        # (both nets accumulate BN statistics with fp32 atomics, so they drift apart by bf16 rounding flips: gradients
        # are compared on the first iteration only, later iterations through the loss they produce)
        assert float(la) == pytest.approx(float(lp), rel=2e-3 if it == 0 else 2e-2), it
        la.backward()
        lp.backward()
        for (k, a), (_, b) in zip(arena.named_parameters(), plain.named_parameters()):
            assert a.grad.data_ptr() == a._u2_grad.data_ptr(), k  # still the arena view
            if it == 0:
                assert rel_err(a.grad, b.grad) < (1e-4 if it == 0 else 2e-2), (it, k)
        opt.step(1.0)
        popt.step()
        if it == 0:
            for (k, a), (_, b) in zip(arena.named_parameters(), plain.named_parameters()):
                assert rel_err(a.detach(), b.detach()) < (1e-4 if it == 0 else 2e-2), (it, k)  # zero-initialised parameters are lr * gradient
    assert len(opt._layout_entries) >= 10  # fwd + dgrad layouts were registered and refreshed by step()
    modes = set()
    for p_, key, ent in opt._layout_entries:  # the batched LDS-tiled refresh == the per-tensor kernel, bit for bit
        n_, cin_, t_, cp_, npad_, mode_ = key
        modes.add(mode_)
        if mode_ == 3:  # a conv bias as the epilogue adds it: fp32, rounded through bf16 (round 4: refreshed by the same launch)
            assert torch.equal(ent[0], p_.detach().bfloat16().float()), key
            continue
        fresh = F._weight_layout(p_.detach().reshape(n_, cin_, t_, 1), cp_, npad_, mode_)
        assert torch.equal(fresh.reshape(-1), ent[0].reshape(-1)), key
    assert modes >= {0, 1, 2}
    # in-place edits through torch invalidate the cached layouts (autograd version check)
    with torch.no_grad():
        arena.c1.weight.mul_(2.0)
        plain.c1.weight.mul_(2.0)
    assert float(arena(xs[0])) == pytest.approx(float(plain(xs[0])), rel=2e-2)
    assert int(arena.state_dict()["c1.norm.num_batches_tracked"]) == 4


def test_sgd_clip_step():
    """per-parameter L2 clip to 1.0 + SGD(momentum .9, wd) vs torch on CPU (solver/build.py:36-37,63-73)."""
    from u2seg_amd.solver import FlatSGD

    torch.manual_seed(0)
    lin = torch.nn.Sequential(torch.nn.Linear(300, 200), torch.nn.Linear(200, 7)).to(DEV)
    ref = torch.nn.Sequential(torch.nn.Linear(300, 200), torch.nn.Linear(200, 7))
    ref.load_state_dict({k: v.cpu() for k, v in lin.state_dict().items()})
    opt = FlatSGD(lin, lr=0.1, momentum=0.9, weight_decay=1e-3, clip_value=1.0)
    ropt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-3)
    for it in range(3):
        x = torch.randn(16, 300)
        opt.zero_grad()
        ropt.zero_grad()
        (lin(x.to(DEV)) ** 2).sum().backward()
        (ref(x) ** 2).sum().backward()
        for p in ref.parameters():
            torch.nn.utils.clip_grad_norm_(p, 1.0, 2.0)
        ropt.step()
        opt.step(1.0)
    for (k, a), (_, b) in zip(lin.state_dict().items(), ref.state_dict().items()):
        assert torch.allclose(a.cpu(), b, rtol=1e-4, atol=1e-5), k


def test_kmeans(golden_dir=None):
    """assign (exact-fp32 MFMA distances) + update vs the reference-generated golden labels / centroids."""
    from u2seg_amd.cluster import kmeans as KM

    g = np.load(os.path.join(ROOT, "tests", "golden", "kmeans_golden.npz"))
    x, init = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["init"]).to(DEV)
    cl, c = KM.kmeans(x, init, int(g["niter"]))
    assert np.array_equal(cl.cpu().numpy(), g["labels"])
    np.testing.assert_allclose(c.cpu().numpy(), g["centroids"], rtol=1e-5, atol=1e-5)
    # empty cluster -> NaN centroid row, K not a multiple of 32, N not a multiple of the tile
    xs = torch.randn((1000, 32), device=DEV)
    cs = torch.randn((5, 32), device=DEV)
    cs[3] = 1e4
    lab = KM.assign(xs, cs)
    ref = O.kmeans_assign(xs.cpu(), cs.cpu())
    assert torch.equal(lab.cpu(), ref)
    cn, cnt = KM.update(xs, lab, 5)
    assert torch.isnan(cn[3]).all() and float(cnt[3]) == 0
    # K = 300 (the reference's setting: all ten 32-centroid tiles present, the branch-free pass) and K = 700 (two full
    # passes + a partial one); labels may differ from the oracle only where the two nearest centroids are tied to 1e-5
    gk = torch.Generator().manual_seed(4)
    for k in (300, 700):
        xs = torch.randn((5003, 64), generator=gk)
        cs = xs[torch.randperm(5003, generator=gk)[:k]] + 0.05 * torch.randn((k, 64), generator=gk)
        lab = KM.assign(xs.to(DEV), cs.to(DEV)).cpu()
        ref = O.kmeans_assign(xs, cs)
        bad = torch.nonzero(lab != ref)[:, 0]
        d_lab = ((xs[bad] - cs[lab[bad]]) ** 2).sum(1)
        d_ref = ((xs[bad] - cs[ref[bad]]) ** 2).sum(1)
        assert bad.numel() <= 5 and torch.allclose(d_lab, d_ref, rtol=1e-5), (k, bad.numel())


def test_kmeans_screened_assign_equals_exact():
    """The split-bf16 screening pass + exact re-check of the undecided points must return the labels of the exact-fp32 kernel
    for every point: on well separated data, on unit-norm clustered data with many near-ties (centroids drawn next to each
    other), with duplicated centroids (exact ties: first index wins), and with K not a multiple of 16."""
    from u2seg_amd.cluster import kmeans as KM

    g = torch.Generator().manual_seed(21)
    cases = []
    x = torch.randn((20000, 768), generator=g)
    cases.append((x, x[torch.randperm(20000, generator=g)[:300]] + 0.01 * torch.randn((300, 768), generator=g)))
    xu = _clustered_unit_rows(30011, 384, 5, nclusters=12)
    cu = xu[torch.randperm(30011, generator=g)[:299]].clone()
    cu[7] = cu[3]                       # an exact duplicate
    cu[100:140] = cu[50:90] + 1e-6      # forty near-duplicates: far below the screening resolution
    cases.append((xu, cu))
    cases.append((torch.randn((1000, 64), generator=g) * 5, torch.randn((17, 64), generator=g) * 5))
    for x, c in cases:
        xd, cd = x.to(DEV), c.to(DEV)
        fast = KM.assign(xd, cd)
        n_checked = KM.last_recheck_count(xd.device)
        exact = KM.assign(xd, cd, exact=True)
        if not torch.equal(fast, exact):  # say which of the two left the fp64 answer (diagnostic for a failure seen once in round 3)
            d = (xd.double() ** 2).sum(1, keepdim=True) - 2 * xd.double() @ cd.double().t() + (cd.double() ** 2).sum(1)[None]
            ref = d.argmin(1)
            raise AssertionError("screened vs exact labels differ at %d points; vs fp64 argmin: screened %d, exact %d, re-checked %s"
                                 % (int((fast != exact).sum()), int((fast != ref).sum()), int((exact != ref).sum()), n_checked))
        ref = O.kmeans_assign(x[:2000], c)
        bad = torch.nonzero(exact[:2000].cpu() != ref)[:, 0]
        d_lab = ((x[bad] - c[exact[:2000].cpu()[bad]]) ** 2).sum(1)
        d_ref = ((x[bad] - c[ref[bad]]) ** 2).sum(1)
        assert torch.allclose(d_lab, d_ref, rtol=1e-5)  # the exact kernel vs the oracle: only at fp32-level ties
        assert 0 <= n_checked <= x.shape[0]
    assert n_checked < 1000  # the last (well separated, K = 17) case re-checks almost nothing


def test_kmeans_shadow_pass_equals_exact(monkeypatch):
    """Round 6: the first screening pass over the 16-bit shadow of x (u2_kmeans_prepare / u2_kmeans_assign_shadow, kmeans_coarse_kernel:
    persistent work-groups, index bits in the distances).  Labels must be the exact-fp32 kernel's for every point - more tiles than CUs
    with a ragged last one, near-duplicate and duplicate centroids, rows of tiny norm (the index bits' share of the margin), a zero row,
    NaN / Inf rows - and the shadow must follow x: reused for the same tensor, re-made after an in-place write and for another tensor."""
    from u2seg_amd.cluster import kmeans as KM

    monkeypatch.setattr(KM, "SHADOW_MIN_POINTS", 256)
    KM.release_shadow()
    g = torch.Generator().manual_seed(61)
    n, d, k = 70001, 768, 300
    centers = torch.randn((k, d), generator=g) * 2
    xm = centers[torch.randint(0, k, (n,), generator=g)] + 0.5 * torch.randn((n, d), generator=g)
    xm[100:164] *= 1e-4
    xm[200] = 0
    xm[300, 5] = float("nan")
    xm[301, 7] = float("inf")
    cm = centers + 0.3 * torch.randn((k, d), generator=g)
    xu = _clustered_unit_rows(30011, 384, 5, nclusters=12)
    cu = xu[torch.randperm(30011, generator=g)[:299]].clone()
    cu[7] = cu[3]
    cu[100:140] = cu[50:90] + 1e-6
    cfar = cm[:40].clone()
    cfar[5] = 1e6                        # a centroid beyond the fp16 range of the scaled shadow: its candidates are not finite
    cases = [(xm, cm), (xu, cu), (torch.randn((1000, 64), generator=g) * 5, torch.randn((17, 64), generator=g) * 5),
             (torch.randn((300, 32), generator=g) * 3, torch.randn((5, 32), generator=g) * 3),   # one step per tile, two tiles
             (xm[:9000, :256] * 1e-12, cm[:, :256] * 1e-12), (xm[:9000, :256] * 1e12, cm[:, :256] * 1e12),   # the power-of-two scale
             (xm[:9000], cfar)]
    for i, (x, c) in enumerate(cases):
        xd, cd = x.to(DEV), c.to(DEV)
        KM._ws_cache.pop("assign:" + str(xd.device), None)   # the first pass starts switched on
        fast = KM.assign(xd, cd)
        und = KM.last_coarse_undecided(xd.device)
        sh = KM._shadow(xd)
        assert sh is not None and KM._shadow(xd) is sh        # made by assign(), found again
        exact = KM.assign(xd, cd, exact=True)
        assert torch.equal(fast, exact), (i, int((fast != exact).sum()), und)
        if i == 0:
            assert 0 < und < n // 50, und                     # the pass ran, decided the mixture and handed on the odd rows
            ref = O.kmeans_assign(x[:600], c)
            lab = exact[:600].cpu()
            bad = torch.nonzero((lab != ref) & torch.isfinite(x[:600]).all(1))[:, 0]
            assert torch.allclose(((x[bad] - c[lab[bad]]) ** 2).sum(1), ((x[bad] - c[ref[bad]]) ** 2).sum(1), rtol=1e-5)  # fp32-level ties only
            # an in-place write invalidates the shadow: row 0 becomes another cluster's member
            other = int(exact[1]) if int(exact[1]) != int(exact[0]) else int(exact[2])
            xd[0] = cd[other]
            fast2 = KM.assign(xd, cd)
            assert KM._shadow(xd) is sh                        # same buffer, re-filled (the version moved)
            assert int(fast2[0]) == other and torch.equal(fast2, KM.assign(xd, cd, exact=True))
            xc = xd.clone()                                    # another tensor object: its own shadow content
            xc[1] = cd[int(fast2[0])]
            fast3 = KM.assign(xc, cd)
            assert int(fast3[1]) == int(fast2[0]) and torch.equal(fast3, KM.assign(xc, cd, exact=True))
    KM.release_shadow()
    KM._ws_cache.pop("assign:" + str(DEV), None)


@pytest.mark.parametrize("k", [321, 641, 800, 1280])
def test_kmeans_more_than_320_centroids(monkeypatch, k):
    """Round 6: K > 320 (u2seg_R50_800 clusters into 800): the first screening pass runs once per block of 320 centroids over the shadow
    of x, the blocks' candidates are merged per point and what stays undecided goes to the exact kernel.  Labels == the exact kernel's,
    with a block of one centroid (321, 641), duplicates across blocks, a ragged last tile and more tiles than CUs."""
    from u2seg_amd.cluster import kmeans as KM

    monkeypatch.setattr(KM, "SHADOW_MIN_POINTS", 256)
    KM.release_shadow()
    g = torch.Generator().manual_seed(70 + k)
    n, d = 66003, 256
    centers = torch.randn((k, d), generator=g) * 2
    x = centers[torch.randint(0, k, (n,), generator=g)] + 0.5 * torch.randn((n, d), generator=g)
    c = centers + 0.3 * torch.randn((k, d), generator=g)
    c[k - 1] = c[5]                      # an exact duplicate in another block: the lower index must win
    c[330 if k > 330 else 10] = c[17] + 1e-6
    x[77] = float("nan")
    xd, cd = x.to(DEV), c.to(DEV)
    fast = KM.assign(xd, cd)
    und = KM.last_recheck_count(xd.device)
    exact = KM.assign(xd, cd, exact=True)
    assert KM.last_recheck_count(xd.device) is None
    assert torch.equal(fast, exact), (k, int((fast != exact).sum()), und)
    assert 0 < und < n // 4, und          # screened: the duplicates' members and the NaN row, not everything
    lab = exact[:400].cpu()
    ref = O.kmeans_assign(x[:400], c)
    bad = torch.nonzero((lab != ref) & torch.isfinite(x[:400]).all(1))[:, 0]
    assert torch.allclose(((x[bad] - c[lab[bad]]) ** 2).sum(1), ((x[bad] - c[ref[bad]]) ** 2).sum(1), rtol=1e-5)
    cnew, counts = KM.update(xd, fast, k)
    assert float(counts.sum()) == n and torch.equal(counts.cpu(), torch.bincount(fast.cpu(), minlength=k).float())
    KM.release_shadow()
    KM._ws_cache.pop("assign:" + str(DEV), None)


def test_kmeans_first_pass_is_translation_invariant(monkeypatch):
    """Round 6: the first screening pass works on x - mean(x), c - mean(x) (argmin_j |x - c_j|^2 does not change under a common translation):
    clustered rows with a common offset of 25 per dimension - |x| |c| is 600 x the cluster spread, the untranslated margin would leave
    every point undecided - are decided by it as the same rows without the offset are; labels == the exact kernel's in both cases."""
    from u2seg_amd.cluster import kmeans as KM

    monkeypatch.setattr(KM, "SHADOW_MIN_POINTS", 256)
    g = torch.Generator().manual_seed(91)
    n, d, k = 30000, 256, 64
    cen = torch.randn((k, d), generator=g) * 2
    x0 = cen[torch.randint(0, k, (n,), generator=g)] + 0.5 * torch.randn((n, d), generator=g)
    c0 = cen + 0.3 * torch.randn((k, d), generator=g)
    left = []
    for off in (0.0, 25.0):
        xd, cd = (x0 + off).to(DEV), (c0 + off).to(DEV)
        KM._ws_cache.pop("assign:" + str(DEV), None)
        KM.release_shadow()
        fast = KM.assign(xd, cd)
        left.append(KM.last_coarse_undecided(xd.device))
        assert torch.equal(fast, KM.assign(xd, cd, exact=True)), off
    assert left[0] < n // 100 and left[1] < n // 20, left
    KM.release_shadow()
    KM._ws_cache.pop("assign:" + str(DEV), None)


def test_kmeans_random_shapes_equal_exact(monkeypatch):
    """Random (N, D, K) through every screening path (shadow forced on; K up to 1280, D from one 32-dimension step up): clustered rows,
    unstructured rows with rows as centroids, bf16-valued rows with norms spread over six decades and a duplicated centroid, rows with a
    large common mean.  Labels of two consecutive calls (the second may run with the first pass switched off) == the exact kernel's."""
    from u2seg_amd.cluster import kmeans as KM

    monkeypatch.setattr(KM, "SHADOW_MIN_POINTS", 256)
    g = torch.Generator().manual_seed(5)
    for i in range(16):
        d = [32, 64, 96, 128, 256, 384, 768][int(torch.randint(0, 7, (1,), generator=g))]
        k = int(torch.randint(2, 1281, (1,), generator=g)) if i % 3 else int(torch.randint(2, 321, (1,), generator=g))
        n = int(torch.randint(256, 40000, (1,), generator=g))
        kind = i % 4
        if kind == 0:
            cen = torch.randn((k, d), generator=g) * 2
            x = cen[torch.randint(0, k, (n,), generator=g)] + 0.5 * torch.randn((n, d), generator=g)
            c = cen + 0.3 * torch.randn((k, d), generator=g)
        else:
            x = torch.randn((n, d), generator=g)
            if kind == 2:
                x = (x * 3).bfloat16().float()
                x[: n // 10] *= 1e-3
                x[n // 10: n // 5] *= 1e3
            elif kind == 3:
                x = x + 10.0
            c = x[torch.randperm(n, generator=g)[:k]].clone() if k <= n else torch.randn((k, d), generator=g) + (10.0 if kind == 3 else 0.0)
            if kind == 2 and k > 3:
                c[k - 1] = c[0]
        xd, cd = x.to(DEV), c.to(DEV)
        KM._ws_cache.pop("assign:" + str(DEV), None)
        KM.release_shadow()
        first, second = KM.assign(xd, cd), KM.assign(xd, cd)
        exact = KM.assign(xd, cd, exact=True)
        assert torch.equal(first, exact) and torch.equal(second, exact), (i, kind, n, d, k, int((first != exact).sum()))
    KM.release_shadow()
    KM._ws_cache.pop("assign:" + str(DEV), None)


def test_kmeans_two_level_screen_modes():
    """The two-level screen (round 4): on clustered data the first pass (leading bf16 pieces only) decides nearly everything and
    the labels are the exact kernel's; on unstructured data it leaves most points undecided, the labels are still the exact
    kernel's, and the pass switches itself off for the following calls on that workspace (its undecided count reads 0)."""
    from u2seg_amd.cluster import kmeans as KM

    g = torch.Generator().manual_seed(33)
    n, d, k = 24000, 768, 300
    centers = torch.randn((k, d), generator=g) * 2
    xm = (centers[torch.randint(0, k, (n,), generator=g)] + 0.5 * torch.randn((n, d), generator=g)).to(DEV)
    cm = (centers + 0.3 * torch.randn((k, d), generator=g)).to(DEV)
    xr = torch.randn((n, d), generator=g).to(DEV)
    cr = xr[torch.randperm(n, generator=g)[:k].to(DEV)].clone()
    KM._ws_cache.pop("assign:" + str(xm.device), None)   # a fresh workspace: the screening state starts with the first pass on
    fast = KM.assign(xm, cm)
    und = KM.last_coarse_undecided(xm.device)
    assert torch.equal(fast, KM.assign(xm, cm, exact=True)) and 0 <= und < n // 100, und
    fast = KM.assign(xr, cr)
    und = KM.last_coarse_undecided(xr.device)
    assert und > n // 4, und                      # the first pass ran and decided little ...
    assert torch.equal(fast, KM.assign(xr, cr, exact=True))
    fast = KM.assign(xr, cr)                      # ... so it is switched off now
    assert KM.last_coarse_undecided(xr.device) == 0 and torch.equal(fast, KM.assign(xr, cr, exact=True))
    KM._ws_cache.pop("assign:" + str(xm.device), None)


@pytest.mark.parametrize("data_kind", ["mixture", "randn"])
def test_kmeans_config4_full_size_properties(data_kind):
    """BASELINE config 4 at its OWN size: N = 1 000 000 x 768, K = 300 - more than 65 535 row blocks,
    label buckets of ~3 300 rows, the two-level screen's full workspace; both of bench.py's data kinds.  One Lloyd iteration
    through the C ABI, checked by size-independent properties (nn_utils.py:353-364):
      * screened labels == the exact-fp32 kernel's labels on a 50 000-row sample spread over the whole range INCLUDING the
        last block (and == the fp64 argmin except at fp32-level ties);
      * counts sum to N and equal bincount(labels);
      * centroids == the mean of their rows, in fp64, for the clusters the sample touches first + a stride over all K;
      * an empty cluster (a far-away centroid) comes out as a NaN row with count 0;
      * a second iteration on the updated centroids still agrees with the exact kernel on the sample."""
    from u2seg_amd.cluster import kmeans as KM

    n, d, k = 1000000, 768, 300
    g = torch.Generator(device=DEV).manual_seed(7)
    if data_kind == "mixture":
        centers = torch.randn((k, d), generator=g, device=DEV) * 2
        x = centers[torch.randint(0, k, (n,), generator=g, device=DEV)] + 0.5 * torch.randn((n, d), generator=g, device=DEV)
        c = centers + 0.3 * torch.randn((k, d), generator=g, device=DEV)
    else:
        x = torch.randn((n, d), generator=g, device=DEV)
        c = x[torch.randperm(n, generator=g, device=DEV)[:k]].clone()
    c[k - 1] = 1.0e3                                   # nobody's nearest centroid: the empty cluster
    KM._ws_cache.pop("assign:" + str(x.device), None)
    sample = torch.cat([torch.arange(0, n, 21, device=DEV)[:47000], torch.arange(n - 3000, n, device=DEV)])
    xs = x[sample]

    def check_labels(lab, cc):
        assert lab.dtype == torch.int64 and int(lab.min()) >= 0 and int(lab.max()) < k
        exact = KM.assign(xs, cc, exact=True)
        assert torch.equal(lab[sample], exact), int((lab[sample] != exact).sum())
        dd = (xs[:8000].double() ** 2).sum(1, keepdim=True) - 2 * xs[:8000].double() @ cc.double().t() + (cc.double() ** 2).sum(1)[None]
        dd = torch.nan_to_num(dd, nan=float("inf"))
        ref = dd.argmin(1)
        bad = torch.nonzero(exact[:8000] != ref)[:, 0]
        assert bad.numel() <= 8
        if bad.numel():
            a, b = dd[bad, exact[bad]], dd[bad, ref[bad]]
            assert torch.allclose(a, b, rtol=1e-5)

    lab = KM.assign(x, c)
    check_labels(lab, c)
    cn, cnt = KM.update(x, lab, k)
    bc = torch.bincount(lab, minlength=k)
    assert float(cnt.double().sum()) == float(n) and torch.equal(cnt.long(), bc)
    assert int(bc[k - 1]) == 0 and bool(torch.isnan(cn[k - 1]).all()) and bool(torch.isfinite(cn[: k - 1][bc[: k - 1] > 0]).all())
    picked = sorted(set(lab[sample[:40]].tolist()) | set(range(0, k - 1, 37)))
    for j in picked:
        rows = torch.nonzero(lab == j)[:, 0]
        if rows.numel() == 0:
            assert bool(torch.isnan(cn[j]).all())
            continue
        want = x[rows].double().mean(0)
        assert float((cn[j].double() - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max())), j
    # second iteration on the updated centroids (the empty cluster's NaN row parked far away again: what a NaN centroid does to
    # the arg-min is the subject of test_kmeans_vs_oracle, not of this size test)
    c2 = cn.clone()
    c2[k - 1] = 1.0e3
    lab2 = KM.assign(x, c2)
    check_labels(lab2, c2)
    assert int((lab2 == k - 1).sum()) == 0
    del x, xs
    KM._ws_cache.pop("assign:" + str(DEV), None)
    torch.cuda.empty_cache()


def _clustered_unit_rows(n, d, seed, nclusters=40):
    g = torch.Generator().manual_seed(seed)
    centers = torch.randn((nclusters, d), generator=g)
    x = centers[torch.randint(0, nclusters, (n,), generator=g)] + 0.4 * torch.randn((n, d), generator=g)
    return torch.nn.functional.normalize(x, dim=1)


def test_knn_vs_oracle_and_reference():
    """u2_knn (fp32-MFMA candidate selection + difference-form refinement) vs the reference-generated lists (its own
    partition loop + merge, tests/golden/knn_golden.npz) and vs the oracle on ragged sizes: distances to 1e-5 relative,
    neighbour ids identical except where two train rows are equidistant (the rule of the reference's verify branch)."""
    from tests.parity_checks import knn_lists_agree
    from u2seg_amd.cluster import knn as KN

    g = np.load(os.path.join(ROOT, "tests", "golden", "knn_golden.npz"))
    x = torch.from_numpy(g["x"]).to(DEV)
    d, ind = KN.partitioned_kNN(x, K=int(g["K"]), partitions_size=int(g["partitions_size"]))
    assert d.dtype == torch.float32 and ind.dtype == torch.int64
    ind_np = ind.cpu().numpy()
    knn_lists_agree(g["x"], d.cpu().numpy(), ind_np, g["d_knns"], g["ind_knns"])
    # ids may only differ where the planted twin rows (5 = 650, 305 = 310) tie: the reference's merge is an unstable argsort
    differs = ind_np != g["ind_knns"]
    assert np.isin(ind_np[differs], [5, 650, 305, 310]).all() and np.isin(g["ind_knns"][differs], [5, 650, 305, 310]).all()
    assert (d[:, 0] == 0).all()  # the row itself (or its exact twin), bit-exact zero like the difference form
    np.testing.assert_allclose(KN.first_order_density(d).cpu().numpy(), 1 / g["d_knns"].mean(1), rtol=1e-5)

    # separate query / train sets, D = 768, sizes off every tile boundary, several K
    pool = _clustered_unit_rows(3011 + 777, 768, 1)
    xt, xq = pool[:3011].contiguous(), pool[3011:].contiguous()
    ind_o, d_o = O.knn(xt, xq, 28)  # the first k columns are the lists for every smaller k
    for k in (1, 5, 20, 28):
        ind_g, d_g = KN.kNN(xt.to(DEV), xq.to(DEV), K=k)
        bad = knn_lists_agree(xq.numpy(), d_g.cpu().numpy(), ind_g.cpu().numpy(), d_o[:, :k].numpy(), ind_o[:, :k].numpy())
        assert bad == 0, (k, bad)
    # fewer train rows than the K + 4 candidate slots, and the K > N_train error of the reference's assumption
    ind_o, d_o = O.knn(xt[:22], xq[:130], 20)
    ind_g, d_g = KN.kNN(xt[:22].to(DEV), xq[:130].to(DEV), K=20)
    assert torch.equal(ind_g.cpu(), ind_o)
    np.testing.assert_allclose(d_g.cpu().numpy(), d_o.numpy(), rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        KN.kNN(xt[:10].to(DEV), xq[:4].to(DEV), K=20)
    with pytest.raises(ValueError):
        KN.kNN(xt.to(DEV), xq.to(DEV), K=29)
    assert KN.kNN(xt.to(DEV), xq[:0].to(DEV), K=20)[0].shape == (0, 20)
    # unnormalised rows with a large common offset: the expanded form used for selection loses digits here, the spare
    # candidates + difference-form refinement must still return the oracle's lists
    xo = (xt[:1500] * 3 + 5).contiguous()
    ind_o, d_o = O.knn(xo, xo, 20)
    ind_g, d_g = KN.kNN(xo.to(DEV), xo.to(DEV), K=20)
    knn_lists_agree(xo.numpy(), d_g.cpu().numpy(), ind_g.cpu().numpy(), d_o.numpy(), ind_o.numpy(), rtol=1e-4)


def test_knn_full_size_properties():
    """N = 100 000 x 768 (the order of one reference partition): size-independent properties instead of an oracle run -
    every list ascending with the row itself first at distance exactly 0; the reported distances equal a direct
    recomputation; no sampled train row is closer than a row's K-th neighbour unless it is in the list."""
    from u2seg_amd.cluster import knn as KN

    n, dim, k = 100000, 768, 20
    x = _clustered_unit_rows(n, dim, 3, nclusters=300).to(DEV)
    d, ind = KN.partitioned_kNN(x, K=k)
    assert bool((ind[:, 0] == torch.arange(n, device=DEV)).all()) and bool((d[:, 0] == 0).all())
    assert bool((d[:, 1:] >= d[:, :-1]).all())
    rows = torch.randperm(n, generator=torch.Generator().manual_seed(0))[:2000].to(DEV)
    direct = ((x[rows][:, None, :] - x[ind[rows]]) ** 2).sum(-1)
    assert torch.allclose(direct, d[rows], rtol=1e-5, atol=1e-6)
    cols = torch.randperm(n, generator=torch.Generator().manual_seed(1))[:4000].to(DEV)
    dd = torch.cdist(x[rows], x[cols]) ** 2
    closer = dd < (d[rows, k - 1] * (1 - 1e-4) - 1e-6)[:, None]
    listed = (ind[rows][:, None, :] == cols[None, :, None]).any(-1)
    assert not bool((closer & ~listed).any())


def _knn_shard_worker(rank, world, port, out):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from u2seg_amd.cluster import knn as KN

    torch.cuda.set_device(0)
    g = np.load(os.path.join(ROOT, "tests", "golden", "knn_golden.npz"))
    x = torch.from_numpy(g["x"]).to(DEV)
    lo, hi = (0, 250) if rank == 0 else (250, x.shape[0])  # uneven shards
    d, ind = KN.partitioned_kNN_sharded(x[lo:hi], K=int(g["K"]))
    out[rank] = (d.cpu().numpy(), ind.cpu().numpy())
    dist.destroy_process_group()


def test_knn_row_sharded_two_ranks():
    """kNN lists with the rows sharded over two ranks (one all-gather of the ragged shards, no other exchange): the
    concatenated lists must equal the reference-generated single-process golden."""
    import socket

    import torch.multiprocessing as mp

    from tests.parity_checks import knn_lists_agree

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_knn_shard_worker, args=(2, port, out), nprocs=2, join=True)
        (d0, i0), (d1, i1) = out[0], out[1]
    g = np.load(os.path.join(ROOT, "tests", "golden", "knn_golden.npz"))
    assert d0.shape[0] == 250 and d1.shape[0] == g["x"].shape[0] - 250
    knn_lists_agree(g["x"], np.concatenate([d0, d1]), np.concatenate([i0, i1]), g["d_knns"], g["ind_knns"])




@pytest.fixture
def fixed_order_statistics(F):
    """The two tests that let the HIP model run freely against a CPU run (no teacher forcing) take the BN column statistics
    from a fixed-order reduction of the stored conv output instead of the conv epilogue's fp32 atomics
    (functional.set_deterministic_stats): the forward pass is then bit-reproducible from run to run, so a comparison that
    holds once holds every time - no retries.  The strict comparisons are the teacher-forced ones
    (tests/test_gpu_bookkeeping.py)."""
    F.set_deterministic_stats(True)
    try:
        yield
    finally:
        F.set_deterministic_stats(False)


@pytest.mark.parametrize("branch", ["per_image_permutations", "batched_keys"])
def test_whole_model_vs_oracle(F, fixed_order_statistics, branch):
    """u2seg_R50_800 on 2 synthetic 192x256 images, name-keyed weights, free running (the teacher-forced 1e-3 comparison is
    tests/test_gpu_bookkeeping.py::test_heads_teacher_forced_losses): the HIP path's 10 losses vs the oracle with bf16
    emulation (2% relative: bf16 accumulation-order noise moves a few proposals across NMS / matching thresholds).
    "per_image_permutations" injects the reference's CPU randperm stream (the per-image sampling branch) and also compares
    with the reference's own fp32 losses (tests/golden/model_small.json, 3%); "batched_keys" runs the branch every training
    step takes (padded / stacked bookkeeping, u2_topk_rows) with injected keys that the oracle turns into permutations."""
    from oracle.model import OracleModel
    from tests.golden.make_fixtures import det_fill
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import make_synthetic_batch
    from u2seg_amd.modeling import build_model, set_permutation_source

    cfg = get_cfg()
    cfg.merge_from_file(CFG)
    cfg.merge_from_list(["MODEL.DEVICE", DEV])
    model = build_model(cfg)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(det_fill(k, v.cpu()).to(DEV))
    model.train()
    sd = {k: v.cpu() for k, v in model.state_dict().items()}

    batch = make_synthetic_batch(2, height=192, width=256, device=DEV)
    if branch == "per_image_permutations":
        # the same CPU randperm stream the reference fixture was generated with (torch.manual_seed(5), CPU generator)
        set_permutation_source(lambda n, device=None: torch.randperm(n))
        torch.manual_seed(5)
        losses = model(batch)
        sum(losses.values()).backward()
        set_permutation_source(None)
        torch.cuda.synchronize()
        om = OracleModel(cfg, sd, emulate_bf16=True)
        torch.manual_seed(5)
        ref = om.train_forward(make_synthetic_batch(2, height=192, width=256))
    else:
        from tests.test_gpu_bookkeeping import KeyRecorder
        from u2seg_amd.modeling import set_key_source

        rec = KeyRecorder(5)
        set_key_source(rec)
        try:
            losses = model(batch)
            sum(losses.values()).backward()
        finally:
            set_key_source(None)
        torch.cuda.synchronize()
        assert len(rec.calls) == 2  # one draw for the anchors, one for the ROIs
        cpu_batch = make_synthetic_batch(2, height=192, width=256)
        ngt = [len(x["instances"]) for x in cpu_batch]

        def key_fn(stage, i, n):
            if stage == "rpn":
                return rec.calls[0][i, :n]
            k = rec.calls[1]
            npad = k.shape[1] - max(ngt)  # the HIP path pads the proposals of every image to the post-NMS top-k
            return torch.cat([k[i, : n - ngt[i]], k[i, npad : npad + ngt[i]]])

        om = OracleModel(cfg, sd, emulate_bf16=True, key_fn=key_fn)
        ref = om.train_forward(cpu_batch)
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "model_small.json")))["losses"]
    report = {k: (float(losses[k]), float(ref[k]), fx[k]) for k in sorted(ref)}
    print(json.dumps(report, indent=1))
    # Dense losses (every pixel / anchor / ROI contributes): 2% vs the bf16-emulating oracle, 3% vs the fp32 reference.
    # Sparse box-regression terms average |delta error| over a handful of foreground ROIs whose membership flips with
    # bf16 rounding noise (the oracle's own bf16-vs-fp32 spread on them is 20-40%), so they only get a sanity band.
    dense = ["loss_sem_seg", "loss_rpn_cls", "loss_cls_stage0", "loss_cls_stage1", "loss_cls_stage2", "loss_mask"]
    for k in dense:
        assert float(losses[k]) == pytest.approx(float(ref[k]), rel=2e-2), (k, report)
        if branch == "per_image_permutations":
            assert float(losses[k]) == pytest.approx(fx[k], rel=3e-2), (k, report)
    assert float(losses["loss_rpn_loc"]) == pytest.approx(float(ref["loss_rpn_loc"]), rel=5e-2), report
    for k in ("loss_box_reg_stage0", "loss_box_reg_stage1", "loss_box_reg_stage2"):
        # free running, the few foreground ROIs these average over differ between the two runs (the proposals come from each
        # side's own RPN maps); with shared ROIs they agree to 1e-5 (teacher-forced test).  Same order of magnitude only.
        assert 0.3 * float(ref[k]) < float(losses[k]) < 3.0 * float(ref[k]), (k, report)
    gn = float(model.backbone.bottom_up.stem.conv1.weight.grad.norm())
    assert gn == gn and gn > 0


def test_whole_model_production_statistics_path(F):
    """The same free-running comparison through the PRODUCTION forward (BN column statistics from the conv epilogues' fp32 atomics,
    the path bench.py runs; the two tests above use the fixed-order switch so that they are bit-reproducible): the dense losses
    stay inside a 2.5 % band of the bf16 oracle, and two runs of the production path within 2 % of each other - the atomics'
    accumulation order is the only difference between them, a last-ulp difference of the statistics that the random-weight
    network amplifies to 0.7 % of loss_sem_seg (measured), which is why the strict comparisons are teacher-forced."""
    from oracle.model import OracleModel
    from tests.golden.make_fixtures import det_fill
    from tests.test_gpu_bookkeeping import KeyRecorder
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import make_synthetic_batch
    from u2seg_amd.modeling import build_model, set_key_source

    cfg = get_cfg()
    cfg.merge_from_file(CFG)
    cfg.merge_from_list(["MODEL.DEVICE", DEV])
    model = build_model(cfg)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(det_fill(k, v.cpu()).to(DEV))
    model.train()
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    batch = make_synthetic_batch(2, height=192, width=256, device=DEV)
    F.set_deterministic_stats(False)
    runs, recs = [], []
    for _ in range(2):
        rec = KeyRecorder(5)
        set_key_source(rec)
        try:
            with torch.no_grad():
                for k, v in model.state_dict().items():   # the forward updates the running statistics: same start for both runs
                    v.copy_(sd[k].to(DEV))
            losses = model(batch)
            sum(losses.values()).backward()
        finally:
            set_key_source(None)
        torch.cuda.synchronize()
        runs.append({k: float(v) for k, v in losses.items()})
        recs.append(rec)
    rec = recs[0]
    cpu_batch = make_synthetic_batch(2, height=192, width=256)
    ngt = [len(x["instances"]) for x in cpu_batch]

    def key_fn(stage, i, n):
        if stage == "rpn":
            return rec.calls[0][i, :n]
        k = rec.calls[1]
        npad = k.shape[1] - max(ngt)
        return torch.cat([k[i, : n - ngt[i]], k[i, npad : npad + ngt[i]]])

    ref = OracleModel(cfg, sd, emulate_bf16=True, key_fn=key_fn).train_forward(cpu_batch)
    dense = ["loss_sem_seg", "loss_rpn_cls", "loss_cls_stage0", "loss_cls_stage1", "loss_cls_stage2", "loss_mask"]
    # Bands: 2.5 % against the oracle and 2 % run to run (without the fixed-order switch the run is not bit-reproducible; round 4
    # had 4 % / 3 % after one full-suite run at 2 % had failed).
    # Round 5, measured over five runs of this test on one box: the dense losses deviate from the oracle by at most 1.3 %
    # (loss_sem_seg; the others <= 0.45 %) and two production runs from each other by at most 0.8 %; loss_rpn_loc (a few dozen
    # foreground anchors) by 4.3 % / 5.4 %.
    for k in dense:
        assert runs[0][k] == pytest.approx(float(ref[k]), rel=2.5e-2), (k, runs[0], float(ref[k]))
        assert runs[1][k] == pytest.approx(runs[0][k], rel=2e-2), (k, runs)
    assert runs[0]["loss_rpn_loc"] == pytest.approx(float(ref["loss_rpn_loc"]), rel=8e-2)


def test_backbone_and_heads_blockwise_vs_oracle(F):
    """Teacher-forced per-layer parity at the tolerance north_star names: every conv + norm (+ residual)(+ ReLU) unit of the
    ResNet, the FPN, every conv + GroupNorm unit and the predictor of the semantic head and the RPN head of the HIP path get
    the bf16 oracle's activations as input and must reproduce the oracle's output of that unit to 1e-3 relative L2.  Whole
    bottleneck blocks (three units + shortcut chained on the HIP side) and the 11-layer semantic head end to end are held to
    2e-3: a random-weight train-mode-BN network amplifies rounding noise ~1.2x per layer, so longer free-running
    chains only measure that amplification (the oracle's own bf16-vs-fp32 feature distance is 40-70%)."""
    from oracle.model import OracleModel
    from tests.golden.make_fixtures import det_fill
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import make_synthetic_batch
    from u2seg_amd.modeling import build_model

    cfg = get_cfg()
    cfg.merge_from_file(CFG)
    cfg.merge_from_list(["MODEL.DEVICE", DEV])
    model = build_model(cfg)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(det_fill(k, v.cpu()).to(DEV))
    model.train()
    sd = {k: v.cpu() for k, v in model.state_dict().items()}
    om = OracleModel(cfg, sd, emulate_bf16=True)
    cap, scap = {}, {}
    with torch.no_grad():
        images, sizes, padded = om.preprocess(make_synthetic_batch(2, height=192, width=256))
        rf = om.backbone(images, cap)
        rsem = om.sem_seg_logits(rf, scap)
        robj, rdl = om.rpn_head(rf)
    unit, chain = {}, {}

    def err(a, b):
        return float((a - b).norm() / b.norm())

    batch = make_synthetic_batch(2, height=192, width=256, device=DEV)
    bu = model.backbone.bottom_up
    with torch.no_grad():
        imgs = [x["image"] for x in batch]
        mean, std = model.pixel_mean.view(-1).float().contiguous(), model.pixel_std.view(-1).float().contiguous()
        unit["stem"] = err(nchw(bu.stem(imgs, mean, std, padded)), cap["stem"])
        prev = cap["stem"]
        res = {}
        for si, nb in zip(range(2, 6), [3, 4, 6, 3]):
            stage = getattr(bu, "res%d" % si)
            for bi in range(nb):
                blk, key = stage[bi], "res%d.%d" % (si, bi)
                xin = nhwc(prev)
                unit[key + ".conv1"] = err(nchw(blk.conv1(xin)), cap[key + ".conv1"])
                unit[key + ".conv2"] = err(nchw(blk.conv2(nhwc(cap[key + ".conv1"]))), cap[key + ".conv2"])
                if blk.shortcut is not None:
                    unit[key + ".shortcut"] = err(nchw(blk.shortcut(xin)), cap[key + ".shortcut"])
                out3 = blk.conv3(nhwc(cap[key + ".conv2"]), residual=nhwc(cap[key + ".shortcut"]), relu=True)
                unit[key + ".conv3"] = err(nchw(out3), cap[key])
                chain[key] = err(nchw(blk(xin)), cap[key])
                prev = cap[key]
            res["res%d" % si] = nhwc(prev)
        feats = model.backbone.forward_features(res)
        for k in ("p2", "p3", "p4", "p5", "p6"):  # lateral 1x1 + BN (+ upsample-add) + output 3x3 + BN: two units
            chain[k] = err(nchw(feats[k]), rf[k])
        rfd = {k: nhwc(v) for k, v in rf.items()}
        head = model.sem_seg_head
        for f, sh in zip(head.in_features, head.scale_heads):  # every conv + GN + ReLU unit on the oracle's input of that unit
            y_ref = rf[f]
            for i, op in enumerate(sh):
                name = "sem_seg_head.%s.%d" % (f, i)
                if name in scap:
                    unit[name] = err(nchw(op(nhwc(y_ref))), scap[name])
                    y_ref = scap[name]
                else:  # the bilinear x2 between two units: the oracle's next input
                    y_ref = bf(TF.interpolate(y_ref, scale_factor=2.0, mode="bilinear", align_corners=False))
        unit["sem_seg_head.predictor"] = err(nchw(head.predictor(nhwc(scap["sem_seg_head.sum"])), 28), rsem)
        chain["sem_logits"] = err(nchw(head.layers(rfd), 28), rsem)
        objs, dlts = model.proposal_generator.rpn_head([rfd[f] for f in model.proposal_generator.in_features])
        for i in range(5):  # 3x3 + ReLU and the two 1x1 predictors: two units
            chain["rpn_obj_l%d" % i] = err(objs[i][..., :3].reshape(2, -1).float().cpu(), robj[i])
            chain["rpn_dlt_l%d" % i] = err(dlts[i][..., :12].reshape(2, -1, 4).float().cpu(), rdl[i])
    print(json.dumps({"unit": unit, "chain": chain}, indent=1))
    for k, v in unit.items():
        assert v < 1e-3, (k, v)
    for k, v in chain.items():
        assert v < 2e-3, (k, v)


def test_inference_tails_vs_oracle_and_reference(F, G):
    """Inference bookkeeping on identical inputs: score filter + per-class NMS + top-k (bit-exact indices), mask paste
    (exact except >= 0.5 ties), panoptic merge (bit-exact vs the reference golden), and the full eval forward of the HIP path
    vs the bf16 oracle (semantic argmax agreement; detections are compared teacher-forced because random-weight scores sit
    within 1e-6 of each other and their order is rounding noise)."""
    from oracle.model import OracleModel
    from tests.golden.make_fixtures import det_fill
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import make_synthetic_batch
    from u2seg_amd.modeling import build_model
    from u2seg_amd.modeling.inference import (combine_semantic_and_instance_outputs, fast_rcnn_inference_single_image,
                                              paste_masks_in_image)
    from u2seg_amd.structures import Instances

    # 1. panoptic merge vs the reference's own output
    inst = Instances((96, 128))
    inst.pred_masks = torch.from_numpy(G["pan_masks"]).to(DEV)
    inst.scores = torch.from_numpy(G["pan_scores"]).to(DEV)
    inst.pred_classes = torch.from_numpy(G["pan_classes"]).to(DEV)
    pan, info = combine_semantic_and_instance_outputs(inst, torch.from_numpy(G["pan_sem"]).to(DEV), 0.5, 4096 // 8, 0.5)
    assert np.array_equal(pan.cpu().numpy(), G["pan_out"])
    ref_info = json.loads(str(G["pan_info"]))
    assert [(d["id"], d["isthing"], d["category_id"]) for d in info] == [(d["id"], d["isthing"], d["category_id"]) for d in ref_info]

    # 2. box filtering + per-class NMS on the oracle's (fp32) boxes and scores
    g = torch.Generator().manual_seed(21)
    boxes = torch.rand((400, 2), generator=g) * 150
    boxes = torch.cat([boxes, boxes + 5 + torch.rand((400, 2), generator=g) * 80], 1)
    scores = torch.softmax(torch.randn((400, 12), generator=g) * 2, dim=1)
    cfg = get_cfg()
    cfg.merge_from_file(CFG)
    cfg.merge_from_list(["MODEL.DEVICE", "cpu", "MODEL.ROI_HEADS.SCORE_THRESH_TEST", 0.05])
    om = OracleModel(cfg, {}, emulate_bf16=False)
    rb, rs, rc = om.box_inference_single(boxes, scores, (180, 240))
    res, kept = fast_rcnn_inference_single_image(boxes.to(DEV), scores.to(DEV), (180, 240), 0.05, 0.5, 100)
    assert torch.equal(res.pred_classes.cpu(), rc) and torch.equal(res.scores.cpu(), rs) and torch.equal(res.pred_boxes.tensor.cpu(), rb)

    # 2b. the batched routine on a ragged batch (different box counts, one image without boxes) == one image at a time
    from u2seg_amd.modeling.inference import fast_rcnn_inference

    bl = [boxes[:400].to(DEV), boxes[100:350].to(DEV), boxes[:0].to(DEV)]
    sl = [scores[:400].to(DEV), scores[100:350].to(DEV), scores[:0].to(DEV)]
    shapes = [(180, 240), (150, 200), (180, 240)]
    batched, _ = fast_rcnn_inference(bl, sl, shapes, 0.05, 0.5, 100)
    for i in range(3):
        one, _ = fast_rcnn_inference_single_image(bl[i], sl[i], shapes[i], 0.05, 0.5, 100)
        assert torch.equal(batched[i].pred_boxes.tensor, one.pred_boxes.tensor) and torch.equal(batched[i].scores, one.scores)
        assert torch.equal(batched[i].pred_classes, one.pred_classes)
    assert len(batched[2]) == 0 and len(batched[0]) == 100

    # 3. mask paste
    probs = torch.rand((7, 28, 28), generator=g)
    pb = torch.tensor([[3.0, 4, 60, 50], [10.5, 2.25, 30.75, 44.5], [0, 0, 128, 96], [100, 70, 127.5, 95.5], [5, 5, 6, 6.5],
                       [40, 30, 90, 80], [-5, -5, 20, 20]])
    ref = OracleModel.paste_masks(probs, pb, (96, 128))
    got_dev = paste_masks_in_image(probs.to(DEV), pb.to(DEV), (96, 128))
    got = got_dev.cpu()
    assert (got != ref).float().mean() < 1e-3
    # the merge may restrict each instance's scan to the pixels its pasted mask can reach (box + half a source texel):
    # same map and segments as the full-image scan, also for two images in one launch
    from u2seg_amd.modeling.inference import combine_semantic_and_instance_outputs_batch
    from u2seg_amd.structures import Boxes

    pi = Instances((96, 128))
    pi.pred_masks, pi.pred_boxes = got_dev, Boxes(pb.to(DEV))
    pi.scores = torch.rand(7, generator=g).to(DEV)
    pi.pred_classes = torch.randint(0, 80, (7,), generator=g).to(DEV)
    sem = torch.randint(0, 28, (96, 128), generator=g).to(DEV)
    sem[:40] = 3
    full = combine_semantic_and_instance_outputs_batch([pi, inst], [sem, torch.from_numpy(G["pan_sem"]).to(DEV)], 0.5, 64, 0.3, 0)
    bounded = combine_semantic_and_instance_outputs_batch([pi], [sem], 0.5, 64, 0.3, 28)
    assert torch.equal(full[0][0], bounded[0][0]) and full[0][1] == bounded[0][1] and len(full[0][1]) > 2
    # the golden's second image with the golden's own thresholds, merged in one launch with an unrelated first image
    gold2 = combine_semantic_and_instance_outputs_batch([pi, inst], [sem, torch.from_numpy(G["pan_sem"]).to(DEV)], 0.5, 4096 // 8, 0.5, 0)
    assert np.array_equal(gold2[1][0].cpu().numpy(), G["pan_out"])
    assert full[1][0].shape == (96, 128)

    # 4. full eval forward: semantic argmax vs the bf16 oracle, detection count and field contract
    cfg = get_cfg()
    cfg.merge_from_file(CFG)
    cfg.merge_from_list(["MODEL.DEVICE", DEV, "MODEL.ROI_HEADS.SCORE_THRESH_TEST", 0.0015])
    model = build_model(cfg)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(det_fill(k, v.cpu()).to(DEV))
    model.eval()
    batch = [{k: v for k, v in x.items() if k != "instances"} for x in make_synthetic_batch(2, height=192, width=256, device=DEV)]
    with torch.no_grad():
        out = model(batch)
    om = OracleModel(cfg, {k: v.cpu() for k, v in model.state_dict().items()}, emulate_bf16=True)
    ref = om.inference([{k: v for k, v in x.items() if k != "instances"} for x in make_synthetic_batch(2, height=192, width=256)])
    gold = np.load(os.path.join(ROOT, "tests", "golden", "inference_small.npz"))
    for i, (o, r) in enumerate(zip(out, ref)):
        assert o["sem_seg"].shape == (28, 192, 256) and o["panoptic_seg"][0].dtype == torch.int32
        agree = (o["sem_seg"].argmax(0).cpu() == r["sem_seg"].argmax(0)).float().mean()
        agree_ref = (o["sem_seg"].argmax(0).cpu().numpy() == gold["sem_argmax_%d" % i]).mean()
        assert agree > 0.97 and agree_ref > 0.95, (float(agree), float(agree_ref))
        inst = o["instances"]
        assert len(inst) == len(r["scores"]) == 100 and inst.pred_masks.shape == (100, 192, 256) and inst.pred_masks.dtype == torch.bool

    # 5. detections, teacher-forced: the cascade box heads of the HIP path run on the oracle's FPN maps and the oracle's RPN
    # proposals; the class probabilities [R, 801] and decoded boxes [R, 4] that enter fast_rcnn_inference must agree with the
    # oracle's (random-weight scores sit within 1e-6 of each other, so the free-running top-100 is rounding noise - the inputs
    # of the selection are the meaningful quantity), and the selection itself, run by the oracle on exactly those inputs, must
    # return the detections the HIP path returned.
    import u2seg_amd.modeling.inference as inf_mod
    from u2seg_amd.structures import Boxes

    om.training = False
    cpu_batch = [{k: v for k, v in x.items() if k != "instances"} for x in make_synthetic_batch(2, height=192, width=256)]
    with torch.no_grad():
        images, sizes, _ = om.preprocess(cpu_batch)
        rf = om.backbone(images)
        objs, dlts = om.rpn_head(rf)
        props = om.rpn_proposals(om.anchors(rf), objs, dlts, sizes)
        outs = om.forward_box(rf, props, None)
        ref_probs = sum(torch.softmax(o[0].float(), dim=-1) for o in outs) * (1.0 / 3)
        ref_boxes = O.apply_deltas(outs[-1][1], torch.cat([p["proposal_boxes"] for p in outs[-1][2]]),
                                   cfg.MODEL.ROI_BOX_CASCADE_HEAD.BBOX_REG_WEIGHTS[2])
        plist = []
        for p in props:
            pi = Instances(p["image_size"])
            pi.proposal_boxes, pi.objectness_logits = Boxes(p["proposal_boxes"].to(DEV)), p["objectness_logits"].to(DEV)
            plist.append(pi)
        seen = {}
        orig = inf_mod.fast_rcnn_inference

        def spy(boxes, scores, image_shapes, *a, **kw):
            seen["boxes"], seen["scores"] = [b.clone() for b in boxes], [s_.clone() for s_ in scores]
            seen["out"] = orig(boxes, scores, image_shapes, *a, **kw)
            return seen["out"]

        inf_mod.fast_rcnn_inference = spy
        try:
            model.roi_heads._forward_box({k: nhwc(v) for k, v in rf.items()}, plist)
        finally:
            inf_mod.fast_rcnn_inference = orig
        got_probs, got_boxes = torch.cat(seen["scores"]).cpu(), torch.cat(seen["boxes"]).cpu()
        assert got_probs.shape == ref_probs.shape and got_boxes.shape == ref_boxes.shape
        # three cascaded stages run free on each side (stage k pools at the boxes stage k-1 predicted): 1e-2 relative L2
        assert float((got_probs - ref_probs).norm() / ref_probs.norm()) < 1e-2
        # pixels, after three cascaded decodes of bf16 deltas (2^-8 relative) on boxes up to 256 px wide
        assert float((got_boxes - ref_boxes).abs().max()) < 2.5 and float((got_boxes - ref_boxes).abs().mean()) < 0.1
        counts = [len(p["proposal_boxes"]) for p in props]
        for i, (b_, s_) in enumerate(zip(got_boxes.split(counts), got_probs.split(counts))):
            rb, rs, rc = om.box_inference_single(b_, s_, sizes[i])
            det = seen["out"][0][i]
            assert torch.equal(det.pred_classes.cpu(), rc) and torch.equal(det.scores.cpu(), rs)
            assert torch.equal(det.pred_boxes.tensor.cpu(), rb)


def _rank_device(rank, backend):
    """Two ranks share cuda:0 over gloo (RCCL refuses two ranks on one device); over RCCL ("nccl") every rank owns a GPU."""
    idx = rank if backend == "nccl" else 0
    torch.cuda.set_device(idx)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return "cuda:%d" % idx


def _spawn_two(worker, backend="gloo"):
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(worker, args=(2, port, out, backend), nprocs=2, join=True)
        return out[0], out[1]


needs_two_gpus = pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank; this box has one")


def _two_rank_worker(rank, world, port, out, backend="gloo"):
    import torch.distributed as dist

    DEV = _rank_device(rank, backend)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from tests.golden.make_fixtures import det_fill
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import make_synthetic_batch
    from u2seg_amd.engine import SimpleTrainer
    from u2seg_amd.modeling import build_model
    from u2seg_amd.solver import build_optimizer

    cfg = get_cfg()
    cfg.merge_from_file(CFG)
    cfg.merge_from_list(["MODEL.DEVICE", DEV])
    model = build_model(cfg)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(det_fill(k, v.cpu()).to(DEV))
    model.train()
    trainer = SimpleTrainer(model, build_optimizer(cfg, model))
    torch.manual_seed(100 + rank)
    losses = trainer.run_step(make_synthetic_batch(1, start_index=rank, height=128, width=160, device=DEV))
    torch.cuda.synchronize()
    total = trainer.check_finite()
    flat = trainer.optimizer.flat_param
    sig = torch.stack([flat.double().sum(), flat.double().abs().sum(), model.backbone.bottom_up.stem.conv1.norm.running_mean.double().sum()]).cpu()
    sigs = [torch.zeros_like(sig) for _ in range(world)]
    dist.all_gather(sigs, sig)
    out[rank] = (total, [s.tolist() for s in sigs])
    dist.destroy_process_group()


def _kmeans_shard_worker(rank, world, port, out, backend="gloo"):
    import torch.distributed as dist

    DEV = _rank_device(rank, backend)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from u2seg_amd.cluster import kmeans as KM

    g = np.load(os.path.join(ROOT, "tests", "golden", "kmeans_golden.npz"))
    x, init = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["init"]).to(DEV)
    n = x.shape[0]
    lo, hi = (0, n // 3) if rank == 0 else (n // 3, n)  # uneven shards
    cl, c = KM.kmeans_sharded(x[lo:hi], x[init].clone(), int(g["niter"]))
    out[rank] = (cl.cpu().numpy(), c.cpu().numpy())
    dist.destroy_process_group()


def test_kmeans_row_sharded_two_ranks():
    """k-means with the rows of x sharded over two ranks (one all-reduce of the K*D + K partial sums per iteration):
    concatenated labels and the centroids of both ranks must equal the reference-generated single-process golden."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_kmeans_shard_worker, args=(2, port, out), nprocs=2, join=True)
        (l0, c0), (l1, c1) = out[0], out[1]
    g = np.load(os.path.join(ROOT, "tests", "golden", "kmeans_golden.npz"))
    assert np.array_equal(np.concatenate([l0, l1]), g["labels"])
    np.testing.assert_allclose(c0, g["centroids"], rtol=1e-5, atol=1e-5)
    assert np.array_equal(c0, c1)  # identical on both ranks


def _syncbn_ragged_worker(rank, world, port, out, backend="gloo"):
    import torch.distributed as dist

    DEV = _rank_device(rank, backend)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group(backend, rank=rank, world_size=world)
    from u2seg_amd.layers import functional as F

    g = torch.Generator().manual_seed(50 + rank)
    c = 64
    shape = (2, c, 11, 13) if rank == 0 else (3, c, 17, 9)  # different element counts per rank (286 vs 459)
    x = bf(torch.randn(shape, generator=g) * 2 + 0.3 * rank)
    gy = bf(torch.randn(shape, generator=g))
    gg = torch.Generator().manual_seed(7)  # the parameters are replicated
    gamma, beta = 1 + 0.2 * torch.randn(c, generator=gg), 0.1 * torch.randn(c, generator=gg)
    xd = nhwc(x).to(DEV).requires_grad_(True)
    gd, bd = gamma.to(DEV).requires_grad_(True), beta.to(DEV).requires_grad_(True)
    rm, rv = torch.zeros(c, device=DEV), torch.ones(c, device=DEV)
    stats = torch.stack([x.sum((0, 2, 3)), (x * x).sum((0, 2, 3))]).to(DEV)
    y = F.batch_norm_act(xd, stats, gd, bd, rm, rv, None, True, 0.1, 1e-5, sync=True)
    y.backward(nhwc(gy).to(DEV))
    torch.cuda.synchronize()
    out[rank] = {"x": x, "gy": gy, "y": nchw(y.detach()), "dx": nchw(xd.grad), "dgamma": gd.grad.cpu(), "dbeta": bd.grad.cpu(),
                 "rm": rm.cpu(), "rv": rv.cpu()}
    dist.destroy_process_group()


def test_syncbn_unequal_counts_two_ranks():
    """SyncBN with a different number of elements per rank (ranks pad their batches to their own image sizes): forward,
    running statistics, input gradient and the local affine gradients of both ranks vs nn.functional.batch_norm on the
    CONCATENATED batch (what nn.SyncBatchNorm, selected at layers/batch_norm.py:187, computes by gathering the per-rank
    counts).  Dividing the all-reduced sums by m_local * world - the round-1 code - fails this test."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_syncbn_ragged_worker, args=(2, port, out), nprocs=2, join=True)
        r = [out[0], out[1]]
    c = 64
    g = torch.Generator().manual_seed(7)
    gamma, beta = (1 + 0.2 * torch.randn(c, generator=g)).requires_grad_(True), (0.1 * torch.randn(c, generator=g)).requires_grad_(True)
    flat = [t["x"].permute(0, 2, 3, 1).reshape(-1, c) for t in r]
    xc = torch.cat(flat).clone().requires_grad_(True)
    rm, rv = torch.zeros(c), torch.ones(c)
    yc = TF.relu(TF.batch_norm(xc, rm, rv, gamma, beta, True, 0.1, 1e-5))
    gyc = torch.cat([t["gy"].permute(0, 2, 3, 1).reshape(-1, c) for t in r])
    yc.backward(gyc)
    n0 = flat[0].shape[0]
    for i, t in enumerate(r):
        sl = slice(0, n0) if i == 0 else slice(n0, None)
        got_y = t["y"].permute(0, 2, 3, 1).reshape(-1, c)
        # bounds: measured 3e-5 (y, the bf16 rounding of the fp32 reference is applied on both sides), 3.1e-3 (dx: one bf16 step
        # of the output) and 2e-7 (the affine gradients, fp32 sums) - round 5: tightened from 1e-2 / 1.5e-2 / 1e-2
        assert rel_err(got_y, bf(yc.detach()[sl])) < 1e-3
        got_dx = t["dx"].permute(0, 2, 3, 1).reshape(-1, c)
        assert rel_err(got_dx, xc.grad[sl]) < 5e-3
        # running statistics: identical on both ranks, equal to the concatenated batch's
        assert torch.allclose(t["rm"], rm, atol=1e-4) and torch.allclose(t["rv"], rv, atol=1e-3)
        # affine gradients are local sums (DDP averages them afterwards): this rank's pixels only
        mu, var = xc.detach().mean(0), xc.detach().var(0, unbiased=False)
        xhat = (xc.detach()[sl] - mu) * torch.rsqrt(var + 1e-5)
        dz = gyc[sl] * (yc.detach()[sl] > 0)
        assert rel_err(t["dbeta"], dz.sum(0)) < 1e-5 and rel_err(t["dgamma"], (dz * xhat).sum(0)) < 1e-5
    assert torch.equal(r[0]["rm"], r[1]["rm"]) and torch.equal(r[0]["rv"], r[1]["rv"])


def test_two_rank_step_on_one_gpu():
    """The multi-process data-parallel path (SyncBN statistic all-reduce inside forward/backward, bucketed gradient
    all-reduce, grad_scale = 1/world in the optimizer kernel) with two ranks sharing cuda:0 over gloo (RCCL refuses two
    ranks on one device): both ranks must finish and hold bit-identical parameters and BN running statistics."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_two_rank_worker, args=(2, port, out), nprocs=2, join=True)
        (t0, s0), (t1, s1) = out[0], out[1]
    assert t0 == t0 and t1 == t1
    assert s0[0] == s0[1] == s1[0] == s1[1], (s0, s1)  # identical parameter arena and running stats on both ranks


# ---- the same three exchanges over RCCL (backend "nccl"), one GPU per rank: run wherever two GPUs are visible --------------
@needs_two_gpus
def test_two_rank_step_over_rccl():
    """engine/defaults.py:60-79 (DDP) + layers/batch_norm.py:187 (SyncBN) over the backend the product uses on a node: 122 SyncBN
    statistic all-reduces inside forward / backward, the head-tail gradient all-reduce started from the autograd hook, the
    bucketed rest, grad_scale = 1 / world.  Both ranks must hold bit-identical parameters and running statistics afterwards, and
    the step must equal the gloo run of the same two ranks to 1e-6 relative in those sums (a two-rank sum is order-free; what
    differs from run to run is the order of the split-K gradient atomics inside a rank)."""
    (t0, s0), (t1, s1) = _spawn_two(_two_rank_worker, "nccl")
    assert t0 == t0 and t1 == t1
    assert s0[0] == s0[1] == s1[0] == s1[1], (s0, s1)
    (g0, gs0), _ = _spawn_two(_two_rank_worker, "gloo")
    assert t0 == pytest.approx(g0, rel=1e-3)
    np.testing.assert_allclose(np.array(s0[0]), np.array(gs0[0]), rtol=1e-6)


@needs_two_gpus
def test_syncbn_unequal_counts_over_rccl():
    """The SyncBN exchange with different element counts per rank over RCCL: every output bit-identical to the gloo run (the
    kernels are deterministic and a two-rank sum has one order)."""
    a, b = _spawn_two(_syncbn_ragged_worker, "nccl"), _spawn_two(_syncbn_ragged_worker, "gloo")
    for ra, rb in zip(a, b):
        for k in ("y", "dx", "dgamma", "dbeta", "rm", "rv"):
            assert torch.equal(ra[k], rb[k]), k


@needs_two_gpus
def test_kmeans_row_sharded_over_rccl():
    """Row-sharded k-means (one all-reduce of K * D + K partial sums per iteration) over RCCL: the golden labels, centroids
    bit-identical on both ranks and to the gloo run."""
    (l0, c0), (l1, c1) = _spawn_two(_kmeans_shard_worker, "nccl")
    g = np.load(os.path.join(ROOT, "tests", "golden", "kmeans_golden.npz"))
    assert np.array_equal(np.concatenate([l0, l1]), g["labels"]) and np.array_equal(c0, c1)
    (_, cg), _ = _spawn_two(_kmeans_shard_worker, "gloo")
    assert np.array_equal(c0, cg)


def test_edge_cases_empty_and_ragged_batches(F):
    """Edge cases the reference exercises in tests/modeling/test_model_e2e.py:103-154: an image without instances
    (half-empty and fully empty batches), images of different sizes in one batch, and a single tiny image."""
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import make_synthetic_batch
    from u2seg_amd.modeling import build_model
    from u2seg_amd.structures import BitMasks, Boxes, Instances

    cfg = get_cfg()
    cfg.merge_from_file(CFG)
    cfg.merge_from_list(["MODEL.DEVICE", DEV])
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()

    def empty_like(sample):
        h, w = sample["image"].shape[-2:]
        inst = Instances((h, w))
        inst.gt_boxes = Boxes(torch.zeros((0, 4), device=DEV))
        inst.gt_classes = torch.zeros(0, dtype=torch.int64, device=DEV)
        inst.gt_masks = BitMasks(torch.zeros((0, h, w), dtype=torch.bool, device=DEV))
        out = dict(sample)
        out["instances"] = inst
        return out

    a = make_synthetic_batch(1, height=160, width=224, device=DEV)[0]
    b = make_synthetic_batch(1, start_index=1, height=128, width=192, device=DEV)[0]  # ragged: padded to 160x224
    for batch in ([a, empty_like(b)], [empty_like(a), empty_like(b)], [b]):
        model.zero_grad()
        losses = model(batch)
        assert len(losses) == 10
        total = sum(losses.values())
        total.backward()
        assert bool(torch.isfinite(total)), {k: float(v) for k, v in losses.items()}
        g = model.backbone.bottom_up.res2[0].conv1.weight.grad
        assert g is not None and bool(torch.isfinite(g).all())
    # fully empty batch: no foreground anywhere -> mask and box-regression losses are exactly zero
    losses = model([empty_like(a), empty_like(b)])
    assert float(losses["loss_mask"]) == 0.0 and float(losses["loss_box_reg_stage0"]) == 0.0
    model.eval()
    with torch.no_grad():
        out = model([{"image": a["image"]}, {"image": b["image"], "height": 256, "width": 384}])
    assert out[0]["sem_seg"].shape == (28, 160, 224) and out[1]["sem_seg"].shape == (28, 256, 384)
    assert out[1]["panoptic_seg"][0].shape == (256, 384)


def test_real_data_pipeline_to_model(F):
    """Small committed dataset -> builtin registration -> DatasetMapper in a loader -> DevicePrefetcher (pinned staging, side
    stream) -> two optimizer steps: the batches arriving in HBM equal the host-side mapper output bit for bit, images of
    different sizes and an image whose only annotation is a crowd region (no gt_masks field) train to finite losses."""
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import (DatasetCatalog, DevicePrefetcher, MetadataCatalog, build_detection_train_loader,
                                register_all_coco)
    from u2seg_amd.engine.trainer import SimpleTrainer
    from u2seg_amd.modeling import build_model
    from u2seg_amd.solver import build_lr_scheduler, build_optimizer

    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "data_golden.json")))
    os.environ["CLUSTER_NUM"] = "800"
    for name in list(DatasetCatalog.keys()):
        DatasetCatalog.remove(name)
    for name in list(MetadataCatalog.keys()):
        MetadataCatalog.remove(name)
    register_all_coco(os.path.join(ROOT, "tests", "golden", "data_small"))
    cfg = get_cfg()
    cfg.merge_from_file(CFG)
    cfg.merge_from_list(fx["input_opts"] + ["DATALOADER.NUM_WORKERS", 2, "SOLVER.IMS_PER_BATCH", 2, "MODEL.DEVICE", DEV,
                                           "DATALOADER.ASPECT_RATIO_GROUPING", False])
    np.random.seed(0)
    host = list(itertools.islice(iter(build_detection_train_loader(cfg, seed=3)), 4))
    dev = list(itertools.islice(iter(DevicePrefetcher(build_detection_train_loader(cfg, seed=3), DEV)), 4))
    ids = []
    for hb, db in zip(host, dev):
        for h, d in zip(hb, db):
            # worker processes draw their own augmentation parameters, so compare what does not depend on them
            assert h["image_id"] == d["image_id"] and (h["height"], h["width"]) == (d["height"], d["width"])
            assert d["image"].is_cuda and d["sem_seg"].is_cuda and d["instances"].gt_boxes.tensor.is_cuda
            assert d["image"].dtype == torch.uint8 and d["image"].shape[1:] == tuple(d["instances"].image_size)
            ids.append(d["image_id"])
    assert 30 in ids  # the crowd-only image is part of the stream
    # single-process loader: the device copy must equal the host batch exactly
    cfg0 = cfg.clone()
    cfg0.merge_from_list(["DATALOADER.NUM_WORKERS", 0])
    np.random.seed(1)
    host = list(itertools.islice(iter(build_detection_train_loader(cfg0, seed=3)), 3))
    np.random.seed(1)
    dev = list(itertools.islice(iter(DevicePrefetcher(build_detection_train_loader(cfg0, seed=3), DEV)), 3))
    for hb, db in zip(host, dev):
        for h, d in zip(hb, db):
            assert torch.equal(h["image"], d["image"].cpu()) and torch.equal(h["sem_seg"], d["sem_seg"].cpu())
            assert torch.equal(h["instances"].gt_boxes.tensor, d["instances"].gt_boxes.tensor.cpu())
            assert h["instances"].has("gt_masks") == d["instances"].has("gt_masks")
            if h["instances"].has("gt_masks"):
                assert torch.equal(h["instances"].gt_masks.tensor, d["instances"].gt_masks.tensor.cpu())
    # the shared-memory slot form (what worker processes send): packed on the host, rebuilt on the device from one block
    from u2seg_amd.data.slots import BatchPacker, SlotRing

    ring = SlotRing(1, 8 << 20, 3)
    pre = DevicePrefetcher([], DEV)
    pre.ring = ring
    for hb in host:
        moved, done = pre._stage_packed(BatchPacker(ring)(hb))
        done.synchronize()
        for h, d in zip(hb, moved):
            assert torch.equal(h["image"], d["image"].cpu()) and torch.equal(h["sem_seg"], d["sem_seg"].cpu())
            assert d["sem_seg"].dtype == torch.int64 and d["image"].is_cuda
            assert torch.equal(h["instances"].gt_boxes.tensor, d["instances"].gt_boxes.tensor.cpu())
            assert torch.equal(h["instances"].gt_classes, d["instances"].gt_classes.cpu())
            if h["instances"].has("gt_masks"):
                assert d["instances"].gt_masks.tensor.dtype == torch.bool
                assert torch.equal(h["instances"].gt_masks.tensor, d["instances"].gt_masks.tensor.cpu())
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    trainer = SimpleTrainer(model, opt, build_lr_scheduler(cfg, opt))
    for batch in dev:
        losses = trainer.run_step(batch)
        assert len(losses) == 10 and bool(torch.isfinite(sum(v.detach() for v in losses.values())))


@pytest.mark.parametrize("branch", ["per_image_permutations", "batched_keys"])
def test_sgd_trajectory_vs_reference(F, fixed_order_statistics, branch):
    """Four training steps through the product path (HIP model, FlatSGD arena + u2_sgd_clip_step, WarmupMultiStepLR,
    SimpleTrainer) against the reference's own four steps (tests/golden/trajectory_small.json: its PanopticFPN, its
    clip-wrapped SGD, its LR schedule, fp32 CPU): the lr of every step exactly, the dense losses of every step within the
    bf16 band of test_whole_model_vs_oracle, then where the parameters and BN running statistics ended up.
    "per_image_permutations" replays the reference's randperm stream through the per-image sampling branch;
    "batched_keys" takes the batched branch of every real training step (its random draws are then not the fixture's: the
    sampled anchors / ROIs are a different uniformly random subset, which the bands below already absorb)."""
    from tests.golden.make_fixtures import det_fill
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import make_synthetic_batch
    from u2seg_amd.engine.trainer import SimpleTrainer
    from u2seg_amd.modeling import build_model, set_permutation_source
    from u2seg_amd.solver import build_lr_scheduler, build_optimizer

    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "trajectory_small.json")))
    cfg = get_cfg()
    cfg.merge_from_file(CFG)
    cfg.merge_from_list(["MODEL.DEVICE", DEV] + fx["overrides"])
    model = build_model(cfg)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(det_fill(k, v.cpu()).to(DEV))
    model.train()
    init = {k: dict(model.named_parameters())[k].detach().clone() for k in fx["param_norm"]}
    opt = build_optimizer(cfg, model)
    trainer = SimpleTrainer(model, opt, build_lr_scheduler(cfg, opt))
    if branch == "per_image_permutations":
        set_permutation_source(lambda n, device=None: torch.randperm(n))
    else:
        from tests.test_gpu_bookkeeping import KeyRecorder
        from u2seg_amd.modeling import set_key_source

        set_key_source(KeyRecorder(fx["seed"]))
    torch.manual_seed(fx["seed"])
    n, (h, w) = fx["num_images"], fx["image_hw"]
    report = []
    try:
        for it in range(fx["steps"]):
            assert opt.lr == pytest.approx(fx["lr"][it], rel=1e-12), it
            losses = trainer.run_step(make_synthetic_batch(n, height=h, width=w, start_index=it * n, device=DEV))
            report.append({k: (float(v.detach()), fx["losses"][it][k]) for k, v in losses.items()})
    finally:
        set_permutation_source(None)
        if branch != "per_image_permutations":
            set_key_source(None)
    print(json.dumps(report, indent=1))
    # Step 0 sees the reference's parameters: the single-step band of test_whole_model_vs_oracle.  Afterwards the bf16 and
    # fp32 runs drift apart and sampling decisions differ; the bf16-emulating oracle run against the same fixture (with
    # 0 / 1e-6 / 3e-6 relative parameter noise per step, standing in for the run-to-run order of split-K accumulation)
    # deviates by up to 2.1 % at step 1 and 5.7 % at steps 2-3 (loss_rpn_cls, 256 sampled anchors per image, is the noisy
    # one; the others stay within 3 %), totals within 0.6 %.  Bands: 4 % / 6 % / 12 % per loss, 2 % / 3 % / 5 % on the total.
    for it, row in enumerate(report):
        band, total_band = ((4e-2, 2e-2), (6e-2, 3e-2), (0.12, 5e-2))[min(it, 2)]
        for k in ("loss_sem_seg", "loss_rpn_cls", "loss_cls_stage0", "loss_cls_stage1", "loss_cls_stage2", "loss_mask"):
            assert row[k][0] == pytest.approx(row[k][1], rel=band), (it, k, row)
        total = sum(v[0] for v in row.values())
        assert total == pytest.approx(sum(v[1] for v in row.values()), rel=total_band), (it, row)
    params = dict(model.named_parameters())
    disp = {}
    for k in fx["param_norm"]:
        assert float(params[k].double().norm()) == pytest.approx(fx["param_norm"][k], rel=1e-3), k  # |disp| / |p| is ~3e-3
        disp[k] = (float((params[k].detach() - init[k]).double().norm()), fx["param_delta_norm"][k])
    print(json.dumps(disp, indent=1))
    for k, (got, want) in disp.items():
        # every tensor is clipped to unit gradient norm, so its displacement is set by the lr schedule and the
        # step-to-step alignment of the gradient directions; bf16 noise leaves that within 25 % (measured 0.01-8 %)
        assert got == pytest.approx(want, rel=0.25), (k, disp)
    sd = model.state_dict()
    for k, v in fx["running_mean_norm"].items():
        assert float(sd[k].double().norm()) == pytest.approx(v, rel=5e-2), k
    assert int(sd["backbone.bottom_up.stem.conv1.norm.num_batches_tracked"]) == fx["num_batches_tracked"] == fx["steps"]


def test_semantic_head_launched_in_pieces_equals_whole_launch(F, fixed_order_statistics, monkeypatch):
    """PanopticFPN.forward (training) hands the semantic head to the ROI heads in pieces that are launched - on the head's own
    stream - from inside the samplers' torch.no_grad() regions (layers/functional.py:defer_pieces, meta_arch/panoptic_fpn.py:90-138
    has one call): same weights, batch and sampling keys with U2_SEM_PIECES=1 / 0 give the same ten losses and the same
    gradients - in particular the head's and, through the FPN maps, the backbone's (a piece launched without a graph would
    leave them at zero / without the head's share)."""
    from tests.golden.make_fixtures import det_fill
    from tests.test_gpu_bookkeeping import KeyRecorder
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import make_synthetic_batch
    from u2seg_amd.modeling import build_model, set_key_source
    from u2seg_amd.solver import build_optimizer

    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "trajectory_small.json")))
    cfg = get_cfg()
    cfg.merge_from_file(CFG)
    cfg.merge_from_list(["MODEL.DEVICE", DEV] + fx["overrides"])
    model = build_model(cfg)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            v.copy_(det_fill(k, v.cpu()).to(DEV))
    model.train()
    opt = build_optimizer(cfg, model)
    n, (h, w) = fx["num_images"], fx["image_hw"]
    batch = make_synthetic_batch(n, height=h, width=w, device=DEV)
    names = {id(p): k for k, p in model.named_parameters()}

    def run(mode):
        monkeypatch.setenv("U2_SEM_PIECES", mode)
        set_key_source(KeyRecorder(fx["seed"]))
        try:
            opt.zero_grad()
            losses = model(batch)
            assert not F._deferred_pieces
            sum(losses.values()).backward()
        finally:
            set_key_source(None)
        F.assert_no_deferred_gradients()
        F.join_all_streams()
        torch.cuda.synchronize()
        return {k: float(v.detach()) for k, v in losses.items()}, opt.flat_grad.clone()

    run("0")  # first use of the streams' scratch
    (l1, g1), (l0, g0) = run("1"), run("0")
    assert list(l1) == list(l0) and len(l1) == 10
    for k in l0:
        assert l1[k] == pytest.approx(l0[k], rel=2e-3, abs=1e-5), (k, l1, l0)
    worst = {}
    for p, off in zip(opt.params, opt.param_offset):
        a, b = g1[off:off + p.numel()].double(), g0[off:off + p.numel()].double()
        group = names[id(p)].split(".")[0]
        rel = float((a - b).norm()) / max(float(b.norm()), 1e-20)
        if names[id(p)].startswith("sem_seg_head"):
            assert float(a.norm()) > 0 and float(b.norm()) > 0, names[id(p)]
        worst[group] = max(worst.get(group, 0.0), rel)
    print(worst)
    # the backward passes differ in the order of fp32 atomic accumulation only (measured: 1e-3 ... 1e-2 on the smallest tensors)
    assert float((g1 - g0).double().norm()) <= 2e-2 * float(g0.double().norm()), worst
    assert worst["sem_seg_head"] <= 5e-2, worst


def test_sem_seg_postprocess_resize(F):
    """modeling/postprocessing.py:77-100: the crop to the image size + bilinear resize to the requested output size
    (u2_bilinear_resize_f32 reads the cropped window in place) vs ATen's interpolate, up- and down-scaling."""
    from u2seg_amd.modeling.inference import sem_seg_postprocess

    g = torch.Generator().manual_seed(3)
    full = torch.randn((28, 160, 224), generator=g).to(DEV)
    for img, out in (((150, 200), (225, 300)), ((160, 224), (97, 133)), ((33, 47), (160, 224)), ((160, 224), (160, 224))):
        got = sem_seg_postprocess(full, img, *out)
        ref = torch.nn.functional.interpolate(full[:, : img[0], : img[1]][None], size=out, mode="bilinear", align_corners=False)[0]
        # fp32 on both sides; ATen's kernel is compiled with fp contraction, this one without: a few 1e-5 on values of order 1
        assert got.shape == ref.shape and float((got - ref).abs().max()) < 1e-4, (img, out)


def test_fpn_lateral_upsample_fusion_is_bit_identical(F):
    """backbone/fpn.py:141-158 fused (u2_affine_upadd inside the lateral's BatchNorm apply) vs the two separate passes
    (BatchNorm apply, then nearest x2 + add): the same bits forward, the same gradients for the lateral's input, the coarser
    level and the affine parameters."""
    from u2seg_amd.layers.modules import BatchNorm2d, Conv2d

    g = torch.Generator().manual_seed(4)
    conv = Conv2d(64, 96, kernel_size=1, bias=False, norm=BatchNorm2d(96)).to(DEV).train()
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * 0.2)
        conv.norm.weight.copy_(1 + 0.3 * torch.randn(96, generator=g))
        conv.norm.bias.copy_(0.2 * torch.randn(96, generator=g))
    x = nhwc(torch.randn((2, 64, 12, 20), generator=g))
    top = nhwc(torch.randn((2, 96, 6, 10), generator=g))
    gy = nhwc(torch.randn((2, 96, 12, 20), generator=g))
    outs = []
    for fused in (True, False):
        xi, ti = x.clone().requires_grad_(True), top.clone().requires_grad_(True)
        conv.zero_grad()
        y = conv(xi, residual_up=ti) if fused else F.fpn_upsample_add(conv(xi), ti)
        y.backward(gy)
        outs.append((y.detach().clone(), xi.grad.clone(), ti.grad.clone(), conv.norm.weight.grad.clone(), conv.norm.bias.grad.clone(),
                     conv.weight.grad.clone()))
    for i, (a, b) in enumerate(zip(*outs)):
        if i in (0, 2):  # the fused output and the coarser level's gradient (2x2 sums of the incoming gradient): the same bits
            assert torch.equal(a, b), i
        elif i == 1:     # behind the BatchNorm backward, whose column sums are fp32 atomics (order-dependent last bits)
            assert rel_err(a.float().cpu(), b.float().cpu()) < 1e-2, i
        else:            # parameter gradients: fp32 atomics as well
            assert rel_err(a.float().cpu(), b.float().cpu()) < 1e-2, i
    ref = bf(bf(TF.batch_norm(bf(TF.conv2d(nchw(x, 64), bf(conv.weight.detach().cpu()))), None, None, conv.norm.weight.detach().cpu(),
                               conv.norm.bias.detach().cpu(), True, 0.1, 1e-5)) + TF.interpolate(nchw(top, 96), scale_factor=2.0, mode="nearest"))
    assert rel_err(nchw(outs[0][0], 96), ref) < 1e-2


def test_r50_300_config_builds_and_steps(F):
    """BASELINE.json configuration 4's model file (u2seg_R50_300.yaml: 300 pseudo classes) through the HIP path: build, one
    training step with finite losses and the 300-way heads, one inference call."""
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import make_synthetic_batch
    from u2seg_amd.engine import SimpleTrainer
    from u2seg_amd.modeling import build_model
    from u2seg_amd.solver import build_optimizer

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "COCO-PanopticSegmentation", "u2seg_R50_300.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", DEV])
    assert cfg.MODEL.ROI_HEADS.NUM_CLASSES == 300
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    assert model.roi_heads.box_predictor[0].cls_score.weight.shape[0] == 301
    assert model.roi_heads.mask_head.predictor.weight.shape[0] == 300
    trainer = SimpleTrainer(model, build_optimizer(cfg, model))
    losses = trainer.run_step(make_synthetic_batch(2, height=256, width=320, num_thing_classes=300, device=DEV))
    assert len(losses) == 10 and trainer.check_finite() > 0
    model.eval()
    with torch.no_grad():
        out = model([{k: v for k, v in x.items() if k != "instances"} for x in make_synthetic_batch(1, height=256, width=320, device=DEV)])
    assert out[0]["sem_seg"].shape[1:] == (256, 320)
    cls = out[0]["instances"].pred_classes
    assert cls.numel() == 0 or int(cls.max()) < 300


@pytest.mark.parametrize("shape", [(2, 48, 64, 64, 54), (1, 200, 336, 64, 54), (3, 7, 5, 32, 3)])
def test_sem_seg_inference_upsample_and_argmax(F, shape):
    """meta_arch/semantic_seg.py:240-244 (F.interpolate x4, bilinear, align_corners=False) and panoptic_fpn.py:173
    (argmax over classes) in one kernel: logits to 1e-5 of ATen's fp32 resampling of the same bf16 map; argmax equal
    wherever ATen's top-2 margin is above the resampling round-off."""
    b, h, w, cp, k = shape
    g = torch.Generator(device="cuda").manual_seed(5)
    x = (torch.randn(b, h, w, cp, device="cuda", generator=g) * 3).to(torch.bfloat16)
    ref = TF.interpolate(x[..., :k].permute(0, 3, 1, 2).float(), scale_factor=4, mode="bilinear", align_corners=False)
    out, amax = F.sem_seg_upsample(x, k, 4)
    assert out.shape == ref.shape and amax.shape == (b, h * 4, w * 4) and amax.dtype == torch.int64
    assert (out - ref).abs().max().item() < 1e-5
    assert torch.equal(amax, out.argmax(dim=1))  # the kernel's argmax is the argmax of the logits it wrote
    top2 = ref.topk(min(2, k), dim=1).values
    decided = (top2[:, 0] - top2[:, -1]) > 1e-4 if k > 1 else torch.ones_like(amax, dtype=torch.bool)
    assert torch.equal(amax[decided], ref.argmax(dim=1)[decided])
    assert decided.float().mean().item() > 0.99


@pytest.mark.parametrize("handles", [2, 3])
def test_gradient_handles_sum_in_kernel(F, handles):
    """A tensor with several consumers: the BatchNorm block tail hands out 2 / 3 autograd handles whose gradients meet
    inside its backward kernel (resnet.py:204-210 + fpn.py:141-146 read a stage output three times), and F.fan_out sums k
    gradients in one kernel; both against plain autograd accumulation on one handle."""
    g = torch.Generator(device="cuda").manual_seed(11 + handles)
    y = (torch.randn((2, 9, 11, 64), device=DEV, generator=g) * 2).bfloat16()
    res = torch.randn((2, 9, 11, 64), device=DEV, generator=g).bfloat16()
    gamma = (1 + 0.2 * torch.randn(64, device=DEV, generator=g))
    beta = 0.1 * torch.randn(64, device=DEV, generator=g)
    ws = [torch.randn((2, 9, 11, 64), device=DEV, generator=g).bfloat16() for _ in range(handles)]
    yf = y.float().reshape(-1, 64)
    stats = torch.stack([yf.sum(0), (yf * yf).sum(0)])

    def run(twin):
        yd, rd = y.clone().requires_grad_(True), res.clone().requires_grad_(True)
        gd, bd = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        out = F.batch_norm_act(yd, stats.clone(), gd, bd, torch.zeros(64, device=DEV), torch.ones(64, device=DEV), rd, True,
                               0.1, 1e-5, twin=twin)
        hs = [out] * handles if not twin else ([out, out._u2_twin] + ([out._u2_third] if handles == 3 else []))
        sum((h * w).float().sum() for h, w in zip(hs, ws)).backward()
        return yd.grad, rd.grad, gd.grad, bd.grad

    ref = run(False)
    got = run(3 if handles == 3 else True)
    for a, b in zip(got, ref):
        assert rel_err(a.float(), b.float()) < 1e-2

    def run_fan(use):
        x = y.clone().requires_grad_(True)
        hs = F.fan_out(x, handles) if use else (x,) * handles
        sum((h * w).float().sum() for h, w in zip(hs, ws)).backward()
        return x.grad

    assert rel_err(run_fan(True).float(), run_fan(False).float()) < 1e-2


def test_multi_stream_step_equals_single_stream(F, monkeypatch):
    """The semantic head on a second stream and the weight gradients on a side stream must not change what is computed.
    The FPN maps are frozen (computed once, fed back as leaves) so that the comparison is not drowned by the run-to-run
    noise of 50 train-mode BN layers (fp32 atomics land in a different order every run; on the full network that alone
    moves single-stream gradients by > 100 % on some parameters): from identical maps, weights and batch, the semantic
    loss, the semantic head's parameter gradients (they pass through both extra streams) and the gradients it sends into
    the FPN maps (bf16) must equal the single-stream run to 1e-2 / a few bf16 ulp, three times in a row."""
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import make_synthetic_batch
    from u2seg_amd.modeling import build_model
    from u2seg_amd.solver import build_optimizer

    cfg = get_cfg()
    cfg.merge_from_file(CFG)
    cfg.merge_from_list(["MODEL.DEVICE", DEV])
    torch.manual_seed(3)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    batch = make_synthetic_batch(4, height=384, width=512, device=DEV)
    names = [n for n, _ in model.named_parameters()]
    sem = [i for i, n in enumerate(names) if n.startswith("sem_seg_head.")]
    assert len(sem) > 10
    with torch.no_grad():
        feats, sizes, hw = model._backbone_features(batch)
    leaves = {}

    def frozen(_batched_inputs):
        leaves.clear()
        leaves.update({k: v.detach().clone().requires_grad_(True) for k, v in feats.items()})
        return dict(leaves), sizes, hw

    monkeypatch.setattr(model, "_backbone_features", frozen)

    def run(multi):
        monkeypatch.setattr(F, "_WGRAD_SIDE", multi)
        monkeypatch.setenv("U2_AUX_STREAM", "1" if multi else "0")
        opt.zero_grad()
        torch.manual_seed(11)
        losses = model(batch)
        losses["loss_sem_seg"].backward()
        torch.cuda.synchronize()
        return (float(losses["loss_sem_seg"]), [opt.params[i].grad.detach().clone() for i in sem],
                {k: v.grad.detach().clone() for k, v in leaves.items() if v.grad is not None})

    ref_loss, ref_p, ref_f = run(False)
    assert sum(float(g.abs().sum()) for g in ref_p) > 0 and len(ref_f) >= 4
    for _ in range(3):
        loss, got_p, got_f = run(True)
        assert loss == pytest.approx(ref_loss, rel=1e-3)
        for i, a, b in zip(sem, got_p, ref_p):
            assert rel_err(a, b) < 1e-2, names[i]
        for k in ref_f:
            assert rel_err(got_f[k].float(), ref_f[k].float()) < 5e-2, k  # bf16 maps: a few ulp of the largest element
