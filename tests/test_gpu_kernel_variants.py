"""Every kernel variant of the convolution family under `pytest -m gpu`, including the ones the dispatch heuristics only
pick at the shapes of the benchmark configuration (u2seg_R50_800, batch 16, 800x1333).

Two kinds of test:
  * forced variants on small shapes: the product path passes variant 0, so the tests steer the launchers through
    U2_CONV_VARIANT / U2_WGRAD_VARIANT (same bits as the C-ABI's variant argument) and check with u2_conv_last_kernel that
    the intended kernel really ran - every template instantiation of conv_igemm.hip and conv_tile.hip gets a forward +
    data-gradient + weight-gradient comparison with fp32 F.conv2d on the same bf16-rounded operands;
  * true benchmark shapes through the AUTOMATIC dispatch (no override): the fpn_output2 layer (16 x 200 x 336, 3x3
    256 -> 256; non-temporal output stores, the deepest LDS ring, XCD-grouped wgrad) forward / dgrad / wgrad and a few
    more layers forward, against an fp32 reference evaluated at sampled output positions.
"""
import contextlib
import os

import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def F():
    assert torch.cuda.is_available(), "these tests need the GPU"
    from u2seg_amd import _hip
    from u2seg_amd.layers import functional

    _hip.load()
    return functional


def bf(x):
    return x.bfloat16().float()


def nhwc(x_nchw):
    b, c, h, w = x_nchw.shape
    cp = (c + 31) // 32 * 32
    out = torch.zeros((b, h, w, cp), dtype=torch.bfloat16, device=DEV)
    out[..., :c] = x_nchw.permute(0, 2, 3, 1).to(DEV)
    return out


def nchw(x_nhwc, c):
    return x_nhwc[..., :c].permute(0, 3, 1, 2).float().cpu()


def rel_err(a, b):
    return float((a.detach() - b.detach()).abs().max() / (b.detach().abs().max() + 1e-12))


@contextlib.contextmanager
def forced(conv=None, wgrad=None):
    old = {k: os.environ.get(k) for k in ("U2_CONV_VARIANT", "U2_WGRAD_VARIANT")}
    try:
        for k, v in (("U2_CONV_VARIANT", conv), ("U2_WGRAD_VARIANT", wgrad)):
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        # the kernel tests see the library's own dispatch: the product path's rule of where the stream-K form may run
        # (layers/functional.py:streamk_region - only the bottom-up backbone asks for it) is tested in test_streamk_region_rule
        from u2seg_amd.layers import functional as _fn
        with _fn.streamk_region():
            yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


# One bf16 rounding of the result: the HIP output is the correctly rounded fp32 sum (up to accumulation-order noise), so against
# the UNROUNDED fp32 reference it is off by at most half a bf16 step, 2^-9 of the element and < 4e-3 of the largest one.
ULP = 4e-3
# Weight gradients: the HIP kernels keep the fp32 sum of bf16 products; the fp32 reference below differentiates through
# bf(w) = w.bfloat16().float(), whose backward rounds the gradient to bf16 (as the reference's autocast does for the gradient of
# a bf16 weight copy) - the same half-step bound applies, with the HIP side the more precise one.
WG_TOL = 4e-3


def last_kernel():
    from u2seg_amd import _hip

    return _hip.call_nostream("u2_conv_last_kernel")


def igemm_code(bk, tm, tn, nst, glds=1):
    return 1000000 + bk * 10000 + (tm // 64) * 1000 + (tn // 64) * 100 + nst * 10 + glds


NEVER_TILE = 15 << 12
# (U2_CONV_VARIANT, expected u2_conv_last_kernel of the FORWARD launch) on the 3x3 64 -> 256 case below (K = 576)
CONV_VARIANTS = [
    (NEVER_TILE, igemm_code(64, 128, 128, 2)),            # 128 x 128, BK 64, 2-stage LDS-DMA ring (the default of that family)
    (NEVER_TILE | 1, igemm_code(64, 128, 128, 2, 0)),     # register-staged operands
    (NEVER_TILE | 32, igemm_code(64, 128, 128, 3)),       # 3-stage ring
    (NEVER_TILE | 48, igemm_code(64, 128, 128, 4)),       # 4-stage ring
    (NEVER_TILE | 4, igemm_code(32, 128, 128, 4)),        # BK 32 (default ring depth 4)
    (NEVER_TILE | 4 | 16, igemm_code(32, 128, 128, 2)),
    (NEVER_TILE | 4 | 32, igemm_code(32, 128, 128, 3)),
    (NEVER_TILE | 4 | 1, igemm_code(32, 128, 128, 2, 0)),
    (NEVER_TILE | 8, igemm_code(64, 256, 128, 2)),        # 256 x 128 tile, 8 waves
    (NEVER_TILE | 8 | 1, igemm_code(64, 256, 128, 2, 0)),
    (NEVER_TILE | 8 | 32, igemm_code(64, 256, 128, 3)),
    (NEVER_TILE | 8 | 4, igemm_code(32, 256, 128, 4)),
    (NEVER_TILE | 8 | 4 | 16, igemm_code(32, 256, 128, 2)),
    (NEVER_TILE | 8 | 4 | 32, igemm_code(32, 256, 128, 3)),
    (NEVER_TILE | 8 | 4 | 1, igemm_code(32, 256, 128, 2, 0)),
    (NEVER_TILE | 256, 256),                              # conv_igemm256_kernel<false>
    (NEVER_TILE | 256 | 1024, 256 + 1024),                # conv_igemm256_kernel<true> (staggered wave groups)
    (NEVER_TILE | 256 | 1024 | 2048, 256 + 1024),
] + [((cfg << 12) | (tiny << 16), 100 + cfg) for cfg in range(1, 7) for tiny in (0, 1)] + [
    # stream-K form (variant bit 27), 24 work-groups: shares begin and end inside tiles, partial tiles are handed over
    ((cfg << 12) | (1 << 16) | (1 << 27), 500 + cfg) for cfg in (1, 2, 3, 4)] + [
    (1 << 24, 300), ((1 << 24) | (1 << 16), 300),         # conv_halo.hip (halo-staged 3x3), 12 tiles / 8 work-groups
]


@pytest.mark.parametrize("variant,code", CONV_VARIANTS)
def test_conv_forced_variant_fwd_bwd(F, variant, code):
    """3x3 64 -> 256 conv with BN statistics on 2 x 36 x 32 (M = 2304: 9 tiles of 256 pixels, so that the persistent
    kernels with the 8-work-group grid walk two tiles): forward, statistics, data gradient (a 3x3 256 -> 64 conv through the
    same launcher) and weight gradient vs fp32 F.conv2d."""
    g = torch.Generator().manual_seed(variant % 9973)
    cin, cout = 64, 256
    x = bf(torch.randn((2, cin, 36, 32), generator=g))
    w = (torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    yr = TF.conv2d(xr, bf(w), None, 1, 1)  # fp32, not rounded: see ULP
    gy = bf(torch.randn(yr.shape, generator=g))
    yr.backward(gy)
    xd = nhwc(x).requires_grad_(True)
    wd = w.detach().to(DEV).requires_grad_(True)
    with forced(conv=variant):
        y, stats = F._Conv2dFn.apply(xd, wd, None, 1, 1, False, True)
        assert last_kernel() == code, "variant %#x ran kernel %d, expected %d" % (variant, last_kernel(), code)
        y.backward(nhwc(gy))
    yy = nchw(y, cout)
    assert rel_err(yy, yr.detach()) < ULP
    assert torch.allclose(stats[0].cpu(), yy.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(stats[1].cpu(), (yy * yy).sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    assert rel_err(nchw(xd.grad, cin), xr.grad) < ULP
    assert rel_err(wd.grad.cpu(), w.grad) < WG_TOL


@pytest.mark.parametrize("tiny", [0, 1])
@pytest.mark.parametrize("cin,cout", [(64, 64), (96, 40), (256, 64), (32, 8)])
def test_conv_halo_64_channel_tiles(F, cin, cout, tiny):
    """conv_halo_kernel<16, 32, 3, 1> (<= 64 output channels: all eight waves along the pixels, four pixel fragments each):
    ragged patches in both directions and a ragged channel tail, statistics / bias + ReLU, forward and both gradients vs fp32;
    bit 17 puts the same launch on the 128-channel tiles (code 300) and the two must agree to the bit."""
    g = torch.Generator().manual_seed(cin * 100 + cout + tiny)
    x = bf(torch.randn((2, cin, 37, 45), generator=g))
    w = (torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5).requires_grad_(True)
    bias = torch.randn(cout, generator=g) * 0.1
    xr = x.clone().requires_grad_(True)
    yr = TF.conv2d(xr, bf(w), None, 1, 1)
    gy = bf(torch.randn(yr.shape, generator=g))
    yr.backward(gy)
    xd = nhwc(x).requires_grad_(True)
    wd = w.detach().to(DEV).requires_grad_(True)
    with forced(conv=(1 << 24) | (tiny << 16)):
        y, stats = F._Conv2dFn.apply(xd, wd, None, 1, 1, False, True)
        assert last_kernel() == 301, last_kernel()
        yb = F.conv2d(nhwc(x), w.detach().to(DEV), bias.to(DEV), 1, 1, relu=True)
        assert last_kernel() == 301
        y.backward(nhwc(gy))
    with forced(conv=(1 << 24) | (1 << 17) | (tiny << 16)):
        y128, stats128 = F._Conv2dFn.apply(nhwc(x), wd.detach(), None, 1, 1, False, True)
        assert last_kernel() == 300, last_kernel()
    assert torch.equal(y, y128)
    yy = nchw(y, cout)
    assert rel_err(yy, yr.detach()) < ULP
    assert torch.allclose(stats[0].cpu(), yy.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(stats[1].cpu(), (yy * yy).sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(stats[0], stats128[0], rtol=1e-5, atol=1e-3)
    ref_b = torch.relu(TF.conv2d(x, bf(w.detach()), bias, 1, 1))
    assert rel_err(nchw(yb, cout), ref_b) < ULP
    assert rel_err(nchw(xd.grad, cin), xr.grad) < ULP
    assert rel_err(wd.grad.cpu(), w.grad) < WG_TOL


STREAM = 1 << 17   # conv_stream.hip wherever it applies; bit 16: 8 work-groups per 256-channel block (many tiles per group)


@pytest.mark.parametrize("tiny", [0, 1])
@pytest.mark.parametrize("cout", [40, 104, 264, 520])
@pytest.mark.parametrize("cin", [32, 64, 128, 256])
def test_conv_stream_kernel(F, cin, cout, tiny):
    """conv_stream_kernel<KS, TM, PSW, NWV> (1x1, weights resident in registers, whole pixel rows through the LDS ring): all
    twelve instantiations (C = 32 / 64 / 128 / 256 x 1 / 2 / 4 waves along the pixels; N = 520: three 256-channel blocks), a
    ragged last pixel tile and a ragged last channel block, with BN statistics and with bias + ReLU, forward and data
    gradient (the transposed 1x1) vs fp32."""
    g = torch.Generator().manual_seed(cin * 1000 + cout + tiny)
    b, h, w_ = 3, 37, 29   # M = 3219: not a multiple of any tile
    x = bf(torch.randn((b, cin, h, w_), generator=g))
    w = (torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5).requires_grad_(True)
    bias = torch.randn(cout, generator=g) * 0.1
    code = 700 + (cin // 32) * 10 + (0 if cout > 128 else 1 if cout > 64 else 2)
    xr = x.clone().requires_grad_(True)
    yr = TF.conv2d(xr, bf(w), None)
    gy = bf(torch.randn(yr.shape, generator=g))
    yr.backward(gy)
    xd = nhwc(x).requires_grad_(True)
    wd = w.detach().to(DEV).requires_grad_(True)
    with forced(conv=STREAM | (tiny << 16)):
        y, stats = F._Conv2dFn.apply(xd, wd, None, 1, 0, False, True)
        assert last_kernel() == code, (last_kernel(), code)
        yb = F.conv2d(nhwc(x), w.detach().to(DEV), bias.to(DEV), 1, 0, relu=True)
        assert last_kernel() == code
        y.backward(nhwc(gy))
    yy = nchw(y, cout)
    assert rel_err(yy, yr.detach()) < ULP
    assert float(y[..., cout:].abs().max()) == 0.0
    assert torch.allclose(stats[0].cpu(), yy.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(stats[1].cpu(), (yy * yy).sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    assert rel_err(nchw(yb, cout), TF.relu(TF.conv2d(x, bf(w.detach()), bf(bias)))) < ULP
    assert rel_err(nchw(xd.grad, cin), xr.grad) < ULP


@pytest.mark.parametrize("variant,code", [(NEVER_TILE, igemm_code(64, 256, 64, 2)), (NEVER_TILE | 1, igemm_code(64, 256, 64, 2, 0)),
                                          (NEVER_TILE | 32, igemm_code(64, 256, 64, 3)), (NEVER_TILE | 48, igemm_code(64, 256, 64, 4)),
                                          (NEVER_TILE | 4, igemm_code(32, 256, 64, 4)), (NEVER_TILE | 4 | 16, igemm_code(32, 256, 64, 2)),
                                          (NEVER_TILE | 4 | 32, igemm_code(32, 256, 64, 3)), (NEVER_TILE | 4 | 1, igemm_code(32, 256, 64, 2, 0))])
def test_conv_narrow_tile_variants(F, variant, code):
    """The 256 x 64 tile family (layers with <= 64 output channels): 1x1 512 -> 40 with bias and ReLU."""
    g = torch.Generator().manual_seed(5 + variant % 97)
    x = bf(torch.randn((2, 512, 19, 23), generator=g))
    w = torch.randn((40, 512, 1, 1), generator=g) / 512 ** 0.5
    b = torch.randn(40, generator=g) * 0.1
    yr = TF.relu(TF.conv2d(x, bf(w), bf(b)))  # autocast rounds the bias like the other operands
    with forced(conv=variant):
        y = F.conv2d(nhwc(x), w.to(DEV), b.to(DEV), 1, 0, relu=True)
        assert last_kernel() == code
    assert rel_err(nchw(y, 40), yr) < ULP
    assert float(y[..., 40:].abs().max()) == 0.0


@pytest.mark.parametrize("variant,code", [(0, 2003), (1, 2001), (2, 2002), (3, 2000), (64, 2103), (128, 2003), (256, 2256),
                                          (256 | 64, 2356), (16, 2003), (32, 2003),
                                          # bit 17: partial tiles + wgrad_reduce_kernel wherever there are >= 2 pixel splits, bit 16: atomics
                                          (131072, 2003), (131072 | 256, 2256), (131072 | 64, 2103), (65536, 2003)])
def test_wgrad_forced_variants(F, variant, code):
    """conv_wgrad_kernel<GLDS, TR> (all four), the XCD-grouped launch and conv_wgrad256_kernel on 3x3 s1 128 -> 136 (the plain
    layout) and on 1x1 (the direct-into-arena layout), vs fp32."""
    g = torch.Generator().manual_seed(3 + variant)
    for (cin, cout, k, pad) in ((128, 136, 3, 1), (256, 264, 1, 0)):
        x = bf(torch.randn((2, cin, 21, 27), generator=g))
        w = (torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5).requires_grad_(True)
        yr = TF.conv2d(x, bf(w), None, 1, pad)
        gy = bf(torch.randn(yr.shape, generator=g))
        yr.backward(gy)
        wd = w.detach().to(DEV).requires_grad_(True)
        with forced(wgrad=variant):
            y, _ = F._Conv2dFn.apply(nhwc(x), wd, None, 1, pad, False, False)
            y.backward(nhwc(gy))
            assert last_kernel() == code, (variant, last_kernel(), code)
        assert rel_err(wd.grad.cpu(), w.grad) < WG_TOL


WS_FORCE, WS_TINY = 1 << 18, 1 << 20


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("tiny", [0, 1])
def test_wgrad_stream_kernel(F, cfg, tiny):
    """wgrad_stream_kernel (1x1 / stride 1: a persistent work-group per pixel range holds a whole block of dW in registers),
    every block configuration forced (U2_WGRAD_VARIANT bit 18, bits 21-23) with 8 pixel ranges and with the full grid, on maps
    whose pixel count is not a multiple of the 32-pixel step, with channel tails on both operands (40, 104, 136, 264, 520: blocks
    tiled over N and over C, half-empty blocks, 8-channel remainders) and a 5-pixel map (ranges without a step), vs fp32.  The 1x1
    weight gradient goes through u2_conv_wgrad_into (the arena layout [N][Cin]) whenever the weight owns a gradient slot."""
    g = torch.Generator().manual_seed(11 + cfg * 2 + tiny)
    for (b, cin, cout, h, w_) in ((1, 64, 256, 37, 29), (2, 256, 40, 33, 21), (1, 128, 136, 45, 23), (1, 264, 520, 19, 45),
                                 (1, 160, 64, 61, 67), (2, 512, 104, 40, 35), (1, 64, 64, 1, 5)):
        x = bf(torch.randn((b, cin, h, w_), generator=g))
        w = (torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5).requires_grad_(True)
        yr = TF.conv2d(x, bf(w), None, 1, 0)
        gy = bf(torch.randn(yr.shape, generator=g))
        yr.backward(gy)
        wd = w.detach().to(DEV).requires_grad_(True)
        with forced(wgrad=WS_FORCE | (WS_TINY if tiny else 0) | (cfg << 21)):
            y, _ = F._Conv2dFn.apply(nhwc(x), wd, None, 1, 0, False, False)
            y.backward(nhwc(gy))
            assert last_kernel() // 100 == 27, (cfg, last_kernel())
            if cfg:
                assert last_kernel() == 2700 + cfg
        assert rel_err(wd.grad.cpu(), w.grad) < WG_TOL, (cfg, tiny, cin, cout)


@pytest.mark.parametrize("name,b,h,w,cin,cout,code", [
    ("res2 conv3 1x1 64->256 @200x336", 16, 200, 336, 64, 256, 2701),
    ("res2 conv1 1x1 256->64 @200x336", 16, 200, 336, 256, 64, 2702),
    ("res3 conv3 1x1 128->512 @100x168", 16, 100, 168, 128, 512, 2703),
    ("fpn lateral2 1x1 256->256 @200x336", 16, 200, 336, 256, 256, 2704),
    ("semantic predictor 1x1 128->28 @200x336", 16, 200, 336, 128, 28, 2702),
    ("res4 conv3 1x1 256->1024 @50x84 (stays on the tile kernels)", 16, 50, 84, 256, 1024, 2256),
])
def test_wgrad_stream_full_shapes_auto_dispatch(F, name, b, h, w, cin, cout, code):
    """Benchmark-shape 1x1 weight gradients through the automatic dispatch: the streaming kernel takes the stride-4 / stride-8
    maps with the expected block configuration (the stride-16 layer stays where it was), and the result agrees with an fp32
    reference formed on the GPU by torch (dW = dY^T X over all 268 800 / 1 075 200 pixels; independent of the HIP kernels)."""
    g = torch.Generator().manual_seed(len(name))
    x = torch.randn((b, h, w, cin), generator=g).bfloat16().to(DEV)
    cp = (cout + 31) // 32 * 32
    gy = torch.zeros((b, h, w, cp), dtype=torch.bfloat16, device=DEV)
    gy[..., :cout] = torch.randn((b, h, w, cout), generator=g).bfloat16().to(DEV)
    wd = (torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5).to(DEV).requires_grad_(True)
    with forced():
        y, _ = F._Conv2dFn.apply(x, wd, None, 1, 0, False, False)
        y.backward(gy)
        assert last_kernel() == code, (name, last_kernel())
    ref = gy[..., :cout].reshape(-1, cout).float().t() @ x.reshape(-1, cin).float()
    assert rel_err(wd.grad.reshape(cout, cin), ref) < 2e-3, name   # fp32 sums of the same bf16 products on both sides


@pytest.mark.parametrize("tiny", [0, 1])
def test_fused_1x1_backward(F, tiny):
    """u2_conv1x1_bwd_fused (wgrad_stream_kernel<.., DG>): the data gradient and the weight gradient of a 1x1 / stride-1 conv in
    one pass over the output gradient - reached from _Conv2dFn.backward when the weight owns an arena slot (solver.FlatSGD) -
    forced on small maps (U2_WDGRAD_VARIANT bit 0; the automatic rule takes >= 200 000 pixels), both block shapes (all of N <= 256
    by 64 input channels; N <= 512 by 64 in two c-tiles), pixel counts that are no multiple of the 32-pixel step, channel tails
    on both sides, vs fp32: dx at one bf16 step, dW like the other weight-gradient kernels; the padded input channels of dx are
    written as zeros.  The same module without the fused launch (bit 2: never) gives the same gradients to fp32 summation order."""
    from u2seg_amd.layers.modules import Conv2d
    from u2seg_amd.solver import FlatSGD

    old_min = F.FUSED_BWD_MIN_PIXELS
    g = torch.Generator().manual_seed(40 + tiny)
    try:
        F.FUSED_BWD_MIN_PIXELS = 0
        for (b, cin, cout, h, w_, code) in ((1, 64, 256, 37, 29, 2751), (2, 40, 200, 33, 21, 2751), (1, 128, 512, 19, 45, 2752),
                                           (1, 72, 264, 45, 23, 2752)):
            x = bf(torch.randn((b, cin, h, w_), generator=g))
            w = torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5
            xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            yr = TF.conv2d(xr, bf(wr), None, 1, 0)
            gy = bf(torch.randn(yr.shape, generator=g))
            yr.backward(gy)
            got = {}
            for mode, variant in (("fused", 1 | (2 if tiny else 0)), ("separate", 4)):
                conv = Conv2d(cin, cout, 1, bias=False).to(DEV)
                with torch.no_grad():
                    conv.weight.copy_(w.to(DEV))
                opt = FlatSGD(conv, lr=0.1)
                opt.zero_grad()
                xd = nhwc(x).requires_grad_(True)
                old = os.environ.get("U2_WDGRAD_VARIANT")
                os.environ["U2_WDGRAD_VARIANT"] = str(variant)
                try:
                    y = conv(xd)
                    y.backward(nhwc(gy))
                    if mode == "fused":
                        assert last_kernel() == code, (cin, cout, last_kernel())
                finally:
                    os.environ.pop("U2_WDGRAD_VARIANT") if old is None else os.environ.__setitem__("U2_WDGRAD_VARIANT", old)
                F.join_all_streams()
                got[mode] = (nchw(xd.grad, cin), conv.weight.grad.detach().float().cpu().clone(), xd.grad[..., cin:].float().abs().max() if xd.grad.shape[3] > cin else torch.zeros(()))
            for mode, (dx, dw, padmax) in got.items():
                assert rel_err(dx, xr.grad) < ULP, (mode, cin, cout)
                assert rel_err(dw, wr.grad) < WG_TOL, (mode, cin, cout)
                assert float(padmax) == 0.0, (mode, cin, cout)
            assert rel_err(got["fused"][1], got["separate"][1]) < 1e-5     # fp32 sums of the same products
    finally:
        F.FUSED_BWD_MIN_PIXELS = old_min


def test_bias_gradient_kernels(F):
    """u2_colsum_add (bias gradient accumulated into its arena slice) and u2_relu_bwd_colsum (ReLU backward + that sum in one pass)
    through the C ABI: dz bit-equal to u2_relu_bwd, the sums against fp32 column sums of the same bf16 values (1e-5 relative: fp32
    atomics over a few hundred partial sums), accumulation on top of what the slice already holds, channels >= n_valid untouched;
    row counts that are no multiple of the block, channel counts with and without padding."""
    from u2seg_amd import _hip

    g = torch.Generator().manual_seed(12)
    for rows, cp, n in ((1000, 32, 15), (16 * 37 * 29, 256, 256), (4111, 64, 40), (7, 32, 32)):
        dout = torch.randn((rows, cp), generator=g).bfloat16().to(DEV)
        out = torch.randn((rows, cp), generator=g).bfloat16().to(DEV)
        base = torch.randn(cp, generator=g).to(DEV)
        # column sums only
        dst = base.clone()
        _hip.call("u2_colsum_add", dout, dst, rows, cp, cp, n)
        want = base.clone()
        want[:n] += dout.float().sum(0)[:n]
        assert rel_err(dst[:n], want[:n]) < 1e-5 and torch.equal(dst[n:], base[n:])
        # ReLU backward + column sums
        dz_ref = torch.empty_like(dout)
        _hip.call("u2_relu_bwd", dout, out, dz_ref, dout.numel())
        dz = torch.empty_like(dout)
        dst = base.clone()
        _hip.call("u2_relu_bwd_colsum", dout, out, dz, dst, torch.zeros(cp, device=DEV), rows, cp, cp, n)
        assert torch.equal(dz, dz_ref)
        assert torch.equal(dz_ref.float(), dout.float() * (out.float() > 0))
        want = base.clone()
        want[:n] += dz_ref.float().sum(0)[:n]
        assert rel_err(dst[:n], want[:n]) < 1e-5 and torch.equal(dst[n:], base[n:])


@pytest.mark.parametrize("tail", [True, False])
def test_fused_1x1_backward_with_batch_norm_apply(F, tail):
    """u2_conv1x1_bwd_fused_bn (wgrad_stream_kernel<8, 1, 2, 4, DG, AP>, code 2761): conv -> batch norm (-> + residual -> ReLU), whose
    backward apply step dx = k1 dz + k2 y + k3 is evaluated by the conv's fused backward launch on the staged rows - autograd carries a
    placeholder from _BatchNormActFn.backward to _Conv2dFn.backward.  Against the same module with the step as its own launch
    (F.LAZY_BN_APPLY False, the fused launch 2751 on the stored dy): the input gradient to the few bf16 roundings of dy that the
    atomically summed coefficients flip between two runs (bit for bit on fixed coefficients: tests/native/selftest wdgrad_bn), dW and
    dgamma / dbeta to fp32 summation order, the residual's gradient identical.  Third leg: the deferral is taken but the fused launcher declines (variant bit 2) -
    the conv materialises dy with u2_norm_bwd_apply and takes the separate launches; same gradients."""
    from u2seg_amd.layers.modules import BatchNorm2d, Conv2d
    from u2seg_amd.solver import FlatSGD

    old_min, old_lazy = F.FUSED_BWD_MIN_PIXELS, F.LAZY_BN_APPLY
    old_env = os.environ.get("U2_WDGRAD_VARIANT")
    g = torch.Generator().manual_seed(91 + int(tail))
    try:
        F.FUSED_BWD_MIN_PIXELS = 0
        F.set_deterministic_stats(True)   # fixed-order batch statistics: the three legs normalise with the same mean / invstd
        for (b, cin, cout, h, w_) in ((1, 64, 256, 37, 29), (2, 40, 192, 33, 21), (3, 64, 256, 64, 50)):   # norm layers: C % 32 == 0
            x = bf(torch.randn((b, cin, h, w_), generator=g))
            w = torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5
            gam, bet = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
            res = bf(torch.randn((b, cout, h, w_), generator=g))
            gy = bf(torch.randn((b, cout, h, w_), generator=g))
            got = {}
            for mode in ("in the conv", "own launch", "declined"):
                F.LAZY_BN_APPLY = mode != "own launch"
                conv = Conv2d(cin, cout, 1, bias=False, norm=BatchNorm2d(cout, sync=False)).to(DEV)
                conv.train()
                with torch.no_grad():
                    conv.weight.copy_(w.to(DEV))
                    conv.norm.weight.copy_(gam.to(DEV))
                    conv.norm.bias.copy_(bet.to(DEV))
                opt = FlatSGD(conv, lr=0.1)
                opt.zero_grad()
                xd = nhwc(x).requires_grad_(True)
                rd = nhwc(res).requires_grad_(True) if tail else None
                os.environ["U2_WDGRAD_VARIANT"] = "1"
                out = conv(xd, residual=rd, relu=tail)
                if mode == "declined":
                    os.environ["U2_WDGRAD_VARIANT"] = "4"
                out.backward(nhwc(gy))
                if mode == "in the conv":
                    assert last_kernel() == 2761, (cin, cout, last_kernel())
                F.assert_no_deferred_gradients()
                F.join_all_streams()
                got[mode] = (xd.grad.clone(), conv.weight.grad.detach().float().clone(), conv.norm.weight.grad.detach().clone(),
                             conv.norm.bias.grad.detach().clone(), rd.grad.clone() if tail else None)
            ref = got["own launch"]
            for mode in ("in the conv", "declined"):
                dx, dw, dg, db, dr = got[mode]
                # (the backward column sums are fp32 atomics: the coefficients differ in their last bits between two runs, which
                # flips a few bf16 roundings of dy; tests/native/selftest wdgrad_bn holds the bit-for-bit comparison on fixed k)
                assert rel_err(dx.float(), ref[0].float()) < 2 * ULP, (mode, cin, cout)   # one bf16 step of single elements
                assert float((dx != ref[0]).float().mean()) < (1.0 if mode == "declined" else 0.02), (mode, cin, cout)
                assert rel_err(dw, ref[1]) < 1e-4, (mode, cin, cout)
                assert rel_err(dg, ref[2]) < 1e-5 and rel_err(db, ref[3]) < 1e-5, (mode, cin, cout)
                if tail:
                    assert torch.equal(dr, ref[4]), (mode, cin, cout)
            assert float(ref[0].float().abs().max()) > 0
    finally:
        F.FUSED_BWD_MIN_PIXELS, F.LAZY_BN_APPLY = old_min, old_lazy
        F.set_deterministic_stats(False)
        os.environ.pop("U2_WDGRAD_VARIANT") if old_env is None else os.environ.__setitem__("U2_WDGRAD_VARIANT", old_env)


def test_deferred_batch_norm_gradient_guards(F):
    """the deferral hands autograd a placeholder instead of the conv output's gradient.  (i) the placeholder is
    zero-filled and comes from a fixed pool; (ii) a conv output with a SECOND consumer makes autograd sum the placeholder with
    that consumer's gradient - the tensor hook on the conv output raises inside that backward pass instead of letting the
    convolution see a sum without the normalisation's part; (iii) entries a failed backward left behind are dropped (and
    reported) by the next FlatSGD.zero_grad, and a clean pass afterwards still takes the fused launch."""
    from u2seg_amd.layers.modules import BatchNorm2d, Conv2d
    from u2seg_amd.solver import FlatSGD

    old_min, old_env = F.FUSED_BWD_MIN_PIXELS, os.environ.get("U2_WDGRAD_VARIANT")
    g = torch.Generator().manual_seed(5)
    try:
        F.FUSED_BWD_MIN_PIXELS = 0
        os.environ["U2_WDGRAD_VARIANT"] = "1"
        conv = Conv2d(64, 256, 1, bias=False, norm=BatchNorm2d(256, sync=False)).to(DEV)
        conv.train()
        opt = FlatSGD(conv, lr=0.1)
        x = nhwc(bf(torch.randn((2, 64, 33, 21), generator=g)))
        ph = F._lazy_placeholder(x.device)
        assert float(ph.float().abs().sum()) == 0.0 and ph.numel() == 1
        # (ii) second consumer of the conv output
        opt.zero_grad()
        xd = x.clone().requires_grad_(True)
        y, stats = F.conv2d(xd, conv.weight, None, want_stats=True)
        assert getattr(y, "_u2_lazy_ok", False)
        n = conv.norm
        out = F.batch_norm_act(y, stats, n.weight, n.bias, n.running_mean, n.running_var, sync=False)
        with pytest.raises(RuntimeError, match="second consumer"):
            (out.float().sum() + y.float().sum()).backward()
        assert not F._LAZY_GRADS
        # (iii) a stale entry is reported by the next zero_grad, then the path works again
        F._LAZY_GRADS[(x.device.index, ph.data_ptr())] = (ph, None, None, None)
        with pytest.raises(RuntimeError, match="never consumed"):
            opt.zero_grad()
        opt.zero_grad()
        xd = x.clone().requires_grad_(True)
        conv(xd).float().sum().backward()
        assert last_kernel() == 2761
        F.assert_no_deferred_gradients()
        F.join_all_streams()
        assert float(xd.grad.float().abs().max()) >= 0 and torch.isfinite(conv.weight.grad).all()
    finally:
        F.FUSED_BWD_MIN_PIXELS = old_min
        F._LAZY_GRADS.clear()
        os.environ.pop("U2_WDGRAD_VARIANT") if old_env is None else os.environ.__setitem__("U2_WDGRAD_VARIANT", old_env)


def test_fused_1x1_backward_full_shape_auto(F):
    """res2 conv3 (1x1 64 -> 256 over 16 x 200 x 336 pixels) through the automatic dispatch: the fused launch (code 2751) takes it;
    dW against an fp32 reference formed by torch on the GPU over all 1 075 200 pixels, dx at 4096 sampled pixels."""
    from u2seg_amd.layers.modules import Conv2d
    from u2seg_amd.solver import FlatSGD

    g = torch.Generator().manual_seed(77)
    b, h, w_, cin, cout = 16, 200, 336, 64, 256
    x = torch.randn((b, h, w_, cin), generator=g).bfloat16().to(DEV).requires_grad_(True)
    gy = torch.randn((b, h, w_, cout), generator=g).bfloat16().to(DEV)
    conv = Conv2d(cin, cout, 1, bias=False).to(DEV)
    opt = FlatSGD(conv, lr=0.1)
    opt.zero_grad()
    with forced():
        y = conv(x)
        y.backward(gy)
        assert last_kernel() == 2751, last_kernel()
    F.join_all_streams()
    wq = conv.weight.detach().reshape(cout, cin).bfloat16().float()
    ref_dw = gy.reshape(-1, cout).float().t() @ x.detach().reshape(-1, cin).float()
    assert rel_err(conv.weight.grad.reshape(cout, cin), ref_dw) < 2e-3
    idx = torch.randint(0, b * h * w_, (4096,), generator=g).to(DEV)
    ref_dx = gy.reshape(-1, cout)[idx].float() @ wq
    assert rel_err(x.grad.reshape(-1, cin)[idx].float(), ref_dx) < ULP


@pytest.mark.parametrize("variant", [4096, 4096 | (1 << 14), 4096 | (1 << 17), 4096 | (1 << 17) | (1 << 14), 4096 | (1 << 16)])
@pytest.mark.parametrize("shape", [(2, 64, 64, 14, 14), (1, 128, 136, 13, 17), (3, 96, 200, 25, 42), (2, 64, 256, 40, 70),
                                   (2, 32, 8, 3, 11), (1, 32, 72, 5, 101)])
def test_wgrad_halo_kernel(F, variant, shape):
    """conv_wgrad_halo_kernel (3x3 / stride 1 / pad 1, all nine taps per work-group over padded pixel coordinates) forced on
    small maps - every row wrap count (W + 1 < 16, < 32, >= 32), partial channel tiles, one to many steps per work-group -
    vs fp32, with the atomic epilogue (bit 16) and with partial blocks + wgrad_halo_reduce_kernel (bit 17); the automatic
    dispatch takes it from ~400 positions per work-group (test_wgrad_halo_mid_size_shapes_auto_dispatch)."""
    b, cin, cout, h, w_ = shape
    g = torch.Generator().manual_seed(h * 100 + w_ + variant)
    x = bf(torch.randn((b, cin, h, w_), generator=g))
    w = (torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5).requires_grad_(True)
    yr = TF.conv2d(x, bf(w), None, 1, 1)
    gy = bf(torch.randn(yr.shape, generator=g))
    yr.backward(gy)
    wd = w.detach().to(DEV).requires_grad_(True)
    with forced(wgrad=variant):
        y, _ = F._Conv2dFn.apply(nhwc(x), wd, None, 1, 1, False, False)
        y.backward(nhwc(gy))
        assert last_kernel() == 2900, last_kernel()
    assert rel_err(wd.grad.cpu(), w.grad) < WG_TOL


@pytest.mark.parametrize("name,b,h,w,cin,cout", [
    ("fpn_output4 / res4 conv2 3x3 256->256 @50x84", 16, 50, 84, 256, 256),
    ("res3 conv2 3x3 128->128 @100x168", 16, 100, 168, 128, 128),
    ("res5 conv2 3x3 512->512 @25x42", 16, 25, 42, 512, 512),
    ("mask head 3x3 256->256 @260x14x14", 260, 14, 14, 256, 256),
    ("fpn_output5 3x3 256->256 @25x42", 16, 25, 42, 256, 256),
    ("res2 conv2 3x3 64->64 @200x336", 16, 200, 336, 64, 64),
])
def test_wgrad_halo_mid_size_shapes_auto_dispatch(F, name, b, h, w, cin, cout):
    """Round 6: the mid-size 3x3 weight gradients of the benchmark step through the automatic dispatch - the nine-tap halo
    kernel with the partial-block epilogue (wgrad_halo.hip: 256 work-groups x 288 KB of plain stores + one reduction pass
    instead of as many atomics) - on a sampled 16 x 16 channel block of all nine taps vs fp32, twice into the same gradient
    (the pass adds into dw, it does not overwrite it)."""
    g = torch.Generator().manual_seed(len(name))
    x = torch.randn((b, h, w, cin), generator=g).bfloat16().to(DEV)
    wt = torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5
    wd = wt.to(DEV).requires_grad_(True)
    gy = torch.randn((b, h, w, cout), generator=g).bfloat16().to(DEV)
    with forced():
        for _ in range(2):
            y, _ = F._Conv2dFn.apply(x, wd, None, 1, 1, False, False)
            y.backward(gy)
            assert last_kernel() == 2900, (name, last_kernel())
    n0, c0 = cout - 16, cin // 2
    xs = x[..., c0:c0 + 16].float().permute(0, 3, 1, 2)
    gs = gy[..., n0:n0 + 16].float().permute(0, 3, 1, 2)
    # dW[n][c][kh][kw] = sum_m gy[m][n] * x[m + (kh - 1, kw - 1)][c]
    xp = TF.pad(xs, (1, 1, 1, 1))
    ref = torch.stack([torch.stack([torch.einsum("bnhw,bchw->nc", gs, xp[:, :, kh:kh + h, kw:kw + w]) for kw in range(3)], -1)
                       for kh in range(3)], -2)
    got = wd.grad[n0:n0 + 16, c0:c0 + 16].float()
    assert rel_err(got.cpu(), 2 * ref.cpu()) < WG_TOL, name


def _sampled_conv_ref(x, w, bias, pos, pad):
    """fp32 conv outputs at sampled positions.  x [B,H,W,C] bf16 (GPU), w [N,C,KH,KW] fp32 (GPU, bf16-rounded by the caller),
    pos int64 [S,3] (b, y, x) -> [S, N] fp32."""
    n, c, kh, kw = w.shape
    b, h, wd_, _ = x.shape
    out = torch.zeros((pos.shape[0], n), dtype=torch.float32, device=x.device)
    for i in range(kh):
        for j in range(kw):
            yy, xx = pos[:, 1] + i - pad, pos[:, 2] + j - pad
            ok = (yy >= 0) & (yy < h) & (xx >= 0) & (xx < wd_)
            patch = x[pos[:, 0], yy.clamp(0, h - 1), xx.clamp(0, wd_ - 1)][:, :c].float() * ok[:, None]
            out += patch @ w[:, :, i, j].t()
    if bias is not None:
        out += bias
    return out


FULL_SHAPES = [
    # name, B, H, W, cin, cout, k, pad, relu+bias, expected forward kernel under the automatic dispatch
    ("fpn_output2 3x3 256->256 @200x336", 16, 200, 336, 256, 256, 3, 1, False, 300),   # conv_halo.hip
    ("rpn conv 3x3 256->256 @100x168 bias relu", 16, 100, 168, 256, 256, 3, 1, True, 501),   # stream-K form of configuration 1
    ("res3 conv2 3x3 128->128 @100x168", 16, 100, 168, 128, 128, 3, 1, False, 300),
    ("res2 conv2 3x3 64->64 @200x336", 16, 200, 336, 64, 64, 3, 1, False, 301),   # 64-channel tiles
    ("res4 conv3 1x1 256->1024 @50x84", 16, 50, 84, 256, 1024, 1, 0, False, 780),            # conv_stream.hip
    ("res2 conv3 1x1 64->256 @200x336", 16, 200, 336, 64, 256, 1, 0, False, 720),
    ("res2 conv1 1x1 256->64 @200x336", 16, 200, 336, 256, 64, 1, 0, False, 782),
    ("res3 conv3 1x1 128->512 @100x168", 16, 100, 168, 128, 512, 1, 0, False, 740),
    ("res5 conv1 1x1 2048->512 @25x42", 16, 25, 42, 2048, 512, 1, 0, False, 502),            # 132 tiles: stream-K
    ("fpn_output4 3x3 256->256 @50x84", 16, 50, 84, 256, 256, 3, 1, False, 501),            # 263 tiles: stream-K
    ("mask head 3x3 256->256 @261x14x14 bias relu", 261, 14, 14, 256, 256, 3, 1, True, 101),
]


@pytest.mark.parametrize("name,b,h,w,cin,cout,k,pad,br,code", FULL_SHAPES)
def test_conv_full_shapes_auto_dispatch_forward(F, name, b, h, w, cin, cout, k, pad, br, code):
    """Benchmark-shape layers through the automatic dispatch: the expected persistent-tile configuration is selected and
    4096 sampled output pixels (all channels) agree with fp32; the BN statistics equal the column sums of the stored output."""
    g = torch.Generator().manual_seed(len(name))
    x = (torch.randn((b, h, w, cin), generator=g)).bfloat16().to(DEV)
    wt = torch.randn((cout, cin, k, k), generator=g) / (cin * k * k) ** 0.5
    bias = (torch.randn(cout, generator=g) * 0.1) if br else None
    with forced():
        y, stats = F._Conv2dFn.apply(x, wt.to(DEV), bias.to(DEV) if br else None, 1, pad, br, not br)
        assert last_kernel() == code, (name, last_kernel())
    pos = torch.stack([torch.randint(0, b, (4096,), generator=g), torch.randint(0, h, (4096,), generator=g),
                       torch.randint(0, w, (4096,), generator=g)], dim=1).to(DEV)
    pos[:64, 1] = 0
    pos[64:128, 2] = w - 1
    pos[128:160] = torch.tensor([b - 1, h - 1, w - 1], device=DEV)  # the last pixel of the last tile
    ref = _sampled_conv_ref(x, bf(wt).to(DEV), bias.to(DEV) if br else None, pos, pad)
    if br:
        ref = ref.relu()
    got = y[pos[:, 0], pos[:, 1], pos[:, 2]][:, :cout].float()
    assert float((got - ref).abs().max() / ref.abs().max()) < ULP, name
    if stats is not None:
        yf = y[..., :cout].float().reshape(-1, cout)
        assert torch.allclose(stats[0], yf.sum(0), rtol=2e-4, atol=0.5)
        assert torch.allclose(stats[1], (yf * yf).sum(0), rtol=2e-4, atol=0.5)


def test_conv_fpn_output2_backward_full_shape(F):
    """The fpn_output2 layer at the benchmark shape (16 x 200 x 336, 3x3 256 -> 256), backward through the automatic
    dispatch: data gradient at sampled pixels and weight gradient on a sampled 16 x 16 channel block (all 9 taps) vs fp32."""
    g = torch.Generator().manual_seed(77)
    b, h, w, c = 16, 200, 336, 256
    x = torch.randn((b, h, w, c), generator=g).bfloat16().to(DEV).requires_grad_(True)
    wt = (torch.randn((c, c, 3, 3), generator=g) / (c * 9) ** 0.5)
    wd = wt.to(DEV).requires_grad_(True)
    gy = torch.randn((b, h, w, c), generator=g).bfloat16().to(DEV)
    with forced():
        y, _ = F._Conv2dFn.apply(x, wd, None, 1, 1, False, True)
        y.backward(gy)
        assert last_kernel() == 2900, last_kernel()  # the wgrad ran last: nine-tap halo kernel (wgrad_halo.hip)
    # data gradient = conv of gy with the flipped, transposed filter
    wflip = bf(wt).flip(2, 3).permute(1, 0, 2, 3).contiguous().to(DEV)
    pos = torch.stack([torch.randint(0, b, (2048,), generator=g), torch.randint(0, h, (2048,), generator=g),
                       torch.randint(0, w, (2048,), generator=g)], dim=1).to(DEV)
    pos[:32, 1] = 0
    pos[32:64, 2] = w - 1
    ref = _sampled_conv_ref(gy, wflip, None, pos, 1)
    got = x.grad[pos[:, 0], pos[:, 1], pos[:, 2]].float()
    assert float((got - ref).abs().max() / ref.abs().max()) < ULP
    # weight gradient block: n in ns, c in cs, all taps
    ns = torch.tensor([0, 1, 17, 63, 64, 100, 127, 128, 129, 190, 200, 222, 240, 250, 254, 255], device=DEV)
    cs = torch.tensor([0, 3, 31, 32, 33, 64, 65, 90, 127, 128, 160, 191, 192, 200, 254, 255], device=DEV)
    gys = gy[..., ns].float()
    xs = TF.pad(x.detach()[..., cs].float(), (0, 0, 1, 1, 1, 1))
    for kh in range(3):
        for kw in range(3):
            ref_w = torch.einsum("bhwn,bhwc->nc", gys, xs[:, kh : kh + h, kw : kw + w])
            got_w = wd.grad[ns][:, cs][:, :, kh, kw]
            assert float((got_w - ref_w).abs().max() / ref_w.abs().max()) < WG_TOL, (kh, kw)


@pytest.mark.parametrize("shape", [(4, 200, 169, 64, 256, 104), (2, 37, 41, 64, 256, None), (1, 90, 70, 512, 2048, 104),
                                   (3, 20, 24, 256, 1024, 104)])
def test_conv_accumulating_epilogue(F, shape):
    """conv_tile_kernel<4, 1, 3, ACC>: out = relu(bf16(conv1x1(x) + bias) + out) in place - the residual tail of a bottleneck at
    inference (backbone/resnet.py:204-210 with the fixed-statistics norm folded into the conv) - vs fp32, with partial tiles;
    a map too small for the persistent kernel at K = 64 goes to conv_igemm_kernel's accumulate path (same rounding rule)."""
    b, h, w, cin, cout, code = shape
    g = torch.Generator().manual_seed(cin + cout)
    x = bf(torch.randn((b, cin, h, w), generator=g))
    wt = torch.randn((cout, cin, 1, 1), generator=g) / cin ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    res = bf(torch.randn((b, cout, h, w), generator=g))
    ref = TF.relu(bf(TF.conv2d(x, bf(wt), bias)) + res)  # the folded norm shift is not a conv bias: fp32
    out = nhwc(res)
    with torch.no_grad(), forced():
        got = F.conv2d_add_(nhwc(x), wt.to(DEV), bias.to(DEV), out, 1, 0, relu=True)
        assert code is None or last_kernel() == code, last_kernel()
    assert got.data_ptr() == out.data_ptr()
    assert rel_err(got.permute(0, 3, 1, 2)[:, :cout].float().cpu(), ref) < ULP


def test_streamk_region_rule(F):
    """Where the product path lets a convolution take the stream-K form (layers/functional.py:streamk_region): inside the region
    (the bottom-up backbone) the library's fill rule decides - fpn_output4's shape takes configuration 501, forward and data
    gradient -, outside it the call carries variant bit 28 and runs whole tiles (101) while the branch streams are on, and the
    rule is off when they are off (set_stream_overlap(False)) or U2_STREAMK_EVERYWHERE=1; results agree either way."""
    g = torch.Generator().manual_seed(5)
    b, h, w, c = 16, 50, 84, 256
    x = torch.randn((b, h, w, c), generator=g).bfloat16().to(DEV)
    wt = (torch.randn((c, c, 3, 3), generator=g) / (c * 9) ** 0.5).to(DEV)
    old = {k: os.environ.pop(k, None) for k in ("U2_CONV_VARIANT", "U2_STREAMK_EVERYWHERE", "U2_AUX_STREAM")}
    try:
        with F.streamk_region():
            xin = x.clone().requires_grad_(True)
            y_in, _ = F._Conv2dFn.apply(xin, wt, None, 1, 1, False, False)
            assert last_kernel() == 501, last_kernel()
        y_out, _ = F._Conv2dFn.apply(x, wt, None, 1, 1, False, False)
        assert last_kernel() == 101, last_kernel()
        # the backward pass of a layer created inside the region keeps the permission outside of it
        y_in.backward(torch.ones_like(y_in))
        F.join_all_streams()
        assert xin.grad is not None
        F.set_stream_overlap(False)
        try:
            F._Conv2dFn.apply(x, wt, None, 1, 1, False, False)
            assert last_kernel() == 501, last_kernel()
        finally:
            F.set_stream_overlap(True)
        os.environ["U2_STREAMK_EVERYWHERE"] = "1"
        F._Conv2dFn.apply(x, wt, None, 1, 1, False, False)
        assert last_kernel() == 501, last_kernel()
        assert float((y_in.float() - y_out.float()).abs().max()) <= 2 ** -7 * float(y_out.float().abs().max())
    finally:
        os.environ.pop("U2_STREAMK_EVERYWHERE", None)
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v
