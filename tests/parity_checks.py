"""Small oracle checks reused by __graft_entry__.smoke() (one HIP launch each, compared with the CPU oracle)."""
import numpy as np
import torch


def smoke_oracle_check():
    from oracle import ops as O
    from u2seg_amd.layers import functional as F

    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    # ROIAlign forward (bf16 features) vs the C oracle
    feat = torch.randn((1, 32, 16, 20), generator=g).bfloat16().float()
    rois = torch.tensor([[0, 4.0, 4.0, 60.0, 50.0], [0, 10.5, 3.25, 33.0, 47.0]])
    ref = O.roi_align(feat, rois, 7, 0.25)
    fd = feat.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)
    out = F.roi_align([fd], rois.to(dev), torch.zeros(2, dtype=torch.int32, device=dev), 7, [0.25])
    err = float((out.permute(0, 3, 1, 2).float().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-2, err
    # NMS keep list: bit exact
    b = torch.rand((300, 2), generator=g) * 100
    b = torch.cat([b, b + 5 + torch.rand((300, 2), generator=g) * 60], 1)
    s = torch.rand(300, generator=g)
    order = torch.sort(s, descending=True, stable=True)[1]
    keep, nk = F.batched_nms(b[order][None].to(dev), torch.zeros((1, 300), dtype=torch.int32, device=dev),
                             torch.tensor([300], dtype=torch.int32, device=dev), 0.5, 300)
    got = order[keep[0, : int(nk[0])].long().cpu()]
    assert got.tolist() == O.nms(b, s, 0.5).tolist()
    print("smoke: oracle checks ok (roi_align rel err %.2e, nms keep list identical, %d kept)" % (err, len(got)))


def knn_lists_agree(x, d_got, ind_got, d_want, ind_want, rtol=1e-5, atol=1e-6):
    """The acceptance rule of the reference's own verify branch (nn_utils.py:268-287): distances equal; where the
    neighbour ids differ, the distance from the row to either id must be the same (tied or duplicated rows)."""
    np.testing.assert_allclose(d_got, d_want, rtol=rtol, atol=atol)
    rows, cols = np.nonzero(ind_got != ind_want)
    for r, c in zip(rows, cols):
        a = float(((x[r] - x[ind_got[r, c]]) ** 2).sum())
        b = float(((x[r] - x[ind_want[r, c]]) ** 2).sum())
        assert abs(a - b) <= atol + rtol * abs(b), (r, c, a, b)
    return len(rows)
