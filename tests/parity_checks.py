"""Small oracle checks reused by __graft_entry__.smoke() (one HIP launch each, compared with the CPU oracle)."""
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
