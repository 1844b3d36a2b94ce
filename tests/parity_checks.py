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


def write_fullwidth_reference_checkpoint(state_shapes, param_names, fx, path):
    """The full-width u2seg_R50_800 checkpoint of tests/golden/fullwidth_checkpoint_golden.json (make_fixtures.py --only fullwidth),
    rebuilt with torch alone: every tensor det_fill(name), one torch.optim.SGD step over the reference's recorded parameter groups on
    the gradients fullwidth_grad(name), saved in the nesting the reference's DefaultTrainer writes.  `state_shapes`: name -> shape
    of every state-dict entry, `param_names`: the parameter names.  Asserts that every tensor has the crc32 the REFERENCE's own
    objects produced, i.e. that the file is the one the reference would have written.  Returns (model state, momentum by name)."""
    import zlib

    from tests.golden.make_fixtures import det_fill, fullwidth_grad

    crc = lambda t: zlib.crc32(t.detach().contiguous().numpy().tobytes())
    assert sorted(state_shapes) == sorted(fx["model_crc32"]), "state-dict keys differ from the reference model's"
    state = {k: det_fill(k, torch.empty(shape)) for k, shape in state_shapes.items()}
    for k in state:
        if k.endswith("num_batches_tracked"):
            state[k] = torch.zeros(state_shapes[k], dtype=torch.long)
    params = {k: torch.nn.Parameter(state[k].clone()) for k in param_names}
    assert sorted(params) == sorted(fx["numbering"])
    groups, at = [], 0
    for g in fx["param_groups"]:
        names = fx["numbering"][at : at + g["n"]]
        at += g["n"]
        groups.append({"params": [params[k] for k in names], **{k: v for k, v in g.items() if k in ("lr", "weight_decay")}})
    g0 = fx["param_groups"][0]
    opt = torch.optim.SGD(groups, lr=fx["lr_of_the_step"], momentum=g0["momentum"], nesterov=g0["nesterov"], foreach=True)
    for g in opt.param_groups:
        g["lr"] = fx["lr_of_the_step"]
    for k, p in params.items():
        p.grad = fullwidth_grad(k, p.detach())
    opt.step()
    for k, p in params.items():
        state[k] = p.detach().clone()
    bad = [k for k, v in state.items() if crc(v) != fx["model_crc32"][k]]
    assert not bad, "rebuilt tensors differ from the reference's: %s" % bad[:5]
    osd = opt.state_dict()
    mom = {fx["numbering"][i]: st["momentum_buffer"] for i, st in osd["state"].items()}
    bad = [k for k, v in mom.items() if crc(v) != fx["momentum_crc32"][k]]
    assert not bad, "rebuilt momentum buffers differ from the reference's: %s" % bad[:5]
    # the groups as the reference's scheduler left them (lr of the NEXT step, initial_lr)
    for g, ref in zip(osd["param_groups"], fx["param_groups"]):
        for k, v in ref.items():
            if k != "n":
                g[k] = v
    it = 0
    torch.save({"model": state, "iteration": it,
                "trainer": {"iteration": it, "hooks": {"LRScheduler": dict(fx["scheduler"])},
                            "_trainer": {"iteration": it, "optimizer": osd}}}, path)
    return state, mom
