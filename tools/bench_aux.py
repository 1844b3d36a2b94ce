#!/usr/bin/env python
"""Secondary measurements (BASELINE.json configs 4 and 5): k-means and kNN lists over synthetic DINO-sized embeddings and
batch panoptic inference.  Prints one JSON line per measurement; numbers are quoted in DESIGN.md."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def bench_kmeans(n, d, k, iters):
    from u2seg_amd.cluster import kmeans as KM

    g = torch.Generator(device="cuda").manual_seed(0)
    centers = torch.randn((k, d), generator=g, device="cuda") * 2
    x = centers[torch.randint(0, k, (n,), generator=g, device="cuda")] + 0.5 * torch.randn((n, d), generator=g, device="cuda")
    torch.manual_seed(0)
    c = x[torch.randperm(n)[:k].cuda()].clone()
    for _ in range(2):
        lab = KM.assign(x, c)
        c2, _ = KM.update(x, lab, k)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_assign = t_update = 0.0
    t0 = time.time()
    for _ in range(iters):
        ev[0].record()
        lab = KM.assign(x, c)
        ev[1].record()
        c, _ = KM.update(x, lab, k)
        ev[2].record()
        torch.cuda.synchronize()
        t_assign += ev[0].elapsed_time(ev[1])
        t_update += ev[1].elapsed_time(ev[2])
    dt = (time.time() - t0) / iters
    flops = 2.0 * n * k * d
    print(json.dumps({"metric": "k-means s/iter", "N": n, "D": d, "K": k, "s_per_iter": dt, "assign_ms": t_assign / iters,
                      "update_ms": t_update / iters, "assign_TFLOPs": flops / (t_assign / iters * 1e-3) / 1e12,
                      "assign_frac_of_fp32_mfma_peak": flops / (t_assign / iters * 1e-3) / 157.3e12,
                      "update_GBps": n * d * 4 / (t_update / iters * 1e-3) / 1e9}))
    # CPU baseline on a bounded sample: one Lloyd iteration over 50k points with plain chunked torch on the host cores
    xs, cs = x[:50000].cpu(), c.cpu()
    t0 = time.time()
    lab = torch.cat([((xs[i:i + 4096, None, :] - cs[None]) ** 2).sum(-1).argmin(1) for i in range(0, xs.shape[0], 4096)])
    torch.zeros((k, d)).scatter_add_(0, lab[:, None].repeat(1, d), xs) / torch.bincount(lab, minlength=k)[:, None]
    dt = time.time() - t0
    print(json.dumps({"metric": "k-means cpu_baseline", "sample": "1 Lloyd iteration over 50k x %d, K=%d (chunked torch, %d threads)" %
                      (d, k, torch.get_num_threads()), "s_per_iter_scaled_to_N": dt * n / 50000}))


def bench_knn(n, d, k):
    """kNN lists of every row against all rows (partitioned_kNN): select kernel = 2 N^2 D flop on the fp32 MFMA."""
    from u2seg_amd.cluster import knn as KN

    g = torch.Generator(device="cuda").manual_seed(0)
    centers = torch.randn((300, d), generator=g, device="cuda")
    x = centers[torch.randint(0, 300, (n,), generator=g, device="cuda")] + 0.4 * torch.randn((n, d), generator=g, device="cuda")
    x = torch.nn.functional.normalize(x, dim=1)
    KN.partitioned_kNN(x[:4096], K=k)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    dk, ind = KN.partitioned_kNN(x, K=k)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1])
    flops = 2.0 * n * n * d
    print(json.dumps({"metric": "kNN lists s", "N": n, "D": d, "K": k, "s": ms * 1e-3, "TFLOPs": flops / (ms * 1e-3) / 1e12,
                      "frac_of_fp32_mfma_peak": flops / (ms * 1e-3) / 157.3e12, "self_first": bool((ind[:, 0] == torch.arange(n, device="cuda")).all())}))
    xs = x[:20000].cpu()
    t0 = time.time()
    for i in range(0, 2000, 128):  # difference-form distances + a sort per 128-row chunk (7.9 GB) on the host cores
        torch.sort(((xs[i:i + 128, None, :] - xs[None]) ** 2).sum(-1), dim=1)
    dt = time.time() - t0
    print(json.dumps({"metric": "kNN cpu_baseline", "sample": "2000 query rows x 20000 train rows x %d (chunked torch, %d threads)" %
                      (d, torch.get_num_threads()), "s_scaled_to_N_squared": dt * (n / 2000.0) * (n / 20000.0)}))


def bench_inference(batch, iters):
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import make_synthetic_batch
    from u2seg_amd.modeling import build_model

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "COCO-PanopticSegmentation", "u2seg_eval_800.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cuda:0"])
    torch.manual_seed(0)
    model = build_model(cfg)
    model.eval()
    data = make_synthetic_batch(batch, device="cuda:0")
    with torch.no_grad():
        out = model(data)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(iters):
            out = model(data)
        torch.cuda.synchronize()
    dt = (time.time() - t0) / iters
    print(json.dumps({"metric": "panoptic inference img/s (u2seg_eval_800, random init)", "batch": batch, "img_per_s": batch / dt,
                      "ms_per_batch": dt * 1e3, "instances_img0": len(out[0]["instances"]),
                      "segments_img0": len(out[0]["panoptic_seg"][1])}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["kmeans", "knn", "inference", "all"])
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--knn-n", type=int, default=200000)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32)
    a = ap.parse_args()
    if a.what in ("kmeans", "all"):
        bench_kmeans(a.n, 768, 300, a.iters)
    if a.what in ("knn", "all"):
        bench_knn(a.knn_n, 768, 20)
    if a.what in ("inference", "all"):
        bench_inference(a.batch, max(1, a.iters // 2))
