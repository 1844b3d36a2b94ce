#!/usr/bin/env python
"""Cost model of the per-layer SyncBN exchange on ONE GPU: two ranks share cuda:0 over gloo (RCCL refuses
two ranks on one device), and the three collectives a training step issues are timed in isolation, host clock around
launch + wait with the device drained in front:
  * forward: one all-reduce of [sum | sumsq | count] = 2C + 1 floats per normalisation layer (61 SyncBN layers: 53 of the ResNet-50, 8 of the FPN; C = 64 ... 2048);
  * backward: one all-reduce of [sum dz | sum dz xhat] = 2C floats per layer (dgamma / dbeta are formed from the LOCAL sums and ride
    the gradient buckets: nothing else is exchanged per layer);
  * the gradient buckets: 76.07 M fp32 in 64 MB buckets.
gloo stages CUDA tensors through host memory, so its latency is an UPPER bound for RCCL over xGMI; the table is the shape of the
cost (a fixed latency times 122 small collectives on the critical path), not a prediction of the 8-GPU number.
usage: python tools/syncbn_cost.py   (prints one JSON line)"""
import json
import os
import socket
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BN_CHANNELS = [64] + [64, 64, 256, 256] + [64, 64, 256] * 2 + [128, 128, 512, 512] + [128, 128, 512] * 3 + \
              [256, 256, 1024, 1024] + [256, 256, 1024] * 5 + [512, 512, 2048, 2048] + [512, 512, 2048] * 2 + [256] * 8   # + FPN laterals / outputs


def worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=2)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    res = {}
    for name, sizes in (("forward_2C+1", [2 * c + 1 for c in BN_CHANNELS]), ("backward_2C", [2 * c for c in BN_CHANNELS])):
        bufs = [torch.randn(n, device=dev) for n in sizes]
        for b in bufs:                      # warm-up: gloo pins / registers its staging buffers on first use
            dist.all_reduce(b)
        torch.cuda.synchronize()
        dist.barrier()
        per = []
        for rep in range(5):
            t0 = time.perf_counter()
            for b in bufs:
                dist.all_reduce(b)          # blocking, as the product issues them (the next kernel needs the statistics)
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t0) / len(bufs))
        res[name] = {"collectives": len(bufs), "us_each_median": sorted(per)[len(per) // 2] * 1e6}
    grad = torch.randn(76066554, device=dev)
    bucket = (64 << 20) // 4
    for rep in range(2):
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        hs = [dist.all_reduce(grad[s : s + bucket], async_op=True) for s in range(0, grad.numel(), bucket)]
        for h in hs:
            h.wait()
        torch.cuda.synchronize()
        res["gradient_buckets"] = {"buckets": len(hs), "ms_total": (time.perf_counter() - t0) * 1e3, "MB": grad.numel() * 4 / 1e6}
    if rank == 0:
        out["r"] = res
    dist.destroy_process_group()


def main():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(worker, args=(port, out), nprocs=2, join=True)
        r = dict(out["r"])
    n = len(BN_CHANNELS)
    r["per_step_on_the_critical_path_ms"] = (r["forward_2C+1"]["us_each_median"] + r["backward_2C"]["us_each_median"]) * n / 1e3
    r["note"] = "two ranks on one MI355X over gloo (host-staged): an upper bound for RCCL; %d BatchNorm layers" % n
    print(json.dumps(r))


if __name__ == "__main__":
    main()
