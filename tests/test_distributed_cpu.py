"""world_size-2 gloo tests of the data-parallel pieces that do not need a GPU: gradient exchange over the flat arena
(what bench.py / the trainer do after backward) and the per-rank data sharding rule."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from u2seg_amd.solver import FlatSGD

    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(40, 30), torch.nn.Linear(30, 5))
    opt = FlatSGD(model, lr=0.1, clip_value=1.0, bucket_bytes=1024)  # tiny buckets -> several all-reduces
    opt.zero_grad()
    x = torch.full((4, 40), float(rank + 1))
    model(x).sum().backward()
    local = opt.flat_grad.clone()
    scale = opt.all_reduce_grads()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ok = torch.allclose(opt.flat_grad, sum(gathered)) and scale == 1.0 / world
    # parameters are views into the arena and grads accumulate in place into it
    ok = ok and model[0].weight.grad.data_ptr() == opt.flat_grad.data_ptr()
    # overlapped exchange: the tail of the arena (the "head" parameters) starts its all-reduce early, the rest follows
    opt.zero_grad()
    model(x).sum().backward()
    local = opt.flat_grad.clone()
    tail = opt.offset_of(model[1].weight)
    ok = ok and tail == model[0].weight.numel() + model[0].bias.numel()
    opt.begin_all_reduce_tail(tail)
    opt.begin_all_reduce_tail(tail)  # idempotent within a step
    scale = opt.all_reduce_grads()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    ok = ok and torch.allclose(opt.flat_grad, sum(gathered)) and scale == 1.0 / world and opt._tail_from is None
    from u2seg_amd.data import make_synthetic_batch

    a = make_synthetic_batch(2, start_index=rank * 2, height=32, width=48)
    sig = torch.tensor([float(a[0]["image"].sum()), float(a[1]["image"].sum())])
    sigs = [torch.zeros(2) for _ in range(world)]
    dist.all_gather(sigs, sig)
    ok = ok and not torch.equal(sigs[0], sigs[1])  # ranks see different images
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_flat_gradient_allreduce_gloo_world2():
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
        assert out[0] and out[1]
