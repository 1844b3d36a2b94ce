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


def _real_arena_worker(rank, world, port, out):
    """The real u2seg_R50_800 model's arena (76 M parameters in the optimizer's own layout, 64 MB buckets): the exchange the
    trainer performs - the heads' tail of the arena started from the model's on_heads_backward_done hook while "backward" is
    still filling the backbone part, the rest in all_reduce_grads - must give, bit for bit, the result of one unbucketed
    all-reduce of the whole arena.  (The kernels need a GPU; the exchange, the bucket bounds, the hook and the tail offset do not.)"""
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from u2seg_amd.config import get_cfg
    from u2seg_amd.engine import SimpleTrainer
    from u2seg_amd.modeling import build_model
    from u2seg_amd.solver import build_optimizer

    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs",
                                     "COCO-PanopticSegmentation", "u2seg_R50_800.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", "cpu"])
    torch.manual_seed(0)
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    SimpleTrainer(model, opt)  # installs the hook that starts the tail exchange
    ok = opt.total == 76066554 and opt.bucket_elems == (64 << 20) // 4
    g = torch.Generator().manual_seed(100 + rank)
    grads = torch.randn(opt.total, generator=g) * torch.logspace(-6, 2, opt.total)  # eight decades of magnitudes
    # reference: one all-reduce over the whole arena
    ref = grads.clone()
    dist.all_reduce(ref)
    # the trainer's sequence: head gradients are final first -> hook -> backbone gradients arrive -> all_reduce_grads
    heads = [m for name, m in model.named_children() if name != "backbone"]
    tail = min(opt.offset_of(p) for m in heads for p in m.parameters() if p.requires_grad)
    ok = ok and 0 < tail < opt.total and max(opt.offset_of(p) for p in model.backbone.parameters() if p.requires_grad) < tail
    opt.zero_grad()
    opt.flat_grad[tail:].copy_(grads[tail:])
    model.on_heads_backward_done()
    ok = ok and opt._tail_from == tail and len(opt._pending) == -(-(opt.total - tail) // opt.bucket_elems)
    opt.flat_grad[:tail].copy_(grads[:tail])  # "backward" finishes the backbone while the tail is being summed
    scale = opt.all_reduce_grads()
    ok = ok and scale == 1.0 / world and opt._tail_from is None and not opt._pending
    ok = ok and torch.equal(opt.flat_grad, ref)
    # a parameter's gradient view sees the summed values (kernels and the optimizer read the arena through these views)
    w = model.backbone.bottom_up.stem.conv1.weight
    off = opt.offset_of(w)
    ok = ok and torch.equal(w.grad.reshape(-1), ref[off : off + w.numel()])
    out[rank] = bool(ok)
    dist.destroy_process_group()


def test_real_model_arena_bucketed_overlapped_exchange_equals_single_allreduce():
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_real_arena_worker, args=(2, port, out), nprocs=2, join=True)
        assert out[0] and out[1]


def _data_eval_worker(rank, world, port, out, gold):
    """Two ranks over the real data path and the evaluators: the unseeded TrainingSampler agrees on one seed and the ranks
    take alternating indices of the same permutation stream; the train loader hands each rank IMS_PER_BATCH / world images;
    the test images are sharded contiguously and the gathered evaluation on rank 0 equals the single-process result."""
    import itertools
    import json
    import tempfile

    import numpy as np
    from PIL import Image

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ["CLUSTER_NUM"] = "800"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import (DatasetCatalog, InferenceSampler, MetadataCatalog, TrainingSampler,
                                build_detection_train_loader, register_all_coco, register_coco_instances)
    from u2seg_amd.data.datasets import load_sem_seg
    from u2seg_amd.evaluation import COCOEvaluator, SemSegEvaluator
    from u2seg_amd.structures import Boxes, Instances

    res = {}
    np.random.seed(100 + rank)  # different numpy streams: the shared seed must come from rank 0
    sampler = TrainingSampler(10)
    mine = list(itertools.islice(iter(sampler), 15))
    seeds = [None] * world
    dist.all_gather_object(seeds, sampler._seed)
    streams = [None] * world
    dist.all_gather_object(streams, mine)
    full = list(itertools.islice(TrainingSampler(10, seed=seeds[0])._infinite_indices(), 30))
    res["sampler"] = seeds[0] == seeds[1] and streams[0] == full[0::2] and streams[1] == full[1::2]
    res["shards"] = list(InferenceSampler(5)._local_indices) == ([0, 1, 2] if rank == 0 else [3, 4])
    register_all_coco(os.path.join(gold, "data_small"))
    fx = json.load(open(os.path.join(gold, "data_golden.json")))
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(os.path.dirname(os.path.dirname(gold)), "configs", "COCO-PanopticSegmentation",
                                     "u2seg_R50_800.yaml"))
    cfg.merge_from_list(fx["input_opts"] + ["DATALOADER.NUM_WORKERS", 0, "SOLVER.IMS_PER_BATCH", 4,
                                           "DATALOADER.ASPECT_RATIO_GROUPING", False])
    batches = list(itertools.islice(iter(build_detection_train_loader(cfg, seed=9)), 2))
    ids = [d["image_id"] for b in batches for d in b]
    all_ids = [None] * world
    dist.all_gather_object(all_ids, ids)
    order = [10, 20, 30, 40]
    want = [order[i] for i in itertools.islice(TrainingSampler(4, seed=9)._infinite_indices(), 8)]
    res["loader"] = all(len(b) == 2 for b in batches) and all_ids[0] == want[0::2] and all_ids[1] == want[1::2]
    # evaluation: rebuild the tiny validation set of the eval fixture (same on both ranks), shard the images, gather
    ev_fx = json.load(open(os.path.join(gold, "eval_golden.json")))
    arrays = np.load(os.path.join(gold, "eval_golden.npz"))
    shared = [tempfile.mkdtemp() if rank == 0 else None]
    dist.broadcast_object_list(shared, src=0)
    root = shared[0]
    img_dir, gt_dir = os.path.join(root, "images"), os.path.join(root, "sem_gt")
    if rank == 0:
        os.makedirs(img_dir)
        os.makedirs(gt_dir)
        for im in ev_fx["images"]:
            stem = im["file_name"][:-4]
            Image.fromarray(np.zeros((im["height"], im["width"], 3), dtype=np.uint8)).save(os.path.join(img_dir, im["file_name"]))
            Image.fromarray(arrays["gt_" + stem], mode="L").save(os.path.join(gt_dir, stem + ".png"))
        json.dump({"images": ev_fx["images"], "annotations": ev_fx["annotations"], "categories": ev_fx["categories"]},
                  open(os.path.join(root, "val.json"), "w"))
    dist.barrier()
    register_coco_instances("tiny_val", {}, os.path.join(root, "val.json"), img_dir)
    DatasetCatalog.get("tiny_val")
    DatasetCatalog.register("tiny_val_sem", lambda: load_sem_seg(gt_dir, img_dir))
    MetadataCatalog.get("tiny_val_sem").set(stuff_classes=[str(c) for c in range(28)], ignore_label=255)
    os.chdir(root)
    inputs, outputs = [], []
    for k in InferenceSampler(len(ev_fx["images"]))._local_indices:
        im, p = ev_fx["images"][k], ev_fx["predictions"][k]
        inst = Instances((im["height"], im["width"]))
        inst.pred_boxes = Boxes(torch.tensor(p["boxes"], dtype=torch.float32))
        inst.scores = torch.tensor(p["scores"], dtype=torch.float32)
        inst.pred_classes = torch.tensor(p["classes"], dtype=torch.int64)
        outputs.append({"instances": inst, "sem_seg": torch.from_numpy(arrays["logits_" + im["file_name"][:-4]])})
        inputs.append({"image_id": im["id"], "file_name": os.path.join(img_dir, im["file_name"])})
    ev = COCOEvaluator("tiny_val", mode="hungarian_matching")
    ev.process(inputs, outputs)
    r = ev.evaluate()
    sem = SemSegEvaluator("tiny_val_sem", mode="hungarian_matching")
    sem.process(inputs, outputs)
    rs = sem.evaluate()
    dist.barrier()  # the mapping file is on disk before any rank reads it in eval mode
    sem2 = SemSegEvaluator("tiny_val_sem", mode="eval")
    sem2.process(inputs, outputs)
    r2 = sem2.evaluate()
    if rank == 0:
        res["eval"] = ({str(k): v for k, v in r["instance_mapping"].items()} == ev_fx["instance_mapping"]
                       and {str(k): v for k, v in rs["semantic_mapping"].items()} == ev_fx["semantic_mapping_file"]
                       and sem2._conf_matrix.tolist() == ev_fx["conf_matrix"]
                       and abs(r2["sem_seg"]["mIoU"] - ev_fx["sem_seg_results"]["mIoU"]) < 1e-9)
    else:
        res["eval"] = r == {} and rs is None and r2 is None
    # ragged row shards of the clustering features meet through one all-gather (cluster/knn.py:gather_rows)
    from u2seg_amd.cluster.knn import gather_rows

    rows = 3 if rank == 0 else 5
    shard = torch.arange(rows * 4, dtype=torch.float32).view(rows, 4) + 100 * rank
    full = gather_rows(shard)
    res["gather_rows"] = (full.shape == (8, 4) and torch.equal(full[:3], torch.arange(12.0).view(3, 4))
                          and torch.equal(full[3:], torch.arange(20.0).view(5, 4) + 100))
    out[rank] = res
    dist.destroy_process_group()


def test_data_path_and_evaluation_two_ranks():
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    port = _free_port()
    with mp.Manager() as m:
        out = m.dict()
        mp.spawn(_data_eval_worker, args=(2, port, out, gold), nprocs=2, join=True)
        results = {r: dict(out[r]) for r in (0, 1)}
    for r in (0, 1):
        assert all(results[r].values()), results
