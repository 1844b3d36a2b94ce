#!/usr/bin/env python
"""Training / evaluation entry point with the CLI contract of the reference's tools/train_net.py:113-164
(--config-file, --num-gpus, --resume, --eval-only, trailing KEY VALUE overrides).  One process per GPU: launch N > 1 with
    python -m torch.distributed.run --nproc-per-node N tools/train_net.py --num-gpus N --config-file ...
Data: when the dataset DATASETS.TRAIN names is on disk (builtin registration under ./datasets or $DETECTRON2_DATASETS,
u2seg_amd/data/datasets.py) batches come from the real pipeline - DatasetMapper in DataLoader workers, aspect-ratio
grouping, DevicePrefetcher into HBM; otherwise (or with DATASETS.TRAIN ("synthetic",)) from the synthetic generator.
Deviation (recorded in DESIGN.md): the reference hard-wires --eval-only to True (engine/defaults.py:109); here it is a flag."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from u2seg_amd.checkpoint import DetectionCheckpointer  # noqa: E402
from u2seg_amd.config import get_cfg  # noqa: E402
from u2seg_amd.data import (DatasetCatalog, DevicePrefetcher, MetadataCatalog, build_detection_train_loader,  # noqa: E402
                            make_synthetic_batch, register_all_coco)
from u2seg_amd.engine import SimpleTrainer, default_argument_parser, launch_info  # noqa: E402
from u2seg_amd.modeling import build_model  # noqa: E402
from u2seg_amd.solver import build_lr_scheduler, build_optimizer  # noqa: E402
from u2seg_amd.utils.env import configure_host_threads  # noqa: E402


def setup(args):
    cfg = get_cfg()
    cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    cfg.freeze()
    return cfg


def real_batches(cfg, device):
    """Iterator over device-resident training batches of the real dataset, or None when it is not on disk."""
    names = [n for n in cfg.DATASETS.TRAIN if n != "synthetic"]
    if not names:
        return None
    register_all_coco()
    for n in names:
        meta = MetadataCatalog.get(n)
        if n not in DatasetCatalog or not os.path.isfile(meta.get("json_file", "")) or not os.path.isdir(meta.get("image_root", "")):
            return None
    loader = build_detection_train_loader(cfg, seed=None if cfg.SEED < 0 else cfg.SEED)
    return iter(DevicePrefetcher(loader, device) if str(device).startswith("cuda") else loader)


def evaluate_on_disk_datasets(cfg, model, eval_mode, device):
    """The reference's Trainer.test (engine/defaults.py:591-640) for the DATASETS.TEST entries that are on disk: first run
    with --eval-mode hungarian_matching (writes ./hungarian_matching/*.json), then with --eval-mode eval.  None when no
    test dataset is available."""
    from u2seg_amd.data import build_detection_test_loader
    from u2seg_amd.evaluation import build_evaluator, inference_on_dataset

    register_all_coco()
    results = {}
    for name in cfg.DATASETS.TEST:
        meta = MetadataCatalog.get(name)
        if name not in DatasetCatalog or not os.path.isfile(meta.get("json_file", "")):
            continue
        loader = build_detection_test_loader(cfg, name)
        stream = DevicePrefetcher(loader, device) if str(device).startswith("cuda") else loader
        results[name] = inference_on_dataset(model, stream, build_evaluator(cfg, name, eval_mode=eval_mode))
    return results or None


def main(args):
    configure_host_threads()
    rank, local_rank, world = launch_info()
    cfg = setup(args)
    if cfg.MODEL.DEVICE.startswith("cuda"):
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo")
    c = cfg.clone()
    c.defrost()
    if cfg.MODEL.DEVICE == "cuda":
        c.MODEL.DEVICE = "cuda:%d" % local_rank
    model = build_model(c)
    per_gpu = max(1, cfg.SOLVER.IMS_PER_BATCH // world)
    if args.eval_only:
        # tools/train_net.py:135-141 of the reference: weights from MODEL.WEIGHTS (or the last checkpoint with --resume)
        if cfg.MODEL.WEIGHTS and os.path.isfile(cfg.MODEL.WEIGHTS):
            DetectionCheckpointer(model, cfg.OUTPUT_DIR).resume_or_load(cfg.MODEL.WEIGHTS, resume=args.resume)
        results = evaluate_on_disk_datasets(cfg, model, args.eval_mode, c.MODEL.DEVICE)
        if results is not None:
            if rank == 0:
                print(results)
            return results
        model.eval()
        with torch.no_grad():
            out = model(make_synthetic_batch(per_gpu, start_index=rank * per_gpu, device=c.MODEL.DEVICE))
        if rank == 0:
            print("inference ok: %d images, %d instances in image 0" % (len(out), len(out[0]["instances"])))
        return out
    model.train()
    opt = build_optimizer(cfg, model)
    sched = build_lr_scheduler(cfg, opt)
    trainer = SimpleTrainer(model, opt, sched)
    checkpointer = DetectionCheckpointer(model, cfg.OUTPUT_DIR, save_to_disk=rank == 0, optimizer=opt, scheduler=sched)
    start_iter = 0
    if (cfg.MODEL.WEIGHTS and os.path.isfile(cfg.MODEL.WEIGHTS)) or args.resume:
        rest = checkpointer.resume_or_load(cfg.MODEL.WEIGHTS, resume=args.resume)
        # engine/defaults.py:410-421: only a run that found its own last_checkpoint continues at iteration + 1
        start_iter = int(rest.get("iteration", -1)) + 1 if args.resume and checkpointer.has_checkpoint() else 0
        sched.resume_at(start_iter)  # lr of iteration start_iter, milestones counted from iteration 0
    elif cfg.MODEL.WEIGHTS and rank == 0:
        print("MODEL.WEIGHTS %s not found: random initialisation" % cfg.MODEL.WEIGHTS)
    stream = real_batches(cfg, c.MODEL.DEVICE)
    if rank == 0:
        print("data: %s" % ("real pipeline over %s" % (cfg.DATASETS.TRAIN,) if stream is not None else "synthetic generator"))
    t0 = time.time()
    for it in range(start_iter, cfg.SOLVER.MAX_ITER):
        if stream is not None:
            batch = next(stream)
        else:
            batch = make_synthetic_batch(per_gpu, start_index=(it * world + rank) * per_gpu, device=c.MODEL.DEVICE)
        trainer.run_step(batch)
        if rank == 0 and (it % 20 == 0 or it == cfg.SOLVER.MAX_ITER - 1):
            total = trainer.check_finite()
            print("iter %d  total_loss %.4f  lr %.6f  %.2f s/iter" % (it, total, opt.lr, (time.time() - t0) / (it - start_iter + 1)))
        if (it + 1) % cfg.SOLVER.CHECKPOINT_PERIOD == 0 or it == cfg.SOLVER.MAX_ITER - 1:
            checkpointer.save("model_%07d" % it if it < cfg.SOLVER.MAX_ITER - 1 else "model_final", iteration=it)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main(default_argument_parser().parse_args())
