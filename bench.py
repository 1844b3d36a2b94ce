"""Headline benchmark: training images/s of u2seg_R50_800 (Panoptic-FPN, cascade ROI heads) on synthetic
COCO-shaped 3x800x1333 batches, bf16 activations / fp32 master weights, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement).  `roofline` is measured live with HIP events
around every launch of the dominant kernel family (the MFMA implicit-GEMM conv) inside the timed region;
`cpu_baseline` times the CPU oracle (oracle/) on a bounded sample (rank 0, N = 1 only)."""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0  # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
CONV_STACK_TRAIN_GFLOP_PER_IMAGE = 885.83  # BASELINE.md section 2 (ResNet-50 + FPN fwd + dgrad + wgrad)


class KernelTimer:
    """Records a HIP event pair around selected C-ABI launches (on torch's current stream, where they run)."""

    def __init__(self, names):
        self.names = set(names)
        self.records = []  # (name, flops, start, end)
        self.enabled = False

    def install(self):
        from u2seg_amd import _hip

        orig = _hip.call
        timer = self

        def timed_call(name, *args):
            if not timer.enabled or name not in timer.names:
                return orig(name, *args)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            orig(name, *args)
            e.record()
            timer.records.append((name, timer.flops(name, args), s, e, tuple(a for a in args if isinstance(a, int)),
                                  timer.alg_bytes(name, args)))

        _hip.call = timed_call
        import u2seg_amd.layers.functional as F

        F._hip.call = timed_call

    @staticmethod
    def alg_bytes(name, a):
        """Algorithmic HBM bytes of one launch: every operand read once, the result written once (bf16 activations)."""
        if name == "u2_conv_igemm":
            b, hin, win, c, ho, wo, n, kh, kw, accum = a[5], a[6], a[7], a[8], a[10], a[11], a[12], a[14], a[15], a[21]
            return 2.0 * (b * hin * win * c + n * kh * kw * c + b * ho * wo * n * (2 if accum else 1))
        if name == "u2_conv_wgrad":
            b, hin, win, c, ho, wo, n, kh, kw = a[3], a[4], a[5], a[6], a[8], a[9], a[10], a[12], a[13]
            return 2.0 * (b * hin * win * c + b * ho * wo * n) + 4.0 * n * kh * kw * c
        return 0.0

    @staticmethod
    def flops(name, a):
        if name == "u2_conv_igemm":
            # (in, wt, out, bias, stats, B, Hin, Win, C, in_ld, Hout, Wout, N, out_ld, KH, KW, ph, pw, mul, div, ...)
            b, c, ho, wo, n, kh, kw, div = a[5], a[8], a[10], a[11], a[12], a[14], a[15], a[19]
            return 2.0 * b * ho * wo * n * kh * kw * c / (div * div)
        if name == "u2_conv_wgrad":
            # (x, dy, dw, B, Hin, Win, C, x_ld, Hout, Wout, N, dy_ld, KH, KW, ...)
            b, c, ho, wo, n, kh, kw = a[3], a[6], a[8], a[9], a[10], a[12], a[13]
            return 2.0 * b * ho * wo * n * kh * kw * c
        return 0.0

    def summary(self):
        out = {}
        for name, fl, s, e, _, by in self.records:
            d = out.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += fl
            d["bytes"] += by
        return out


CPU_BASELINE_THREADS = 32     # more threads than this slow the oracle's small fp32 convolutions down (256 threads: 15x slower)
CPU_BASELINE_TIMEOUT_S = 150  # the default bench.py run must stay within a few minutes


def cpu_baseline_worker(sample_hw=(800, 1333)):
    """Times the CPU oracle's train iteration (fwd + bwd, fp32) on 1 synthetic image; returns the JSON object."""
    from oracle.model import OracleModel
    import u2seg_amd.data as data

    cores = min(os.cpu_count() or 1, CPU_BASELINE_THREADS)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    om = OracleModel.from_config_file(os.path.join(ROOT, "configs", "COCO-PanopticSegmentation", "u2seg_R50_800.yaml"))
    nimg = 2  # BASELINE.json configs[0]: the reference's own CPU case is 2 images, 1 train iteration
    batch = data.make_synthetic_batch(nimg, height=sample_hw[0], width=sample_hw[1])
    t0 = time.time()
    losses = om.train_forward(batch)
    sum(losses.values()).backward()
    dt = time.time() - t0
    return {"value": nimg / dt, "unit": "img/s", "cores": cores, "kind": "port",
            "sample": "%d synthetic %dx%d images, 1 fwd+bwd iteration of the fp32 CPU oracle on %d of the host's %d hardware "
                      "threads (%.1f s, includes first-call overheads)" % (nimg, sample_hw[0], sample_hw[1], cores,
                                                                           os.cpu_count() or 1, dt)}


def cpu_baseline():
    """Runs the oracle in a child process with a hard time limit, so that a slow host cannot stall the benchmark."""
    import subprocess

    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only"], capture_output=True, text=True,
                             timeout=CPU_BASELINE_TIMEOUT_S, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": "img/s", "cores": CPU_BASELINE_THREADS, "kind": "port",
                "sample": "oracle child process failed: " + out.stderr.strip()[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": "img/s", "cores": CPU_BASELINE_THREADS, "kind": "port",
                "sample": "1 synthetic 800x1333 image did not finish one fp32 oracle iteration within %d s" % CPU_BASELINE_TIMEOUT_S}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="internal: time the CPU oracle and print its JSON object")
    ap.add_argument("--per-layer", action="store_true", help="debug: print the conv launches grouped by shape")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline_worker()))
        return

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(dev))

    from u2seg_amd import _hip
    from u2seg_amd.config import get_cfg
    from u2seg_amd.data import make_synthetic_batch
    from u2seg_amd.engine import SimpleTrainer
    from u2seg_amd.modeling import build_model
    from u2seg_amd.solver import build_lr_scheduler, build_optimizer

    _hip.load()
    timer = KernelTimer(["u2_conv_igemm", "u2_conv_wgrad"])
    timer.install()

    torch.manual_seed(1234)  # identical initial weights on every rank
    cfg = get_cfg()
    cfg.merge_from_file(os.path.join(ROOT, "configs", "COCO-PanopticSegmentation", "u2seg_R50_800.yaml"))
    cfg.merge_from_list(["MODEL.DEVICE", dev, "SOLVER.IMS_PER_BATCH", args.batch * world])
    model = build_model(cfg)
    model.train()
    opt = build_optimizer(cfg, model)
    sched = build_lr_scheduler(cfg, opt)
    trainer = SimpleTrainer(model, opt, sched)
    # a few cached synthetic batches resident in HBM (tools/benchmark.py:108-115 caches 100 batches the same way)
    nb = 2
    batches = [make_synthetic_batch(args.batch, start_index=(rank * nb + i) * args.batch, height=args.height,
                                    width=args.width, device=dev) for i in range(nb)]
    torch.manual_seed(1000 + rank)  # per-rank sampling randomness

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        trainer.run_step(batches[i % nb])
    barrier()
    # HIP-event pairs around every conv launch cost ~5% of a step on the host side, so only the last step(s) of the
    # timed region carry them: the roofline numbers are a sample of the timed region, `value` stays (almost) undisturbed.
    sampled = max(1, args.steps // 8)
    t0 = time.time()
    for i in range(args.steps):
        timer.enabled = i >= args.steps - sampled
        trainer.run_step(batches[i % nb])
    barrier()
    dt = time.time() - t0
    timer.enabled = False
    total = trainer.check_finite()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        imgs_per_s = args.batch * world * args.steps / dt
        ks = timer.summary()
        dom = max(ks.items(), key=lambda kv: kv[1]["ms"]) if ks else None
        roofline = None
        if dom is not None:
            name, d = dom
            achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
            traffic, traffic_note = None, "no PMC profile committed"
            pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
            if os.path.exists(pmc):
                pj = json.load(open(pmc))
                key = {"u2_conv_igemm": "conv_igemm", "u2_conv_wgrad": "conv_wgrad"}[name]
                traffic = pj[key]["hbm_bytes_per_launch"]
                traffic_note = "HBM bytes per launch from profiles/r01_pmc_traffic.json (" + pj["note"] + ")"
            roofline = {"kernel": {"u2_conv_igemm": "conv_igemm_kernel (fwd + dgrad launches)",
                                   "u2_conv_wgrad": "conv_wgrad_kernel"}[name],
                        "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                        "frac": achieved / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_note": traffic_note,
                        "algorithmic_bytes_per_launch_avg": d["bytes"] / d["launches"],
                        "launches_per_step": d["launches"] / sampled, "avg_launch_ms": d["ms"] / d["launches"],
                        "sampled_steps": "the last %d of the %d timed steps carry the HIP events" % (sampled, args.steps),
                        "flop_per_launch_avg": d["flops"] / d["launches"],
                        "kernel_ms_per_step": {k: v["ms"] / sampled for k, v in ks.items()},
                        "kernel_tflops": {k: v["flops"] / (v["ms"] * 1e-3) / 1e12 for k, v in ks.items()},
                        "conv_stack_frac_of_peak_e2e": imgs_per_s / world * CONV_STACK_TRAIN_GFLOP_PER_IMAGE * 1e9 / (PEAK_BF16_TFLOPS * 1e12)}
        out = {
            "metric": "training images/sec (whole node) u2seg_R50_800 @ 800x1333",
            "value": imgs_per_s, "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "u2seg_R50_800.yaml bf16, batch %d per GPU, %dx%d synthetic COCO-panoptic batches, "
                                   "random init, SGD+per-param clip" % (args.batch, args.height, args.width),
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world},
            "final_total_loss": total,
            "roofline": roofline,
        }
        if args.per_layer:
            agg = {}
            for name, fl, s, e, shape, _ in timer.records:
                d = agg.setdefault((name, shape), [0, 0.0, 0.0])
                d[0] += 1
                d[1] += s.elapsed_time(e)
                d[2] += fl
            for (name, shape), d in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
                print("LAYER %-14s %-70s n=%3d  %7.3f ms/step  %7.1f TF/s" % (name, shape, d[0] / sampled, d[1] / sampled,
                                                                             d[2] / (d[1] * 1e-3) / 1e12), file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
