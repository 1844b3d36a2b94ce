"""Headline benchmark: training images/s of u2seg_R50_800 (Panoptic-FPN, cascade ROI heads) on synthetic
COCO-shaped 3x800x1333 batches, bf16 activations / fp32 master weights, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement).  `roofline` is measured live with HIP events
around every launch of the dominant kernel family (the MFMA implicit-GEMM conv) inside the timed region;
`cpu_baseline` times the CPU oracle (oracle/) on a bounded sample (rank 0, N = 1 only).

The default workload is BASELINE.json's metric (configs[1]: training images/s).  Two more workloads make the other
single-GPU configurations driver-measurable with the same JSON contract:
    python bench.py --workload kmeans   # configs[3]: Lloyd iterations over 1M x 768 synthetic DINO embeddings, K = 300 (s/iter)
    python bench.py --workload infer    # configs[4]: u2seg_eval_800 panoptic inference, batch 32 (img/s)"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_TBPS = 8.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
CONV_STACK_TRAIN_GFLOP_PER_IMAGE = 885.83  # BASELINE.md section 2 (ResNet-50 + FPN fwd + dgrad + wgrad)


class KernelTimer:
    """Records a HIP event pair around selected C-ABI launches (on torch's current stream, where they run)."""

    def __init__(self, names):
        self.names = set(names)
        self.records = []  # (name, flops, start, end, int args, bytes) of the steps timed in one stream
        self.overlapped = []  # the same for the steps timed with the side streams active
        self.enabled = False
        self.tag = "serial"

    def install(self):
        from u2seg_amd import _hip

        timer = self

        def wrap(orig):
            def timed_call(name, *args):
                if not timer.enabled or name not in timer.names:
                    return orig(name, *args)
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                rc = orig(name, *args)
                e.record()
                if rc in (None, 0):   # (a launcher that answered "shape not served" launched nothing)
                    (timer.records if timer.tag == "serial" else timer.overlapped).append(
                        (name, timer.flops(name, args), s, e, tuple(a for a in args if isinstance(a, int)), timer.alg_bytes(name, args)))
                return rc

            return timed_call

        _hip.call = wrap(_hip.call)
        _hip.call_status = wrap(_hip.call_status)

    @staticmethod
    def alg_bytes(name, a):
        """Algorithmic HBM bytes of one launch: every operand read once, the result written once (bf16 activations)."""
        if name == "u2_conv_igemm":
            b, hin, win, c, ho, wo, n, kh, kw, accum = a[5], a[6], a[7], a[8], a[10], a[11], a[12], a[14], a[15], a[21]
            return 2.0 * (b * hin * win * c + n * kh * kw * c + b * ho * wo * n * (2 if accum else 1))
        if name in ("u2_conv_wgrad", "u2_conv_wgrad_into"):
            b, hin, win, c, ho, wo, n, kh, kw = a[3], a[4], a[5], a[6], a[8], a[9], a[10], a[12], a[13]
            return 2.0 * (b * hin * win * c + b * ho * wo * n) + 4.0 * n * kh * kw * c
        if name in ("u2_kmeans_assign", "u2_kmeans_update"):
            return 4.0 * a[4] * a[5]  # x read once (the fused ideal reads it once per iteration)
        if name == "u2_kmeans_assign_shadow":   # (x, shadow, c, ws, labels, N, D, K, ...)
            return 4.0 * a[5] * a[6]
        if name == "u2_conv1x1_bwd_fused":   # (x, dy, wt, dx, dw, M, C, x_ld, N, ...): dy and x read once, dx written, dW
            m, c, n = a[5], a[6], a[8]
            return 2.0 * m * (n + 2 * c) + 2.0 * n * c + 4.0 * n * c
        if name == "u2_conv1x1_bwd_fused_bn":   # (x, dz, y, k1, k2, k3, wt, dx, dw, M, C, x_ld, N, ...): dz, y and x read once
            m, c, n = a[9], a[10], a[12]
            return 2.0 * m * (2 * n + 2 * c) + 2.0 * n * c + 4.0 * n * c
        return 0.0

    @staticmethod
    def flops(name, a):
        if name == "u2_conv1x1_bwd_fused":   # data gradient + weight gradient
            return 4.0 * a[5] * a[6] * a[8]
        if name == "u2_conv1x1_bwd_fused_bn":
            return 4.0 * a[9] * a[10] * a[12]
        if name == "u2_conv_igemm":
            # (in, wt, out, bias, stats, B, Hin, Win, C, in_ld, Hout, Wout, N, out_ld, KH, KW, ph, pw, mul, div, ...)
            b, c, ho, wo, n, kh, kw, div = a[5], a[8], a[10], a[11], a[12], a[14], a[15], a[19]
            return 2.0 * b * ho * wo * n * kh * kw * c / (div * div)
        if name in ("u2_conv_wgrad", "u2_conv_wgrad_into"):
            # (x, dy, dw, B, Hin, Win, C, x_ld, Hout, Wout, N, dy_ld, KH, KW, ...)
            b, c, ho, wo, n, kh, kw = a[3], a[6], a[8], a[9], a[10], a[12], a[13]
            return 2.0 * b * ho * wo * n * kh * kw * c
        if name == "u2_kmeans_assign":
            # (x, c, cnorm_ws, labels, N, D, K): |c|^2 - 2 x.c for every (point, centroid) pair
            return 2.0 * a[4] * a[5] * a[6]
        if name == "u2_kmeans_assign_shadow":
            return 2.0 * a[5] * a[6] * a[7]
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
CPU_BASELINE_TIMEOUT_S = 200  # the default bench.py run must stay within a few minutes
CFG_DIR = os.path.join(ROOT, "configs", "COCO-PanopticSegmentation")
KMEANS_D, KMEANS_K = 768, 300


def _timed_iterations(fn, warmup=1, timed=3):
    """1 warm-up + 3 timed iterations; returns the mean seconds of the timed ones."""
    for _ in range(warmup):
        fn()
    t0, c0 = time.time(), time.process_time()
    for _ in range(timed):
        fn()
    dt = time.time() - t0
    _timed_iterations.cpus_effective = (time.process_time() - c0) / max(dt, 1e-9)   # CPU seconds per wall second of the sample
    return dt / timed


def cpu_baseline_worker(workload):
    """Times the CPU oracle on a bounded sample of the workload (1 warm-up + 3 timed iterations); returns the JSON object."""
    cores = min(os.cpu_count() or 1, CPU_BASELINE_THREADS)
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    host = "%d of the host's %d hardware threads" % (cores, os.cpu_count() or 1)
    if workload == "kmeans":
        from oracle import ops

        n = 20000
        g = torch.Generator().manual_seed(0)
        x = torch.randn((n, KMEANS_D), generator=g)
        c = x[torch.randperm(n, generator=g)[:KMEANS_K]].clone()

        def it():
            cl = ops.kmeans_assign(x, c)
            ops.kmeans_update(x, cl, KMEANS_K)

        dt = _timed_iterations(it)
        return {"value": dt * (1000000 / n), "unit": "s/iter", "cores": cores, "kind": "port",
                "sample": "Lloyd iterations (oracle/ops.py, chunked plain-torch form of nn_utils.py:325-364) over %d x %d points, "
                          "K = %d, 1 warm-up + 3 timed on %s: %.2f s/iter, scaled linearly to N = 1M" % (n, KMEANS_D, KMEANS_K, host, dt)}
    from oracle.model import OracleModel
    import u2seg_amd.data as data

    if workload == "infer":
        om = OracleModel.from_config_file(os.path.join(CFG_DIR, "u2seg_eval_800.yaml"))
        batch = [{k: v for k, v in x.items() if k != "instances"} for x in data.make_synthetic_batch(1)]
        with torch.no_grad():
            dt = _timed_iterations(lambda: om.inference(batch))
        return {"value": 1 / dt, "unit": "img/s", "cores": cores, "kind": "port",
                "sample": "1 synthetic 800x1333 image, full panoptic inference of the fp32 CPU oracle, 1 warm-up + 3 timed on %s "
                          "(%.1f s per image)" % (host, dt)}
    om = OracleModel.from_config_file(os.path.join(CFG_DIR, "u2seg_R50_800.yaml"))
    nimg = 2  # BASELINE.json configs[0]: the reference's own CPU case is 2 images, 1 train iteration
    batch = data.make_synthetic_batch(nimg)

    def it():
        for p in om.parameters().values():
            p.grad = None
        losses = om.train_forward(batch)
        sum(losses.values()).backward()

    dt = _timed_iterations(it)
    return {"value": nimg / dt, "unit": "img/s", "cores": cores, "kind": "port",
            "sample": "%d synthetic 800x1333 images, fwd+bwd iteration of the fp32 CPU oracle, 1 warm-up + 3 timed on %s "
                      "(%.1f s per iteration)" % (nimg, host, dt)}


def cpu_baseline(workload):
    """Runs the oracle in a child process with a hard time limit, so that a slow host cannot stall the benchmark."""
    import subprocess

    unit = "s/iter" if workload == "kmeans" else "img/s"
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", workload],
                             capture_output=True, text=True, timeout=CPU_BASELINE_TIMEOUT_S,
                             env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
        for line in reversed(out.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "unit": unit, "cores": CPU_BASELINE_THREADS, "kind": "port",
                "sample": "oracle child process failed: " + out.stderr.strip()[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "unit": unit, "cores": CPU_BASELINE_THREADS, "kind": "port",
                "sample": "the oracle sample did not finish 1 + 3 iterations within %d s" % CPU_BASELINE_TIMEOUT_S}


def _pmc_traffic(kernel_key):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary (profiles/rNN_pmc_traffic.json)."""
    for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            pj = json.load(open(path))
            if kernel_key in pj:
                return pj[kernel_key]["hbm_bytes_per_launch"], "HBM bytes per launch from profiles/%s (%s)" % (name, pj["note"])
    return None, "no PMC profile committed for this kernel"


def conv_roofline(timer, sampled, steps, imgs_per_s_per_gpu=None):
    ks = timer.summary()
    conv = {k: v for k, v in ks.items() if k in ("u2_conv_igemm", "u2_conv_wgrad", "u2_conv_wgrad_into")}
    if not conv:
        return None
    name, d = max(conv.items(), key=lambda kv: kv[1]["ms"])
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    traffic, note = _pmc_traffic({"u2_conv_igemm": "conv_igemm"}.get(name, "conv_wgrad"))
    # launches whose whole reduction is <= 256 deep (1x1 convs on <= 256 channels) are HBM-bound by construction: report the two
    # groups next to the all-launch average the `frac` is computed from
    deep = [r for r in timer.records if r[0] == name and TimerShape.k_depth(r) > 256]
    shallow = [r for r in timer.records if r[0] == name and TimerShape.k_depth(r) <= 256]

    def tf(rs):
        ms = sum(r[2].elapsed_time(r[3]) for r in rs)
        return (sum(r[1] for r in rs) / (ms * 1e-3) / 1e12, ms / sampled) if rs and ms > 0 else (None, 0.0)

    def tbs(rs):
        ms = sum(r[2].elapsed_time(r[3]) for r in rs)
        return sum(r[5] for r in rs) / (ms * 1e-3) / 1e12 if rs and ms > 0 else None

    out = {"kernel": {"u2_conv_igemm": "implicit-GEMM conv, forward + data-gradient launches (conv_halo_kernel / conv_tile_kernel incl. "
                                       "its stream-K form / conv_igemm_kernel: every kernel behind u2_conv_igemm)",
                      "u2_conv_wgrad": "conv_wgrad_kernel", "u2_conv_wgrad_into": "conv_wgrad_kernel"}[name],
           "bound": "mfma", "achieved": achieved, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_BF16_TFLOPS,
           "traffic": traffic, "traffic_note": note,
           "algorithmic_bytes_per_launch_avg": d["bytes"] / d["launches"],
           "launches_per_step": d["launches"] / sampled, "avg_launch_ms": d["ms"] / d["launches"],
           "sampled_steps": "the last %d of the %d timed steps carry these HIP events and run every kernel in one stream (the "
                            "normal step overlaps the semantic head and the weight gradients on further streams, which times "
                            "co-running kernels into each other: see kernel_ms_per_step_overlapped, the %d steps before)"
                            % (sampled, steps, sampled),
           "flop_per_launch_avg": d["flops"] / d["launches"],
           "kernel_ms_per_step": {k: v["ms"] / sampled for k, v in ks.items()},
           "kernel_tflops": {k: v["flops"] / (v["ms"] * 1e-3) / 1e12 for k, v in ks.items() if v["ms"] > 0},
           "by_reduction_depth": {"K>256 (MFMA-bound)": {"tflops": tf(deep)[0], "ms_per_step": tf(deep)[1]},
                                  "K<=256 (HBM-bound 1x1 layers)": {"tflops": tf(shallow)[0], "ms_per_step": tf(shallow)[1],
                                                                   "algorithmic_TB_per_s": tbs(shallow)}}}
    # Speed of light of THIS launch mix: every launch at max(flop / MFMA peak, algorithmic bytes / HBM peak).  A 1x1 conv moves
    # 2 (K + N) bytes per pixel for 2 K N flop - 51 flop/B on the res2 layers, 205 on res4 - far left of the 312 flop/B ridge of
    # 2.5 PFLOP/s over 8 TB/s, so a third of the launches is bandwidth-bound by construction and the mix cannot reach the MFMA peak.
    recs = [r for r in timer.records if r[0] == name]
    sol_s = sum(max(r[1] / (PEAK_BF16_TFLOPS * 1e12), r[5] / (PEAK_HBM_TBPS * 1e12)) for r in recs)
    if sol_s > 0:
        att = sum(r[1] for r in recs) / sol_s / 1e12
        out["launch_mix_roofline"] = {"hbm_peak_TB_per_s": PEAK_HBM_TBPS, "attainable_tflops": att,
                                      "frac_of_attainable": achieved / att,
                                      "launches_hbm_bound": sum(1 for r in recs if r[5] / (PEAK_HBM_TBPS * 1e12) >
                                                                r[1] / (PEAK_BF16_TFLOPS * 1e12)) / sampled}
    if timer.overlapped:
        ov = {}
        for rname, _fl, s_, e_, _i, _b in timer.overlapped:
            ov[rname] = ov.get(rname, 0.0) + s_.elapsed_time(e_)
        out["kernel_ms_per_step_overlapped"] = {k: v / sampled for k, v in ov.items()}
    if imgs_per_s_per_gpu is not None:
        out["conv_stack_frac_of_peak_e2e"] = imgs_per_s_per_gpu * CONV_STACK_TRAIN_GFLOP_PER_IMAGE * 1e9 / (PEAK_BF16_TFLOPS * 1e12)
    return out


class TimerShape:
    @staticmethod
    def k_depth(rec):
        """Reduction depth KH * KW * C of a recorded conv launch (record[4] = the integer arguments of the call)."""
        name, ints = rec[0], rec[4]
        if name == "u2_conv_igemm":   # (B, Hin, Win, C, in_ld, Hout, Wout, N, out_ld, KH, KW, ...)
            return ints[3] * ints[9] * ints[10]
        return 1 << 30


def per_layer_report(timer, sampled):
    agg = {}
    for name, fl, s, e, shape, _ in timer.records:
        d = agg.setdefault((name, shape), [0, 0.0, 0.0])
        d[0] += 1
        d[1] += s.elapsed_time(e)
        d[2] += fl
    for (name, shape), d in sorted(agg.items(), key=lambda kv: -kv[1][1])[:90]:
        print("LAYER %-14s %-70s n=%3d  %7.3f ms/step  %7.1f TF/s" % (name, shape, d[0] / sampled, d[1] / sampled,
                                                                     d[2] / (d[1] * 1e-3) / 1e12), file=sys.stderr)


def _cgroup_cpu_stat():
    """nr_throttled / throttled_usec of this container's CPU controller (cgroup v2), {} where it is not readable."""
    out = {}
    try:
        with open("/sys/fs/cgroup/cpu.stat") as fh:
            for line in fh:
                k, v = line.split()
                out[k] = int(v)
        with open("/sys/fs/cgroup/cpu.max") as fh:
            out["cpu_max"] = fh.read().strip()
    except (OSError, ValueError):
        pass
    return out


def _per_step_stats(marks, sampled):
    """Per-step times of the timed region from the (host time, HIP event) mark in front of every step: the mean hides a single
    allocation or host hiccup in a short region, the median and the minimum do not.  `device_ms` is the distance of the marks on
    the main stream, `host_ms` the distance of the host clock at submission (equal when the step ends in a host read-back)."""
    dev = [marks[i][1].elapsed_time(marks[i + 1][1]) for i in range(len(marks) - 1)]
    host = [(marks[i + 1][0] - marks[i][0]) * 1e3 for i in range(len(marks) - 1)]

    def stats(v):
        w = sorted(v)
        return {"median": w[len(w) // 2], "min": w[0], "max": w[-1], "mean": sum(w) / len(w)}

    over = dev[: len(dev) - sampled] or dev
    return {"device_ms": stats(dev), "host_ms": stats(host), "device_ms_overlapped_steps": stats(over),
            "device_ms_each": [round(x, 2) for x in dev], "host_ms_each": [round(x, 2) for x in host],
            "note": "the last %d step(s) run in one stream with HIP events around every conv launch (roofline sample)" % sampled}


def cpu_baselines_parallel(workloads):
    """The oracle samples of several workloads at once, one child process each (32 threads each, GPUs hidden)."""
    import concurrent.futures as cf

    with cf.ThreadPoolExecutor(max_workers=len(workloads)) as ex:
        return dict(zip(workloads, ex.map(cpu_baseline, workloads)))


def _spawn_ranks(n):
    """Re-runs this command line under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1, a free port)."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    from u2seg_amd.utils.env import configure_host_threads

    # the OpenMP pool sized for the container's CPU quota, not the host's cores (U2_HOST_THREADS=0: leave it alone, for A/B runs)
    if os.environ.get("U2_HOST_THREADS", "1") != "0":
        configure_host_threads()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["train", "kmeans", "infer"], default="train")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (train: 16, infer: 32)")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--kmeans-n", type=int, default=1000000, help="points in total (sharded over the GPUs)")
    ap.add_argument("--kmeans-data", choices=["randn", "mixture"], default="mixture",
                    help="SURVEY 8(d) config 4: x = randn(N, 768) (base case) or the mixture of 300 Gaussians, sigma 0.5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="train workload only: do not append the k-means (config 4) and inference (config 5) measurements as "
                         "`extra_workloads` (they run after the timed training region, N = 1 only)")
    ap.add_argument("--serial", action="store_true",
                    help="every step in one stream (no side / second stream): the mode the roofline sample steps run in; used "
                         "for the rocprofv3 summary that the per-kernel durations of the roofline object are checked against")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="internal: time the CPU oracle and print its JSON object")
    ap.add_argument("--per-layer", action="store_true", help="debug: print the conv launches grouped by shape")
    args = ap.parse_args()
    if args.cpu_baseline_only:
        res = cpu_baseline_worker(args.workload)
        # what the threads actually got: the container's CPU quota is shared by the oracle samples running side by side
        from u2seg_amd.utils.env import cgroup_cpu_quota

        res["cpus_effective"] = round(getattr(_timed_iterations, "cpus_effective", 0.0), 1)
        res["cgroup_cpu_quota"] = cgroup_cpu_quota()
        print(json.dumps(res))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks here (one process per GPU, the launcher the driver would
        # use) and pass rank 0's JSON line through; under torchrun the environment already carries WORLD_SIZE
        sys.exit(_spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is what runs" % (args.gpus, world), file=sys.stderr)
    # U2_BENCH_SHARE_GPU=1 (a smoke test of the N > 1 code path on a one-GPU box, never a measurement): every rank on cuda:0 over
    # gloo - RCCL refuses two ranks on one device.  The line it prints says so in `data`.
    share_gpu = world > 1 and os.environ.get("U2_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))

    from u2seg_amd import _hip

    _hip.load()
    timer = KernelTimer(["u2_conv_igemm", "u2_conv_wgrad", "u2_conv_wgrad_into", "u2_conv1x1_bwd_fused", "u2_conv1x1_bwd_fused_bn",
                         "u2_kmeans_assign", "u2_kmeans_assign_shadow",
                         "u2_kmeans_update"])
    timer.install()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step_fn, steps, warmup):
        """W untimed steps, then exactly K timed ones between barriers; MAX over ranks.  Only the last step(s) carry the
        per-launch HIP events (they cost host time): the roofline numbers are a sample of the timed region."""
        from u2seg_amd.layers import functional as Fn

        timer.records, timer.overlapped = [], []
        for i in range(warmup):
            # the first warm-up step runs in one stream like the sampled steps at the end of the timed region: the library's
            # per-stream scratch (stream-K hand-over slots, weight-gradient partial tiles: hipMalloc on first use of a stream)
            # then exists for both modes before the clock starts
            Fn.set_stream_overlap(not args.serial and not (i == 0 and warmup > 1))
            step_fn(i)
        Fn.set_stream_overlap(not args.serial)
        barrier()
        # one step in sixteen carries the roofline sample (round 6: was one in eight - the serial sampled steps are ~5 % longer than
        # the overlapped ones, so the sample itself lowers the mean it is reported beside; K = 20: one step, 0.25 % of the mean)
        sampled = max(1, steps // 16)

        marks = []
        # the collector's generation-2 passes walk the whole module tree (tens of ms of host time in the middle of a step that is host-bound in
        # places): collect once here, keep what is alive out of later passes; U2_BENCH_GC=0 leaves the collector alone
        import gc

        manage_gc = os.environ.get("U2_BENCH_GC", "1") != "0"
        if manage_gc:
            gc.collect()
            gc.freeze()
        cg0, cpu0 = _cgroup_cpu_stat(), time.process_time()
        t0 = time.time()
        for i in range(steps):
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((time.time(), ev))
            # The last `sampled` steps carry the per-launch HIP events of the roofline object and run all their kernels in ONE
            # stream: in the normal step the semantic head and the weight gradients run on further streams, and a kernel that
            # shares the CUs with another one is timed into it (the conv family measures 25-35 % longer per launch while the
            # step gets 5 % shorter).  The `sampled` steps before them are timed the same way WITH the overlap, for the record.
            serial = i >= steps - sampled
            timer.enabled = i >= steps - 2 * sampled
            timer.tag = "serial" if serial else "overlapped"
            Fn.set_stream_overlap(not serial and not args.serial)
            step_fn(warmup + i)
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        barrier()
        dt = time.time() - t0
        marks.append((t0 + dt, ev))
        timed.per_step = _per_step_stats(marks, sampled)
        # host CPU of the timed region: this process' CPU seconds per wall second, and what the container's CPU quota did to it
        # (cgroup v2 cpu.stat: a throttled period stops EVERY thread of the container, the launching thread included)
        cg1 = _cgroup_cpu_stat()
        timed.per_step["host_cpu"] = {"process_cpu_s_per_wall_s": round((time.process_time() - cpu0) / max(dt, 1e-9), 2),
                                      "torch_threads": torch.get_num_threads(),
                                      "cgroup_throttled_periods": cg1.get("nr_throttled", 0) - cg0.get("nr_throttled", 0),
                                      "cgroup_throttled_ms": round((cg1.get("throttled_usec", 0) - cg0.get("throttled_usec", 0)) / 1e3, 1),
                                      "cgroup_cpu_max": cg1.get("cpu_max")}
        timer.enabled = False
        Fn.set_stream_overlap(True)
        if manage_gc:
            gc.unfreeze()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t[0])
        return dt, sampled

    def run_train(steps, warmup):
        from u2seg_amd.config import get_cfg
        from u2seg_amd.data import make_synthetic_batch
        from u2seg_amd.engine import SimpleTrainer
        from u2seg_amd.modeling import build_model
        from u2seg_amd.solver import build_lr_scheduler, build_optimizer

        out = {}
        batch = args.batch or 16
        torch.manual_seed(1234)  # identical initial weights on every rank
        cfg = get_cfg()
        cfg.merge_from_file(os.path.join(CFG_DIR, "u2seg_R50_800.yaml"))
        cfg.merge_from_list(["MODEL.DEVICE", dev, "SOLVER.IMS_PER_BATCH", batch * world])
        model = build_model(cfg)
        model.train()
        opt = build_optimizer(cfg, model)
        trainer = SimpleTrainer(model, opt, build_lr_scheduler(cfg, opt))
        # a few cached synthetic batches resident in HBM (tools/benchmark.py:108-115 caches 100 batches the same way)
        nb = 2
        batches = [make_synthetic_batch(batch, start_index=(rank * nb + i) * batch, height=args.height, width=args.width,
                                        device=dev) for i in range(nb)]
        torch.manual_seed(1000 + rank)  # per-rank sampling randomness
        dt, sampled = timed(lambda i: trainer.run_step(batches[i % nb]), steps, warmup)
        total = trainer.check_finite()
        imgs_per_s = batch * world * steps / dt
        out.update({"metric": "training images/sec (whole node) u2seg_R50_800 @ 800x1333", "value": imgs_per_s, "unit": "img/s",
                    "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "dtype": "bf16",
                    "config": {"workload": "u2seg_R50_800.yaml bf16, batch %d per GPU, %dx%d synthetic COCO-panoptic batches, "
                                           "random init, SGD+per-param clip" % (batch, args.height, args.width),
                               "global_batch": batch * world, "parallelism": "dp%d" % world},
                    "final_total_loss": total, "per_step": timed.per_step})
        if rank == 0:
            out["roofline"] = conv_roofline(timer, sampled, steps, imgs_per_s / world)
            if args.per_layer:
                per_layer_report(timer, sampled)
        return out

    def run_infer(steps, warmup):
        from u2seg_amd.config import get_cfg
        from u2seg_amd.data import make_synthetic_batch
        from u2seg_amd.modeling import build_model

        out = {}
        batch = args.batch or 32
        torch.manual_seed(1234)
        cfg = get_cfg()
        cfg.merge_from_file(os.path.join(CFG_DIR, "u2seg_eval_800.yaml"))
        cfg.merge_from_list(["MODEL.DEVICE", dev])
        model = build_model(cfg)
        model.eval()
        data = [{k: v for k, v in x.items() if k != "instances"}
                for x in make_synthetic_batch(batch, start_index=rank * batch, height=args.height, width=args.width, device=dev)]
        res = {}

        def step(i):
            with torch.no_grad():
                res["out"] = model(data)

        dt, sampled = timed(step, steps, warmup)
        imgs_per_s = batch * world * steps / dt
        out.update({"metric": "panoptic inference images/sec (whole node) u2seg_eval_800 @ 800x1333", "value": imgs_per_s,
                    "unit": "img/s", "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak",
                    "dtype": "bf16", "steps": steps, "warmup": warmup,
                    "config": {"workload": "u2seg_eval_800.yaml bf16, batch %d per GPU, %dx%d synthetic images, random init, full "
                                           "post-processing (box NMS, pasted masks, semantic argmax, panoptic merge)"
                                           % (batch, args.height, args.width),
                               "global_batch": batch * world, "parallelism": "replicas%d" % world},
                    "instances_image0": len(res["out"][0]["instances"]), "segments_image0": len(res["out"][0]["panoptic_seg"][1]),
                    "per_step": timed.per_step})
        if rank == 0:
            out["roofline"] = conv_roofline(timer, sampled, steps)
            if args.per_layer:
                per_layer_report(timer, sampled)
        return out

    def run_kmeans(steps, warmup, data_kind, k=KMEANS_K):
        from u2seg_amd.cluster import kmeans as KM

        out = {}
        n_local = args.kmeans_n // world
        g = torch.Generator(device=dev).manual_seed(rank)
        gc = torch.Generator(device=dev).manual_seed(12345)  # the mixture centres / the initial centroids are the same on every rank
        if data_kind == "mixture":
            centers = torch.randn((k, KMEANS_D), generator=gc, device=dev) * 2
            x = centers[torch.randint(0, k, (n_local,), generator=g, device=dev)] + \
                0.5 * torch.randn((n_local, KMEANS_D), generator=g, device=dev)
            state = {"c": centers + 0.3 * torch.randn((k, KMEANS_D), generator=gc, device=dev)}
            what = "mixture of %d Gaussians, sigma 0.5" % k
        else:  # SURVEY 8(d) base case: unstructured data, the reference's init (random rows of x)
            x = torch.randn((n_local, KMEANS_D), generator=g, device=dev)
            state = {"c": x[torch.randperm(n_local, generator=g, device=dev)[:k]].clone()}
            what = "x = randn(N, %d), initial centroids = random rows" % KMEANS_D
            if world > 1:
                dist.broadcast(state["c"], 0)
        rechecks = []

        def step(i):
            lab = KM.assign(x, state["c"])
            state["c"], _ = KM.update_sharded(x, lab, k) if world > 1 else KM.update(x, lab, k)

        # once per run of kmeans(), not per iteration: the 16-bit shadow of x the first screening pass streams (timed on its own here;
        # the reference's niter is 100 - nn_utils.py:382 - so it adds a hundredth of this to an iteration)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        made = KM._shadow(x)
        e1.record()
        torch.cuda.synchronize()
        prepare_ms = e0.elapsed_time(e1) if made is not None else None
        dt, sampled = timed(step, steps, warmup)
        rechecks.append(KM.last_recheck_count(x.device))
        s_per_iter = dt / steps
        out.update({"metric": "k-means seconds per Lloyd iteration (u2seg_R50_%d Instance_Clustering: N = 1M x 768 DINO-sized "
                              "embeddings, K = %d)" % (k, k), "value": s_per_iter, "unit": "s/iter", "ms_per_step": s_per_iter * 1e3,
                    "higher_is_better": False, "scaling": "strong", "steps": steps, "warmup": warmup,
                    "dtype": "f32 (distances screened in split bf16 - the first pass over a 16-bit shadow of x made once per run, "
                             "one_time_shadow_prepare_ms - undecided points in exact fp32)",
                    "config": {"workload": "Lloyd iterations (assign + update) over %d x %d synthetic embeddings (%s), K = %d, rows "
                                           "sharded over the GPUs" % (n_local * world, KMEANS_D, what, k),
                               "parallelism": "rows%d" % world},
                    "finite_centroids": bool(torch.isfinite(state["c"]).all()), "per_step": timed.per_step,
                    "one_time_shadow_prepare_ms": prepare_ms})
        if rank == 0:
            ks = timer.summary()
            if "u2_kmeans_assign_shadow" in ks:      # the product path (cluster/kmeans.py assign): same E step, x's 16-bit shadow beside x
                ks["u2_kmeans_assign"] = ks.pop("u2_kmeans_assign_shadow")
            a = ks.get("u2_kmeans_assign")
            if a:
                # one Lloyd iteration needs ONE pass over X (N*D*4 bytes) once the distances run on bf16-class
                # MFMA, so the iteration is priced against HBM; the E step's algorithmic 2*N*K*D flop are reported beside it
                # (algorithmic, not the 3 bf16 piece products the screening kernel executes per product)
                alg_bytes = 4.0 * n_local * KMEANS_D
                # measured HBM bytes of one iteration: the committed PMC passes are of THIS configuration only (N = 1 M on one GPU,
                # K = 300, clustered data: E step over the 16-bit shadow 1.6 GB + M step 3.1 GB)
                km_traffic, km_note = None, "no PMC profile committed for this configuration"
                kp = os.path.join(ROOT, "profiles", "r06_km_pmc_traffic.json")
                if os.path.exists(kp) and world == 1 and n_local == 1000000 and k == KMEANS_K and data_kind == "mixture":
                    kj = json.load(open(kp))
                    km_traffic, km_note = kj["kmeans_iteration_hbm_bytes"], "HBM bytes per iteration from profiles/r06_km_pmc_traffic.json (%s)" % kj["note"]
                ach = alg_bytes / s_per_iter / 1e12
                out["roofline"] = {"kernel": "Lloyd iteration = u2_kmeans_assign_shadow (kmeans_coarse_kernel: |c|^2 - 2 x.c as hi.hi over "
                                             "the 16-bit shadow of x; kmeans_screen_kernel: hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16 "
                                             "for what it leaves undecided; exact-fp32 kmeans_assign_kernel for what that leaves) + "
                                             "u2_kmeans_update (label-bucketed segmented sums)",
                                   "bound": "hbm", "achieved": ach, "peak": PEAK_HBM_TBPS, "unit": "TB/s", "frac": ach / PEAK_HBM_TBPS,
                                   "traffic": km_traffic, "traffic_note": km_note, "algorithmic_bytes_per_iter": alg_bytes,
                                   "e_step_algorithmic_tflops": a["flops"] / (a["ms"] * 1e-3) / 1e12,
                                   "e_step_frac_of_bf16_mfma_peak": a["flops"] / (a["ms"] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                                   "rechecked_points_last_call": rechecks[-1],
                                   "avg_launch_ms": a["ms"] / a["launches"],
                                   "kernel_ms_per_iter": {k: v["ms"] / sampled for k, v in ks.items()},
                                   "update_GB_per_s": (ks["u2_kmeans_update"]["bytes"] / (ks["u2_kmeans_update"]["ms"] * 1e-3) / 1e9
                                                       if "u2_kmeans_update" in ks else None)}
        return out

    def release():
        import gc

        from u2seg_amd.cluster import kmeans as KM
        from u2seg_amd.layers import functional as Fn

        KM.release_shadow()   # the k-means workloads' 16-bit shadow of x (half of x)
        gc.collect()
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        Fn.release_library_scratch()  # the conv library's own scratch (not visible to torch's allocator)

    out = {"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "vs_baseline": None, "data": "synthetic"}
    if share_gpu:
        out["data"] = "synthetic; SMOKE TEST of the multi-rank code path: %d ranks share cuda:0 over gloo (U2_BENCH_SHARE_GPU=1) - not a measurement" % world
    if args.workload == "train":
        out.update(run_train(args.steps, args.warmup))
    elif args.workload == "infer":
        out.update(run_infer(args.steps, args.warmup))
    else:
        out.update(run_kmeans(args.steps, args.warmup, args.kmeans_data))
    extra = {}
    if args.workload == "train" and not args.no_extra:
        # BASELINE.json configs[3] and configs[4] in the same invocation, after the timed training region and
        # with the training state released; each entry is a full bench object (own timed region, roofline, cpu_baseline at N = 1).
        # At N > 1 every rank takes part: k-means shards the rows (the M step all-reduces K*D + K partial sums), inference runs
        # as independent replicas; the barriers of timed() keep the ranks together.
        release()
        # (kmeans_k800: the cluster count of u2seg_R50_800 - the screening kernels hold 320 centroids, so this is the path by blocks)
        for name, fn in (("kmeans", lambda: run_kmeans(30, 5, "mixture")), ("kmeans_randn", lambda: run_kmeans(30, 5, "randn")),
                         ("kmeans_k800", lambda: run_kmeans(30, 5, "mixture", 800)), ("infer", lambda: run_infer(20, 5))):
            try:
                extra[name] = dict({"n_gpus": world, "data": "synthetic"}, **fn())
            except Exception as e:  # an auxiliary workload must never take the headline line down with it
                extra[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            release()
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            kinds = [args.workload] + (["kmeans", "infer"] if extra else [])
            cb = cpu_baselines_parallel(kinds)
            out["cpu_baseline"] = cb[args.workload]
            for name in extra:
                if "error" not in extra[name]:
                    if name != "kmeans_k800":   # (the CPU sample is timed at K = 300)
                        extra[name]["cpu_baseline"] = cb["kmeans" if name.startswith("kmeans") else "infer"]
        if extra:
            out["extra_workloads"] = extra
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
