"""Captures the ROIAlign backward gather of one real training step and times it per level with ablations (U2_ROI_ABL)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import u2seg_amd.layers.functional as F

cap = {}
orig = F._roi_gather
def capture(shapes, scales, sets, device):
    if "args" not in cap:
        cap["args"] = (shapes, scales, [tuple(t.clone() if isinstance(t, torch.Tensor) else t for t in st[:6]) for st in sets], device)
    return orig(shapes, scales, sets, device)
F._roi_gather = capture
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-extra", "--no-cpu-baseline"]
import runpy
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
shapes, scales, sets, device = cap["args"]
print("shapes", shapes, "sets", [(tuple(s[0].shape), s[4], s[5]) for s in sets])
for si, st in enumerate(sets):
    seg = st[2].cpu().view(-1)
    cnt = (seg[1:] - seg[:-1]).view(shapes[0][0], len(shapes))
    print("set", si, "P", st[4], "rois per level (sum over images)", cnt.sum(0).tolist(), "max per (image, level)", cnt.max(0).values.tolist())
    r = st[0]
    w = (r[:, 3] - r[:, 1]); h = (r[:, 4] - r[:, 2])
    print("   box w mean %.1f h mean %.1f (image px)" % (float(w.mean()), float(h.mean())))

def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

tot = 0.0
for lv in range(len(shapes)):
    os.environ["U2_ROI_LEVEL"] = str(lv)
    t = timeit(lambda: orig(shapes, scales, sets, device))
    tot += t
    print("level %d  %8.1f us" % (lv, t))
print("sum %8.1f us" % tot)
