"""Times u2_roi_align_fwd on the ROI sets of one real training step (captured) - box pooler (P = 7) and mask pooler (P = 14)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from u2seg_amd import _hip
import u2seg_amd.layers.functional as F

calls = []
orig = _hip.call
def spy(name, *args):
    if name == "u2_roi_align_fwd" and len(calls) < 4:
        calls.append(tuple(a.clone() if isinstance(a, torch.Tensor) and a.numel() < (1 << 24) else a for a in args))
    return orig(name, *args)
_hip.call = spy
F._hip.call = spy
sys.argv = ["bench.py", "--steps", "1", "--warmup", "1", "--no-extra", "--no-cpu-baseline"]
import runpy
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
_hip.call = orig
F._hip.call = orig
import ctypes
for unr in ('1', '4', '1', '4'):
    os.environ['U2_ROI_FWD_UNR'] = unr
    print('pixel loads in flight per item:', unr)
    tot = 0.0
    for args in calls:
        ptrs, hs, ws, sc, nl, rois, levels, out, r, c, ph, pw = args
        feats = [torch.randn((16, hs[i], ws[i], c), device="cuda").bfloat16() for i in range(nl)]
        p2 = (ctypes.c_void_p * nl)(*[f.data_ptr() for f in feats])
        out = torch.empty((r, ph, pw, c), dtype=torch.bfloat16, device="cuda")
        fn = lambda: orig("u2_roi_align_fwd", p2, hs, ws, sc, nl, rois, levels, out, r, c, ph, pw)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 20 * 1e3
        tot += t
        print("ROIs %5d  P %2d  %8.1f us" % (r, ph, t))
    print("sum %8.1f us" % tot)
