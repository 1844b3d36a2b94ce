"""Batch-level (padded) views of the per-image lists the reference passes between RPN and ROI heads.

The reference hands ``list[Instances]`` from the proposal generator to the ROI heads (proposal_generator/rpn.py:482-533 ->
roi_heads/roi_heads.py:220-302) and loops over images on the host.  Here the RPN keeps its result as padded device
tensors plus per-image counts that stay on the device; the ROI heads' training path consumes those directly (no
host synchronisation, a handful of batched kernels), and the ragged ``list[Instances]`` the reference API promises is
only materialised - with the one device->host copy it needs - when somebody actually indexes or iterates it."""
import collections.abc
import threading
import weakref

import numpy as np
import torch

from ..structures import Boxes, Instances


_const_cache = collections.OrderedDict()  # key -> [tensor, upload event or None, stream id of the upload]


def _cache_get(key, dev):
    """A cached constant for whichever stream asks: the stream that uploaded it is ordered by itself, any other stream waits
    for the upload's event until that has completed once (then the entry is plain read-only memory).  No host synchronisation."""
    ent = _const_cache.get(key)
    if ent is None:
        return None
    _const_cache.move_to_end(key)
    if ent[1] is not None:
        if ent[1].query():
            ent[1] = None
        else:
            cur = torch.cuda.current_stream(dev)
            if cur.cuda_stream != ent[2]:
                cur.wait_event(ent[1])
                ent[0].record_stream(cur)
    return ent[0]


def _cache_put(key, host, dev):
    if dev.type == "cuda":
        out, ev = _PinnedRing.get(dev).upload(host, want_event=True)
        _const_cache[key] = [out, ev, torch.cuda.current_stream(dev).cuda_stream]
    else:
        out = host.to(dev)
        _const_cache[key] = [out, None, 0]
    if len(_const_cache) > 512:
        _const_cache.popitem(last=False)
    return out


def device_constant(values, dtype, device):
    """Small host-known tensor on the device.  A pageable host->device copy makes the host wait for everything already
    queued on the stream, i.e. it is a hidden synchronisation; values that repeat from step to step (image sizes, box
    counts, ...) are therefore cached on the device (read-only!), uploaded once from the pinned staging ring; other streams
    are ordered behind that upload by its event (round 5: was pin_memory() + a stream synchronisation per new value)."""
    dev = torch.device(device)
    key = (repr(values), dtype, str(dev))
    hit = _cache_get(key, dev)
    if hit is not None:
        return hit
    return _cache_put(key, torch.tensor(values, dtype=dtype), dev)


class _PinnedRing:
    """Pinned staging slots allocated ONCE per device: `tensor.pin_memory()` per upload goes to hipHostMalloc whenever
    the caching host allocator has no block whose last copy has retired - milliseconds each, and it synchronises (measured:
    batch-32 inference inside the default bench.py run, after the training workload, 47 -> 68 ms per batch).  A slot is
    reused only after the event recorded behind its last copy has completed.  64 slots of 64 KB for the usual vectors of
    counts / offsets plus 4 slots of 4 MB for the rare long one (per-ROI image indices); anything larger is copied through a
    one-off pinned block.  Slot selection is under a lock: uploads may come from the autograd thread as well."""

    RINGS = ((64, 65536), (4, 4 << 20))
    _rings = {}
    _guard = threading.Lock()

    def __init__(self, dev):
        self.bufs = [torch.empty((n, size), dtype=torch.uint8).pin_memory() for n, size in self.RINGS]
        self.events = [[None] * n for n, _ in self.RINGS]
        self.next = [0] * len(self.RINGS)
        self.dev = dev
        self.lock = threading.Lock()

    @classmethod
    def get(cls, dev):
        key = str(dev)
        ring = cls._rings.get(key)
        if ring is None:
            with cls._guard:
                ring = cls._rings.get(key)
                if ring is None:
                    ring = cls._rings[key] = cls(dev)
        return ring

    def upload(self, host, want_event=False):
        nbytes = host.numel() * host.element_size()
        out = torch.empty(host.shape, dtype=host.dtype, device=self.dev)
        if nbytes == 0:
            return (out, None) if want_event else out
        r = next((k for k, (_, size) in enumerate(self.RINGS) if nbytes <= size), None)
        cur = torch.cuda.current_stream(self.dev)
        if r is None:  # larger than any slot: a one-off pinned block (the caching host allocator keeps it alive until the copy retires)
            out.copy_(host.pin_memory(), non_blocking=True)
            ev = None
            if want_event:
                ev = torch.cuda.Event()
                ev.record(cur)
            return (out, ev) if want_event else out
        with self.lock:
            i = self.next[r]
            self.next[r] = (i + 1) % self.RINGS[r][0]
            ev = self.events[r][i]
            if ev is not None and not ev.query():
                ev.synchronize()
            stage = self.bufs[r][i, :nbytes].view(host.dtype).view(host.shape)
            stage.copy_(host)
            out.copy_(stage, non_blocking=True)
            ev = self.events[r][i] = torch.cuda.Event()  # a fresh event: the previous one may be held by a cache entry
            ev.record(cur)
        return (out, ev) if want_event else out

    def warm(self):
        """Touch every pinned page and create the events before a timed region (first touch of pinned memory faults it in)."""
        for b in self.bufs:
            b.zero_()


def device_upload(values, dtype, device):
    """Host-known values that CHANGE from batch to batch (box counts, offsets): one non-blocking copy from a pinned staging slot
    on the current stream - no cache entry (they would only evict the constants that do repeat), no allocation and no
    synchronisation."""
    host = torch.tensor(values, dtype=dtype)
    dev = torch.device(device)
    if dev.type != "cuda":
        return host.to(dev)
    return _PinnedRing.get(dev).upload(host)


class LazyProposals(collections.abc.Sequence):
    """list[Instances] (fields proposal_boxes, objectness_logits) backed by padded tensors.

    boxes [B, P, 4], logits [B, P], counts int32 [B] (device); rows >= counts[i] are padding.  ``finite`` is a device
    bool that is False if the RPN produced Inf/NaN; it is checked whenever counts are brought to the host."""

    def __init__(self, image_sizes, boxes, logits, counts, finite=None, training=False):
        self.image_sizes = list(image_sizes)
        self.boxes, self.logits, self.counts = boxes, logits, counts
        self.finite, self.training = finite, training
        self._items = None

    def host_counts(self):
        if self.finite is not None:
            vals = torch.cat([self.counts.to(torch.int64), self.finite.reshape(1).to(torch.int64)]).tolist()
            check_finite(bool(vals[-1]), self.training)
            self.finite = None
            return vals[:-1]
        return self.counts.tolist()

    def _materialize(self):
        if self._items is None:
            nk = self.host_counts()
            items = []
            for i, size in enumerate(self.image_sizes):
                res = Instances(size)
                res.proposal_boxes = Boxes(self.boxes[i, : nk[i]])
                res.objectness_logits = self.logits[i, : nk[i]]
                items.append(res)
            self._items = items
        return self._items

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, i):
        return self._materialize()[i]

    def __iter__(self):
        return iter(self._materialize())


def check_finite(ok, training):
    if training and not ok:
        raise FloatingPointError("Predicted boxes or scores contain Inf/NaN. Training has diverged.")


class PaddedTargets:
    """gt boxes [B, G, 4] fp32 / classes [B, G] int64 (zero padded, G >= 1) and counts int32 [B] of a list of gt
    Instances; built once per step and shared by the RPN and the ROI heads."""

    _cache = None  # (weakref to the first Instances, ids, PaddedTargets)

    def __init__(self, gt_instances, device):
        self.num = [len(x) for x in gt_instances]
        b, g = len(gt_instances), max(1, max(self.num) if self.num else 1)
        self.boxes = torch.zeros((b, g, 4), dtype=torch.float32, device=device)
        has_cls = all(x.has("gt_classes") for x in gt_instances)
        self.classes = torch.zeros((b, g), dtype=torch.int64, device=device) if has_cls else None
        some = [inst for i, inst in enumerate(gt_instances) if self.num[i]]
        if some:
            # one concatenation + one indexed copy per field instead of a slice assignment per image (32 small launches per step)
            rows = device_constant([i * g + j for i in range(b) for j in range(self.num[i])], torch.int64, device)
            self.boxes.view(b * g, 4).index_copy_(0, rows, torch.cat([x.gt_boxes.tensor.to(device=device, dtype=torch.float32)
                                                                      for x in some]))
            if has_cls:
                self.classes.view(b * g).index_copy_(0, rows, torch.cat([x.gt_classes.to(device=device, dtype=torch.int64)
                                                                        for x in some]))
        self.counts = device_upload(self.num, torch.int32, device)

    @classmethod
    def of(cls, gt_instances, device):
        c = cls._cache
        ids = tuple(id(x) for x in gt_instances)
        if c is not None and c[0]() is gt_instances[0] and c[1] == ids and c[2].boxes.device == torch.device(device):
            return c[2]
        out = cls(gt_instances, device)
        cls._cache = (weakref.ref(gt_instances[0]), ids, out)
        return out


class BatchList(list):
    """list[Instances] whose images all hold the same number of boxes, plus the stacked tensors the per-image fields
    are views of: ``boxes`` [B, S, 4], ``gt_classes`` [B, S] and ``gt_boxes`` [B, S, 4] (training; optionally ``logits`` / ``match``).  Heads use the
    stacked form (no per-image concatenation); everything else sees a plain list."""

    boxes = gt_classes = gt_boxes = None
    logits = match = None   # objectness logits [B, S] and matched gt row [B, S] of the sampler (the row index its lazy gt_masks hold)

    @property
    def stacked(self):
        return self.boxes is not None


def proposals_from_list(proposals, training=False):
    """list[Instances] -> LazyProposals (padding copies: one small kernel per image)."""
    if isinstance(proposals, LazyProposals):
        return proposals
    n = [len(p) for p in proposals]
    dev = proposals[0].proposal_boxes.tensor.device
    pmax = max(max(n), 1)
    boxes = torch.zeros((len(n), pmax, 4), dtype=torch.float32, device=dev)
    logits = torch.zeros((len(n), pmax), dtype=torch.float32, device=dev)
    for i, p in enumerate(proposals):
        if n[i]:
            boxes[i, : n[i]] = p.proposal_boxes.tensor
            if p.has("objectness_logits"):
                logits[i, : n[i]] = p.objectness_logits
    out = LazyProposals([p.image_size for p in proposals], boxes, logits, device_upload(n, torch.int32, dev),
                        None, training)
    out._items = list(proposals)
    return out


def image_index(sizes, device):
    """float32 image index per ROI for per-image box counts `sizes` (host ints): built on the host, no device sync."""
    dev = torch.device(device)
    key = ("image_index", tuple(sizes), str(dev))
    hit = _cache_get(key, dev)
    if hit is None:
        # numpy on purpose: ATen's CPU repeat_interleave is a parallel region with a grain of 1 - it wakes the whole OpenMP pool for
        # 16 elements, and the woken threads spin for milliseconds (u2seg_amd/utils/env.py)
        idx = torch.from_numpy(np.repeat(np.arange(len(sizes), dtype=np.float32), np.asarray(sizes, dtype=np.int64)))
        hit = _cache_put(key, idx, dev)
    return hit
