"""Batch-level (padded) views of the per-image lists the reference passes between RPN and ROI heads.

The reference hands ``list[Instances]`` from the proposal generator to the ROI heads (proposal_generator/rpn.py:482-533 ->
roi_heads/roi_heads.py:220-302) and loops over images on the host.  Here the RPN keeps its result as padded device
tensors plus per-image counts that stay on the device; the ROI heads' training path consumes those directly (no
host synchronisation, a handful of batched kernels), and the ragged ``list[Instances]`` the reference API promises is
only materialised - with the one device->host copy it needs - when somebody actually indexes or iterates it."""
import collections.abc
import weakref

import torch

from ..structures import Boxes, Instances


_const_cache = collections.OrderedDict()


def device_constant(values, dtype, device):
    """Small host-known tensor on the device.  A pageable host->device copy makes the host wait for everything already
    queued on the stream, i.e. it is a hidden synchronisation; values that repeat from step to step (image sizes, box
    counts, ...) are therefore cached on the device (read-only!) and the rest goes through pinned memory."""
    key = (repr(values), dtype, str(device))
    hit = _const_cache.get(key)
    if hit is not None:
        _const_cache.move_to_end(key)
        return hit
    host = torch.tensor(values, dtype=dtype)
    dev = torch.device(device)
    if dev.type == "cuda":
        out = host.pin_memory().to(dev, non_blocking=True)
        # a cached constant is handed to whichever stream asks next: finish the copy once, here (first use of a value only)
        torch.cuda.current_stream(dev).synchronize()
    else:
        out = host.to(dev)
    _const_cache[key] = out
    if len(_const_cache) > 512:
        _const_cache.popitem(last=False)
    return out


class _PinnedRing:
    """A few pinned staging slots allocated ONCE per device: `tensor.pin_memory()` per upload goes to hipHostMalloc whenever
    the caching host allocator has no block whose last copy has retired - milliseconds each, and it synchronises (measured:
    batch-32 inference inside the default bench.py run, after the training workload, 47 -> 68 ms per batch).  A slot is
    reused only after the event recorded behind its last copy has completed."""

    SLOTS, SLOT_BYTES = 64, 65536
    _rings = {}

    def __init__(self, dev):
        self.buf = torch.empty((self.SLOTS, self.SLOT_BYTES), dtype=torch.uint8).pin_memory()
        self.events = [None] * self.SLOTS
        self.next = 0
        self.dev = dev

    @classmethod
    def get(cls, dev):
        key = str(dev)
        ring = cls._rings.get(key)
        if ring is None:
            ring = cls._rings[key] = cls(dev)
        return ring

    def upload(self, host):
        nbytes = host.numel() * host.element_size()
        i = self.next
        self.next = (i + 1) % self.SLOTS
        ev = self.events[i]
        if ev is not None and not ev.query():
            ev.synchronize()
        stage = self.buf[i, :nbytes].view(host.dtype).view(host.shape)
        stage.copy_(host)
        out = torch.empty(host.shape, dtype=host.dtype, device=self.dev)
        out.copy_(stage, non_blocking=True)
        if ev is None:
            ev = self.events[i] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.dev))
        return out


def device_upload(values, dtype, device):
    """Host-known values that CHANGE from batch to batch (box counts, offsets): one non-blocking copy from a pinned staging slot
    on the current stream - no cache entry (they would only evict the constants that do repeat) and no synchronisation."""
    host = torch.tensor(values, dtype=dtype)
    dev = torch.device(device)
    if dev.type != "cuda":
        return host.to(dev)
    if host.numel() == 0 or host.numel() * host.element_size() > _PinnedRing.SLOT_BYTES:
        return host.pin_memory().to(dev, non_blocking=True)
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
        for i, inst in enumerate(gt_instances):
            if self.num[i]:
                self.boxes[i, : self.num[i]] = inst.gt_boxes.tensor
                if has_cls:
                    self.classes[i, : self.num[i]] = inst.gt_classes
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
    key = ("image_index", tuple(sizes), str(device))
    hit = _const_cache.get(key)
    if hit is None:
        idx = torch.repeat_interleave(torch.arange(len(sizes), dtype=torch.float32), torch.tensor(sizes, dtype=torch.int64))
        dev = torch.device(device)
        hit = _const_cache[key] = idx.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else idx.to(dev)
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()  # cached for any stream: finish the upload once
        if len(_const_cache) > 512:
            _const_cache.popitem(last=False)
    return hit
