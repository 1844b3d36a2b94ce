"""Samplers, batching and loader construction (detectron2/data/build.py:46-75, 217-360, 437-660;
samplers/distributed_sampler.py:16-95, 236-277; common.py:241-300), plus the host -> HBM hand-over.

One process per GPU: every rank walks the same seeded infinite permutation stream and keeps every world_size-th index
(no exchange between ranks on the data path).  Decoding and resizing run in DataLoader worker processes; `DevicePrefetcher`
moves the mapped batches through pinned buffers on a side stream so the copy of batch i + 1 overlaps the step on batch i."""
import contextlib
import itertools
import logging
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.utils.data as torchdata

from .catalog import DatasetCatalog
from .dataset_mapper import DatasetMapper

logger = logging.getLogger(__name__)


def _rank_world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shared_random_seed():
    """A seed every rank agrees on (utils/comm.py:221-233): rank 0 draws it from numpy's global stream (so a seeded
    run stays reproducible), the others receive it."""
    seed = [int(np.random.randint(2 ** 31))]
    if _rank_world()[1] > 1:
        dist.broadcast_object_list(seed, src=0)
    return seed[0]


class TrainingSampler(torchdata.Sampler):
    """Infinite stream shuffle(range(size)) + shuffle(range(size)) + ... from one seeded generator; rank r takes
    elements r, r + world, ... (distributed_sampler.py:16-77)."""

    def __init__(self, size, shuffle=True, seed=None):
        if not isinstance(size, int):
            raise TypeError("TrainingSampler(size=) expects an int. Got type {}.".format(type(size)))
        if size <= 0:
            raise ValueError("TrainingSampler(size=) expects a positive int. Got {}.".format(size))
        self._size, self._shuffle = size, shuffle
        self._seed = int(shared_random_seed() if seed is None else seed)
        self._rank, self._world_size = _rank_world()

    def __iter__(self):
        yield from itertools.islice(self._infinite_indices(), self._rank, None, self._world_size)

    def _infinite_indices(self):
        g = torch.Generator()
        g.manual_seed(self._seed)
        while True:
            if self._shuffle:
                yield from torch.randperm(self._size, generator=g).tolist()
            else:
                yield from range(self._size)


class InferenceSampler(torchdata.Sampler):
    """Contiguous shards covering every sample exactly once; the first size % world ranks get one more
    (distributed_sampler.py:236-277)."""

    def __init__(self, size):
        assert size > 0
        self._size = size
        rank, world = _rank_world()
        self._local_indices = self._get_local_indices(size, world, rank)

    @staticmethod
    def _get_local_indices(total_size, world_size, rank):
        shard, left = divmod(total_size, world_size)
        sizes = [shard + int(r < left) for r in range(world_size)]
        begin = sum(sizes[:rank])
        return range(begin, min(begin + sizes[rank], total_size))

    def __iter__(self):
        yield from self._local_indices

    def __len__(self):
        return len(self._local_indices)


class MapDataset(torchdata.Dataset):
    """dataset[i] -> map_func(dataset[i]); a sample the mapper rejects (returns None) is replaced by another one drawn
    from a per-dataset seeded stream, as in common.py:60-115."""

    def __init__(self, dataset, map_func):
        self._dataset, self._map_func = dataset, map_func
        self._rng = np.random.RandomState(42)
        self._fallback = set(range(len(dataset)))

    def __len__(self):
        return len(self._dataset)

    def __getitem__(self, idx):
        retry, cur = 0, int(idx)
        while True:
            data = self._map_func(self._dataset[cur])
            if data is not None:
                self._fallback.add(cur)
                return data
            retry += 1
            self._fallback.discard(cur)
            cur = int(self._rng.choice(sorted(self._fallback)))
            if retry >= 3:
                logger.warning("Failed to apply `_map_func` for idx: {}, retry count: {}".format(idx, retry))


class _SampledStream(torchdata.IterableDataset):
    """A map-style dataset walked in sampler order.  With several DataLoader workers, worker k of n takes the chunks
    k, k + n, ... of `chunk` consecutive indices, so that each worker produces whole per-GPU batches
    (common.py:_shard_iterator_dataloader_worker, 22-50)."""

    def __init__(self, dataset, sampler, chunk=1):
        self.dataset, self.sampler, self.chunk = dataset, sampler, chunk

    def __iter__(self):
        info = torchdata.get_worker_info()
        stream = iter(self.sampler)
        if info is not None and info.num_workers > 1:
            n, k, c = info.num_workers, info.id, self.chunk
            stream = (i for pos, i in enumerate(stream) if (pos // c) % n == k)
        for idx in stream:
            yield self.dataset[idx]


class AspectRatioGroupedDataset(torchdata.IterableDataset):
    """Landscape and portrait images fill separate buckets; a bucket is emitted when it holds batch_size samples
    (common.py:241-283) - less padding per batch."""

    def __init__(self, dataset, batch_size):
        self.dataset, self.batch_size = dataset, batch_size
        self._buckets = [[], []]

    def __iter__(self):
        for d in self.dataset:
            bucket = self._buckets[0 if d["width"] > d["height"] else 1]
            bucket.append(d)
            if len(bucket) == self.batch_size:
                data = bucket[:]
                del bucket[:]
                yield data


def keep_big_buffers_on_heap():
    """glibc hands allocations above 128 KB to mmap and returns them to the kernel on free, so every mapped sample
    (tens of MB of masks and resized images) page-faults its buffers in again; raising the mmap / trim thresholds lets a
    worker recycle them (mapper latency 23.8 -> 18.5 ms per image in the build container).  Best effort, Linux only."""
    try:
        import ctypes

        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 1 << 30)  # M_MMAP_THRESHOLD
        libc.mallopt(-1, 1 << 30)  # M_TRIM_THRESHOLD
    except Exception:
        pass


def worker_init_reset_seed(worker_id):
    """Every worker process gets its own numpy / torch / python RNG stream (build.py:655-660, utils/env.py:22-38)."""
    import random

    keep_big_buffers_on_heap()

    seed = (torch.initial_seed() + worker_id) % 2 ** 31
    np.random.seed(seed)
    torch.manual_seed(seed)
    random.seed(seed)


def trivial_batch_collator(batch):
    return batch


def _first(batch):
    return batch[0]


def filter_images_with_only_crowd_annotations(dataset_dicts):
    return [x for x in dataset_dicts if any(ann.get("iscrowd", 0) == 0 for ann in x["annotations"])]


def get_detection_dataset_dicts(names, filter_empty=True):
    """Concatenated dataset dicts of the named datasets; images without a non-crowd annotation are dropped for training
    (build.py:217-291)."""
    if isinstance(names, str):
        names = [names]
    assert len(names), names
    per_dataset = [DatasetCatalog.get(n) for n in names]
    for n, dicts in zip(names, per_dataset):
        assert len(dicts), "Dataset '{}' is empty!".format(n)
    dataset_dicts = list(itertools.chain.from_iterable(per_dataset))
    if filter_empty and "annotations" in dataset_dicts[0]:
        dataset_dicts = filter_images_with_only_crowd_annotations(dataset_dicts)
    assert len(dataset_dicts), "No valid data found in {}.".format(",".join(names))
    return dataset_dicts


class BatchIndexStream:
    """Per-GPU batches of dataset indices in sampler order; with `landscape` (one bool per dataset index) the landscape and
    portrait images fill separate buckets exactly like AspectRatioGroupedDataset does after mapping - the orientation
    of an image is known from its dataset dict, so the grouping can happen before any pixel is read and a worker can map a
    whole batch."""

    def __init__(self, sampler, batch_size, landscape=None):
        self.sampler, self.batch_size, self.landscape = sampler, batch_size, landscape
        self._buckets = [[], []]

    def __iter__(self):
        for idx in self.sampler:
            bucket = self._buckets[0 if (self.landscape is None or self.landscape[idx]) else 1]
            bucket.append(idx)
            if len(bucket) == self.batch_size:
                batch = bucket[:]
                del bucket[:]
                yield batch


def build_batch_data_loader(dataset, sampler, total_batch_size, *, aspect_ratio_grouping=False, num_workers=0,
                            collate_fn=None, slot_megabytes=None, **kwargs):
    """Iterable of lists of mapped dicts, total_batch_size / world_size per list (build.py:294-359).

    num_workers == 0: the reference's sample-level structure.  With workers, and when every dataset dict carries its width
    and height, batches are formed on the index stream and travel through shared-memory slots (data/slots.py)."""
    world = _rank_world()[1]
    assert total_batch_size > 0 and total_batch_size % world == 0, \
        "Total batch size ({}) must be divisible by the number of gpus ({}).".format(total_batch_size, world)
    batch_size = total_batch_size // world
    base = getattr(dataset, "_dataset", None)
    sized = base is not None and all("width" in d and "height" in d for d in base)
    if num_workers > 0 and collate_fn is None and (sized or not aspect_ratio_grouping):
        from .slots import BatchPacker, HostUnpacked, SlotRing

        landscape = [d["width"] > d["height"] for d in base] if aspect_ratio_grouping else None
        prefetch = kwargs.pop("prefetch_factor", 2)
        if slot_megabytes is None:
            # ~10 MB per image covers a 1024 x 1333 sample (4.1 MB pixels, 1.4 MB label bytes, masks at 0.17 MB each);
            # a batch that still does not fit travels the ordinary way (BatchPacker returns the list)
            slot_megabytes = min(512, max(16, 10 * batch_size))
        ring = SlotRing(num_workers, slot_megabytes << 20, prefetch + 2)
        loader = torchdata.DataLoader(dataset, batch_sampler=BatchIndexStream(sampler, batch_size, landscape),
                                      num_workers=num_workers, collate_fn=BatchPacker(ring), prefetch_factor=prefetch,
                                      worker_init_fn=worker_init_reset_seed, **kwargs)
        return HostUnpacked(loader, ring)
    stream = _SampledStream(dataset, sampler, chunk=batch_size)
    if aspect_ratio_grouping:
        loader = torchdata.DataLoader(stream, num_workers=num_workers, collate_fn=_first, batch_size=1,
                                      worker_init_fn=worker_init_reset_seed, **kwargs)
        grouped = AspectRatioGroupedDataset(loader, batch_size)
        return grouped if collate_fn is None else map(collate_fn, grouped)
    return torchdata.DataLoader(stream, batch_size=batch_size, drop_last=True, num_workers=num_workers,
                                collate_fn=trivial_batch_collator if collate_fn is None else collate_fn,
                                worker_init_fn=worker_init_reset_seed, **kwargs)


def build_detection_train_loader(cfg, mapper=None, *, dataset=None, sampler=None, seed=None):
    """build.py:437-551 for the sampler the U2Seg configs name (TrainingSampler)."""
    if dataset is None:
        dataset = get_detection_dataset_dicts(cfg.DATASETS.TRAIN, filter_empty=cfg.DATALOADER.FILTER_EMPTY_ANNOTATIONS)
    if mapper is None:
        mapper = DatasetMapper(cfg, True)
    if sampler is None:
        assert cfg.DATALOADER.SAMPLER_TRAIN == "TrainingSampler", cfg.DATALOADER.SAMPLER_TRAIN
        sampler = TrainingSampler(len(dataset), seed=seed)
    return build_batch_data_loader(MapDataset(dataset, mapper), sampler, cfg.SOLVER.IMS_PER_BATCH,
                                   aspect_ratio_grouping=cfg.DATALOADER.ASPECT_RATIO_GROUPING,
                                   num_workers=cfg.DATALOADER.NUM_WORKERS)


def build_detection_test_loader(cfg, dataset_name, mapper=None, batch_size=1):
    """build.py:554-645: every sample exactly once, in order, sharded contiguously over the ranks."""
    dataset = get_detection_dataset_dicts(dataset_name, filter_empty=False)
    if mapper is None:
        mapper = DatasetMapper(cfg, False)
    mapped = MapDataset(dataset, mapper)
    return torchdata.DataLoader(mapped, batch_size=batch_size, sampler=InferenceSampler(len(dataset)), drop_last=False,
                                num_workers=cfg.DATALOADER.NUM_WORKERS, collate_fn=trivial_batch_collator)


class _StagingArena:
    """One reusable block of (pinned) host memory: the tensors of a batch are packed into it back to back and copied to
    the device from there.  Pinning is paid once (hipHostMalloc costs milliseconds per call, a mapped batch is ~100 MB),
    not per tensor per step."""

    ALIGN = 256

    def __init__(self, pinned):
        self.pinned, self.buf, self.used, self.done = pinned, None, 0, None

    def reset(self, nbytes):
        if self.done is not None:
            self.done.synchronize()  # the device copies that read this block have finished
            self.done = None
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(int(nbytes * 1.25) + self.ALIGN, dtype=torch.uint8, pin_memory=self.pinned)
        self.used = 0

    def put(self, t):
        """Copy of `t` inside the block (same dtype and shape).  The copy is a plain single-threaded numpy memcpy: a
        torch copy_ fans a few MB out over every host core, which on a 128-thread box costs more than the copy."""
        n = t.numel() * t.element_size()
        start = self.used
        self.used = (start + n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        dst = self.buf[start:start + n].view(t.dtype).view(t.shape)
        np.copyto(dst.numpy(), t.numpy(), casting="no")
        return dst

    @classmethod
    def footprint(cls, tensors):
        return sum((t.numel() * t.element_size() + cls.ALIGN - 1) // cls.ALIGN * cls.ALIGN for t in tensors)


class DevicePrefetcher:
    """Wraps an iterable of batches (lists of mapped dicts) and yields them with every tensor resident on `device`.

    The next batch is packed into one of three reusable pinned staging blocks and copied on a dedicated HIP stream while
    the caller still computes on the current one; before a batch is handed out the consumer stream waits on the copy's
    event, and the tensors are recorded on the consumer stream so the caching allocator does not recycle them early.
    Label maps travel as bytes (the mapper's int64 has 8x the volume for values 0..255) and are widened on the device."""

    def __init__(self, loader, device, _pinned=True):
        self.ring = getattr(loader, "ring", None)  # HostUnpacked: take the packed stream and unpack on the device instead
        self.loader, self.device = getattr(loader, "packed", loader), torch.device(device)
        self.on_gpu = self.device.type == "cuda"
        assert self.on_gpu or not _pinned, "DevicePrefetcher moves batches into HBM; it needs a GPU device"
        self.stream = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.arenas = [_StagingArena(_pinned) for _ in range(3)]
        self.turn = 0
        self.seconds_waiting_for_loader = self.seconds_staging = 0.0  # where the host time of this iterator went

    @staticmethod
    def _tensors_of(v):
        if isinstance(v, torch.Tensor):
            return [v]
        if hasattr(v, "get_fields"):
            return [getattr(f, "tensor", f) for f in v.get_fields().values()]
        return []

    @staticmethod
    def _narrow(t):
        """Host-side representation that is copied: int64 label maps as uint8 when their values allow it (numpy, single
        thread, for the reason given at _StagingArena.put)."""
        if t.dtype == torch.int64 and t.dim() == 2 and t.numel() > 4096:
            a = t.numpy()
            if 0 <= int(a.min()) and int(a.max()) <= 255:
                return torch.from_numpy(a.astype(np.uint8)), torch.int64
        return t, None

    def _move(self, arena, t):
        host, widen = self._narrow(t)
        staged = arena.put(host)
        dev = staged.to(self.device, non_blocking=True) if self.on_gpu else staged.clone()  # (CPU: test configuration)
        return dev.to(widen) if widen is not None else dev

    def _stage_packed(self, packed):
        """A batch that arrived in a shared-memory slot: one memcpy into the pinned block, one host-to-device copy, the
        tensors rebuilt on the device as views of that block (label maps and bit-packed masks widened there)."""
        from .slots import unpack

        arena = self.arenas[self.turn]
        self.turn = (self.turn + 1) % len(self.arenas)
        arena.reset(self.ring.slot_bytes)  # full slot size from the start: the pinned block is allocated exactly once
        np.copyto(arena.buf[: packed.nbytes].numpy(), self.ring.view(packed.slot)[: packed.nbytes])
        ctx = torch.cuda.stream(self.stream) if self.on_gpu else contextlib.nullcontext()
        with ctx:
            staged = arena.buf[: packed.nbytes]
            block = staged.to(self.device, non_blocking=True) if self.on_gpu else staged.clone()
            moved = unpack(block, packed.samples)
            done = None
            if self.on_gpu:
                done = torch.cuda.Event()
                done.record(self.stream)
                arena.done = done  # (the consumer-stream bookkeeping of the device tensors happens in __iter__)
        return moved, done

    def _stage(self, batch):
        if not isinstance(batch, list):
            return self._stage_packed(batch)
        arena = self.arenas[self.turn]
        self.turn = (self.turn + 1) % len(self.arenas)
        arena.reset(_StagingArena.footprint([t for d in batch for v in d.values() for t in self._tensors_of(v)]))
        ctx = torch.cuda.stream(self.stream) if self.on_gpu else contextlib.nullcontext()
        with ctx:
            moved = []
            for d in batch:
                out = {}
                for k, v in d.items():
                    if isinstance(v, torch.Tensor):
                        out[k] = self._move(arena, v)
                    elif hasattr(v, "get_fields"):  # Instances: field by field, wrappers (Boxes, BitMasks) rebuilt
                        inst = type(v)(v.image_size)
                        for name, field in v.get_fields().items():
                            t = self._move(arena, getattr(field, "tensor", field))
                            inst.set(name, type(field)(t) if hasattr(field, "tensor") else t)
                        out[k] = inst
                    else:
                        out[k] = v
                moved.append(out)
            done = None
            if self.on_gpu:
                done = torch.cuda.Event()
                done.record(self.stream)
                arena.done = done
        return moved, done

    def _record(self, batch):
        cur = torch.cuda.current_stream(self.device)
        for d in batch:
            for v in d.values():
                for t in self._tensors_of(v):
                    t.record_stream(cur)

    def _next_staged(self, it):
        t0 = time.perf_counter()
        try:
            batch = next(it)
        except StopIteration:
            return None
        t1 = time.perf_counter()
        staged = self._stage(batch)
        self.seconds_waiting_for_loader += t1 - t0
        self.seconds_staging += time.perf_counter() - t1
        return staged

    def __iter__(self):
        it = iter(self.loader)
        pending = self._next_staged(it)
        while pending is not None:
            batch, done = pending
            pending = self._next_staged(it)
            if self.on_gpu:
                torch.cuda.current_stream(self.device).wait_event(done)
                self._record(batch)
            yield batch
