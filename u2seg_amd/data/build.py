"""Samplers, batching and loader construction (detectron2/data/build.py:46-75, 217-360, 437-660;
samplers/distributed_sampler.py:16-95, 236-277; common.py:241-300), plus the host -> HBM hand-over.

One process per GPU: every rank walks the same seeded infinite permutation stream and keeps every world_size-th index
(no exchange between ranks on the data path).  Decoding and resizing run in DataLoader worker processes; `DevicePrefetcher`
moves the mapped batches through pinned buffers on a side stream so the copy of batch i + 1 overlaps the step on batch i."""
import itertools
import logging

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


def build_batch_data_loader(dataset, sampler, total_batch_size, *, aspect_ratio_grouping=False, num_workers=0,
                            collate_fn=None, **kwargs):
    """Iterable of lists of mapped dicts, total_batch_size / world_size per list (build.py:294-359)."""
    world = _rank_world()[1]
    assert total_batch_size > 0 and total_batch_size % world == 0, \
        "Total batch size ({}) must be divisible by the number of gpus ({}).".format(total_batch_size, world)
    batch_size = total_batch_size // world
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


class DevicePrefetcher:
    """Wraps an iterable of batches (lists of mapped dicts) and yields them with every tensor resident on `device`.

    The next batch is staged through pinned host memory and copied on a dedicated HIP stream while the caller still
    computes on the current one; before a batch is handed out the consumer stream waits on the copy's event, and the
    tensors are recorded on the consumer stream so the caching allocator does not recycle them early."""

    def __init__(self, loader, device):
        self.loader, self.device = loader, torch.device(device)
        assert self.device.type == "cuda", "DevicePrefetcher moves batches into HBM; it needs a GPU device"
        self.stream = torch.cuda.Stream(device=self.device)

    def _stage(self, batch):
        with torch.cuda.stream(self.stream):
            moved = [{k: self._to_device(v) for k, v in d.items()} for d in batch]
            done = torch.cuda.Event()
            done.record(self.stream)
        return moved, done

    def _to_device(self, v):
        if isinstance(v, torch.Tensor):
            return (v if v.is_pinned() else v.pin_memory()).to(self.device, non_blocking=True)
        if hasattr(v, "to") and hasattr(v, "get_fields"):  # Instances: move field by field through pinned memory
            out = type(v)(v.image_size)
            for name, field in v.get_fields().items():
                t = getattr(field, "tensor", field)
                t = (t if t.is_pinned() else t.pin_memory()).to(self.device, non_blocking=True)
                out.set(name, type(field)(t) if hasattr(field, "tensor") else t)
            return out
        return v

    @staticmethod
    def _record(obj, stream):
        if isinstance(obj, torch.Tensor):
            obj.record_stream(stream)
        elif hasattr(obj, "get_fields"):
            for field in obj.get_fields().values():
                getattr(field, "tensor", field).record_stream(stream)

    def __iter__(self):
        it = iter(self.loader)
        try:
            pending = self._stage(next(it))
        except StopIteration:
            return
        while pending is not None:
            batch, done = pending
            try:
                pending = self._stage(next(it))
            except StopIteration:
                pending = None
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(done)
            for d in batch:
                for v in d.values():
                    self._record(v, cur)
            yield batch
