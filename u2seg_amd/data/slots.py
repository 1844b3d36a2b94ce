"""Batch transport from DataLoader workers to the training process through preallocated shared-memory slots.

torch's default hand-over pickles every tensor of every sample into a fresh shared-memory segment that the training
process has to map and fault in page by page; on the GPU boxes this run uses that costs ~50 us per 4 KB page - 1.4 s for
the ~100 MB of a 16-image batch, 20 times the cost of producing it.  Here a worker maps a whole per-GPU batch and packs it
into one slot of a ring that was allocated (and faulted in) once before the workers were forked; only a small layout
record travels through the queue.  The training process copies the slot into pinned memory with one memcpy, ships it with
one host-to-device copy and rebuilds the tensors on the device as views of that block.

Packing also shrinks what has to move: label maps (int64 in the mapper's contract, values 0..255) travel as bytes and
boolean instance masks as bits (8x each); both are widened again on the device."""
import numpy as np
import torch
import torch.utils.data as torchdata

from ..structures import Instances

ALIGN = 256


def _align(n):
    return (n + ALIGN - 1) // ALIGN * ALIGN


class SlotRing:
    """`num_workers * depth` slots of `slot_bytes`; worker w cycles through slots w * depth ... w * depth + depth - 1.
    depth must exceed the number of batches a worker can have in flight (DataLoader prefetch_factor) by two: one being
    read by the training process, one of slack."""

    def __init__(self, num_workers, slot_bytes, depth):
        self.depth, self.slot_bytes = depth, int(slot_bytes)
        self.mem = torch.zeros((max(1, num_workers) * depth, self.slot_bytes), dtype=torch.uint8).share_memory_()

    def view(self, index):
        return self.mem[index].numpy()


def _encode(t):
    """(numpy array to store, encoding, original dtype name, original shape)."""
    a = t.numpy()
    if t.dtype == torch.int64 and a.ndim == 2 and a.size > 4096 and 0 <= int(a.min()) and int(a.max()) <= 255:
        return a.astype(np.uint8), "u8", "int64", tuple(a.shape)
    if t.dtype == torch.bool and a.ndim == 3 and a.shape[0] > 0:
        return np.packbits(a, axis=-1), "bits", "bool", tuple(a.shape)
    return np.ascontiguousarray(a), "raw", str(t.dtype).replace("torch.", ""), tuple(a.shape)


class PackedBatch:
    __slots__ = ("slot", "nbytes", "samples")

    def __init__(self, slot, nbytes, samples):
        self.slot, self.nbytes, self.samples = slot, nbytes, samples


class BatchPacker:
    """collate_fn of the training loader: list of mapped dicts -> PackedBatch (or the list itself when there is no ring or
    the batch does not fit its slot)."""

    def __init__(self, ring):
        self.ring = ring
        self._count = 0  # per process: every worker holds its own copy

    def __call__(self, batch):
        if self.ring is None:
            return batch
        staged, cursor, samples = [], 0, []
        for d in batch:
            entries = []
            for key, v in d.items():
                if isinstance(v, torch.Tensor):
                    arr, enc, dtype, shape = _encode(v)
                    entries.append((key, "tensor", enc, dtype, shape, cursor, arr.shape))
                    staged.append((cursor, arr))
                    cursor = _align(cursor + arr.nbytes)
                elif isinstance(v, Instances):
                    fields = []
                    for name, f in v.get_fields().items():
                        arr, enc, dtype, shape = _encode(getattr(f, "tensor", f))
                        fields.append((name, type(f) if hasattr(f, "tensor") else None, enc, dtype, shape, cursor, arr.shape))
                        staged.append((cursor, arr))
                        cursor = _align(cursor + arr.nbytes)
                    entries.append((key, "instances", tuple(v.image_size), fields))
                else:
                    entries.append((key, "value", v))
            samples.append(entries)
        if cursor > self.ring.slot_bytes:
            return batch
        info = torchdata.get_worker_info()
        slot = (info.id if info is not None else 0) * self.ring.depth + self._count % self.ring.depth
        self._count += 1
        buf = self.ring.view(slot)
        for off, arr in staged:
            buf[off:off + arr.nbytes] = arr.reshape(-1).view(np.uint8)
        return PackedBatch(slot, cursor, samples)


_SHIFTS = {}


def _decode(block, enc, dtype, shape, off, stored_shape):
    """Tensor of the original dtype / shape from its stored form inside `block` (uint8, host or device)."""
    n = int(np.prod(stored_shape)) * (1 if enc != "raw" else torch.empty((), dtype=getattr(torch, dtype)).element_size())
    raw = block[off:off + n]
    if enc == "raw":
        return raw.view(getattr(torch, dtype)).view(shape)
    if enc == "u8":
        return raw.view(stored_shape).to(torch.int64)
    if raw.device.type == "cpu":
        return torch.from_numpy(np.unpackbits(raw.view(stored_shape).numpy(), axis=-1, count=shape[2]).view(np.bool_))
    key = raw.device
    if key not in _SHIFTS:
        _SHIFTS[key] = torch.arange(7, -1, -1, dtype=torch.uint8, device=raw.device)
    bits = (raw.view(stored_shape).unsqueeze(-1) >> _SHIFTS[key]) & 1  # most significant bit first, like np.packbits
    return bits.view(stored_shape[0], stored_shape[1], -1)[:, :, : shape[2]].to(torch.bool)


def unpack(block, samples):
    """list of mapped dicts rebuilt from a block (torch uint8, on the host or on the device) and its layout."""
    out = []
    for entries in samples:
        d = {}
        for e in entries:
            if e[1] == "tensor":
                _, _, enc, dtype, shape, off, stored = e
                d[e[0]] = _decode(block, enc, dtype, shape, off, stored)
            elif e[1] == "instances":
                inst = Instances(e[2])
                for name, wrapper, enc, dtype, shape, off, stored in e[3]:
                    t = _decode(block, enc, dtype, shape, off, stored)
                    inst.set(name, wrapper(t) if wrapper is not None else t)
                d[e[0]] = inst
            else:
                d[e[0]] = e[2]
        out.append(d)
    return out


class HostUnpacked:
    """Iterable over plain lists of mapped dicts for consumers that stay on the host: copies each packed batch out of its
    slot.  `packed` exposes the raw stream for DevicePrefetcher."""

    def __init__(self, packed, ring):
        self.packed, self.ring = packed, ring

    def __iter__(self):
        for b in self.packed:
            if isinstance(b, PackedBatch):
                block = torch.from_numpy(self.ring.view(b.slot)[: b.nbytes].copy())
                yield unpack(block, b.samples)
            else:
                yield b
