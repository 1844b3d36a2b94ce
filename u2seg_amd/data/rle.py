"""COCO run-length masks without pycocotools (the reference decodes its pseudo-label masks with
pycocotools.mask.decode: detectron2/data/detection_utils.py:306-309, 430-431; INPUT.MASK_FORMAT "bitmask").

Format (cocoapi common/maskApi.c, restated): a mask of size h x w is scanned column-major; `counts` holds alternating
run lengths starting with a run of zeros.  The compressed string form writes every count, from the third on as the
difference to the count two places earlier, in 5-bit groups, low group first: bit 5 of a group marks "more follows",
the last group's bit 4 is the sign, each character is group + 48."""
import numpy as np


def _counts_from_string(s):
    if isinstance(s, bytes):
        s = s.decode("ascii")
    counts, p, n = [], 0, len(s)
    while p < n:
        x, k, more = 0, 0, True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1F) << (5 * k)
            more = bool(c & 0x20)
            p += 1
            k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(counts) > 2:
            x += counts[-2]
        counts.append(x)
    return counts


def _string_from_counts(counts):
    out = []
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            g = x & 0x1F
            x >>= 5
            more = (x != -1) if (g & 0x10) else (x != 0)
            if more:
                g |= 0x20
            out.append(chr(g + 48))
    return "".join(out)


def counts_of(rle):
    c = rle["counts"]
    return [int(v) for v in c] if isinstance(c, (list, tuple, np.ndarray)) else _counts_from_string(c)


def decode(rle):
    """{"size": [h, w], "counts": str | bytes | list} -> uint8 [h, w] of 0/1 (pycocotools.mask.decode of one RLE)."""
    h, w = rle["size"]
    counts = np.asarray(counts_of(rle), dtype=np.int64)
    if counts.sum() != h * w or (counts < 0).any():
        raise ValueError("RLE counts sum to %d, mask has %d pixels" % (int(counts.sum()), h * w))
    values = np.zeros(len(counts), dtype=np.uint8)
    values[1::2] = 1
    return np.repeat(values, counts).reshape((w, h)).T  # column-major scan; a Fortran-ordered view like pycocotools'


def encode(mask):
    """uint8 / bool [h, w] -> {"size": [h, w], "counts": compressed str} (pycocotools.mask.encode of one mask)."""
    m = np.asarray(mask)
    assert m.ndim == 2, m.shape
    h, w = m.shape
    flat = (m.T.reshape(-1) != 0).astype(np.int8)
    change = np.flatnonzero(np.diff(flat)) + 1
    bounds = np.concatenate([[0], change, [flat.size]])
    counts = np.diff(bounds).tolist()
    if flat.size and flat[0] == 1:
        counts = [0] + counts
    if flat.size == 0:
        counts = []
    return {"size": [int(h), int(w)], "counts": _string_from_counts(counts)}


def compress(rle):
    """Uncompressed {"counts": list} -> compressed string form (pycocotools.mask.frPyObjects on an RLE dict)."""
    return {"size": list(rle["size"]), "counts": _string_from_counts(counts_of(rle))}


def area(rle):
    return int(sum(counts_of(rle)[1::2]))


def to_bbox(rle):
    """[x, y, w, h] of the tight box (pycocotools.mask.toBbox); all zeros for an empty mask."""
    m = decode(rle)
    ys, xs = np.nonzero(m)
    if ys.size == 0:
        return [0.0, 0.0, 0.0, 0.0]
    return [float(xs.min()), float(ys.min()), float(xs.max() - xs.min() + 1), float(ys.max() - ys.min() + 1)]
