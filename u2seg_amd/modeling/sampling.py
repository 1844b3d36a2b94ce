"""Random fg/bg subsampling (detectron2/modeling/sampling.py:9-54) with an injectable permutation source.

The reference draws ``torch.randperm(n, device=labels.device)``, which is not reproducible across devices; parity
tests install a deterministic ``perm_fn`` (shared with the CPU oracle) through ``set_permutation_source``."""
import torch

_perm_fn = None
_key_fn = None


def set_key_source(fn):
    """fn(shape, device) -> float32 tensor of sampling keys in [0, 1); None restores torch.rand on the device.

    The batched samplers (RPN._subsample_batched, ROIHeads._label_and_sample_padded - the branch every training step
    runs) draw ONE key per candidate and keep the candidates with the smallest keys.  That is the reference's
    ``positive[randperm(P)[:num_pos]]`` (sampling.py:38-54) with the permutation ``argsort(key[positive])``; parity tests
    inject the keys here and hand the oracle that permutation, so the two can be compared bit for bit."""
    global _key_fn
    _key_fn = fn


def random_keys(shape, device):
    if _key_fn is not None:
        k = _key_fn(tuple(shape), device)
        assert tuple(k.shape) == tuple(shape) and k.dtype == torch.float32
        return k.to(device)
    return torch.rand(shape, device=device)


def set_permutation_source(fn):
    """fn(n: int, device) -> LongTensor permutation of range(n); None restores torch.randperm on the device."""
    global _perm_fn
    _perm_fn = fn


def permutation_source():
    return _perm_fn


def _perm(n, device):
    if _perm_fn is not None:
        return _perm_fn(n, device).to(device)
    return torch.randperm(n, device=device)


def subsample_labels(labels, num_samples, positive_fraction, bg_label):
    positive = torch.nonzero((labels != -1) & (labels != bg_label), as_tuple=True)[0]
    negative = torch.nonzero(labels == bg_label, as_tuple=True)[0]
    num_pos = int(num_samples * positive_fraction)
    num_pos = min(positive.numel(), num_pos)
    num_neg = num_samples - num_pos
    num_neg = min(negative.numel(), num_neg)
    perm1 = _perm(positive.numel(), positive.device)[:num_pos]
    perm2 = _perm(negative.numel(), negative.device)[:num_neg]
    return positive[perm1], negative[perm2]
