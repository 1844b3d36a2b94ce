"""Brute-force kNN over the embeddings on the GPU and the first-order density score derived from it
(u2seg/Instance_Clustering/shared/utils/nn_utils.py:204-299, selective_labeling/usl-imagenet.py:105-110).

``kNN`` and ``partitioned_kNN`` keep the reference's names, argument meaning and return order.  The reference tiles both
sides into partitions of 130 000 rows because the pairwise reduction has to fit its GPU, then merges the per-partition
lists; the merged lists are the global K smallest per row (its own ``verify`` branch asserts exactly that), which is what
one pass of u2_knn over the resident feature matrix produces, so ``partitions_size`` only survives as an accepted
argument."""
import ctypes

import torch
import torch.distributed as dist

from .. import _hip


def _workspace(nq, nt, d, k, device):
    n = ctypes.c_longlong(0)
    rc = _hip.call_nostream("u2_knn_workspace_ints", nq, nt, d, k, ctypes.addressof(n))
    if rc != 0:
        raise ValueError("kNN: K must lie in [1, 28] (got %d)" % k)
    return torch.empty(n.value, dtype=torch.int32, device=device)


def kNN(x_train, x_test, K=20):
    """nn_utils.py:204-227: (ind_knn int64 [N_test, K], d_knn fp32 [N_test, K]); squared L2 distances in ascending order."""
    assert len(x_train.shape) == 2
    assert len(x_test.shape) == 2
    assert x_train.is_cuda and x_test.is_cuda and x_train.shape[1] == x_test.shape[1]
    x_train, x_test = x_train.float().contiguous(), x_test.float().contiguous()
    nq, nt, d = x_test.shape[0], x_train.shape[0], x_train.shape[1]
    if nt < K:
        raise ValueError("kNN: %d train rows < K = %d (the reference assumes at least K rows, nn_utils.py:236)" % (nt, K))
    # a query row with non-finite features has no ranked neighbour: it keeps NaN distances and index -1
    d_knn = torch.full((nq, K), float("nan"), dtype=torch.float32, device=x_test.device)
    ind_knn = torch.full((nq, K), -1, dtype=torch.int64, device=x_test.device)
    ws = _workspace(nq, nt, d, K, x_test.device)
    _hip.call("u2_knn", x_test, x_train, ws, d_knn, ind_knn, nq, nt, d, K)
    return ind_knn, d_knn


def partitioned_kNN(feats_list, K=20, partitions_size=130000):
    """nn_utils.py:230-299 (recompute=True branch, without the .npy cache): (d_knns fp32 [N, K], ind_knns int64 [N, K]) of
    every row against all rows, the row itself included at distance 0."""
    del partitions_size  # the whole matrix is resident; see the module docstring
    x = feats_list if feats_list.is_cuda else feats_list.cuda()
    ind_knns, d_knns = kNN(x, x, K=K)
    return d_knns, ind_knns


def partitioned_kNN_sharded(feats_local, K=20, group=None):
    """The same lists when the rows of the feature matrix are sharded over the ranks of `group`:
    one all-gather of the shards (the only exchange this step has), then every rank ranks its own rows against the full
    set.  Returned indices address the gathered matrix (rank-major row order); lists stay sharded."""
    x = feats_local.float().contiguous()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        ind, d = kNN(x, x, K=K)
        return d, ind
    full = gather_rows(x, group)
    ind, d = kNN(full, x, K=K)
    return d, ind


def gather_rows(x, group=None):
    """Concatenate the (ragged) row shards of every rank in rank order: sizes first, then padded shards."""
    world = dist.get_world_size(group)
    sizes = [torch.zeros(1, dtype=torch.int64, device=x.device) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([x.shape[0]], dtype=torch.int64, device=x.device), group=group)
    sizes = [int(s) for s in sizes]
    mine = torch.zeros((max(sizes), x.shape[1]), dtype=x.dtype, device=x.device)
    mine[: x.shape[0]] = x
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], dim=0)


def first_order_density(d_knns):
    """usl-imagenet.py:108-110: score_first_order = 1 / mean_k d_knn."""
    return 1 / d_knns.mean(dim=1)
