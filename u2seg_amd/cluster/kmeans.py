"""Lloyd k-means over DINO embeddings on the GPU (u2seg/Instance_Clustering/shared/utils/nn_utils.py:304-405).

``run_kmeans`` mirrors the reference's ``run_kMeans`` / ``KMeans`` contract: initial centroids are
``x[randperm(N)[:K]]`` drawn from the CPU generator after ``torch.manual_seed(seed)``, a fixed number of iterations,
no convergence test, empty clusters turn into NaN rows (the reference notes this at usl-imagenet.py:135)."""
import os
import weakref

import torch
import torch.distributed as dist

from .. import _hip


_ws_cache = {}


def _update_workspace(n, d, k, device):
    """Slab workspace of u2_kmeans_update, kept between iterations (the kernel overwrites every slab it reads)."""
    need = _hip.call_nostream("u2_kmeans_update_workspace_floats", n, d, k)
    key = str(device)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < need:
        ws = _ws_cache[key] = torch.empty(need, dtype=torch.float32, device=device)
    return ws


_shadow_cache = {}
SHADOW = os.environ.get("U2_KM_SHADOW", "1") != "0"     # measurement switch: 0 = every E step reads the fp32 x
MAX_SCREENED_K = 1280                                   # four blocks of 320 centroids (kmeans.hip: KM_MAX_BLOCKS)
SHADOW_MIN_POINTS = 1 << 16                             # below this the E step is a few microseconds either way


def _shadow(x):
    """16-bit shadow of x for the first screening pass (u2_kmeans_prepare), made once per tensor: the entry is keyed by the tensor
    OBJECT (weak reference - a new tensor at a recycled address cannot match) and its version counter (any in-place write to the
    storage, through any view, invalidates it)."""
    n, d = x.shape
    if not SHADOW or n < SHADOW_MIN_POINTS or d % 32 != 0:
        return None
    key = str(x.device)
    ent = _shadow_cache.get(key)
    if ent is not None and ent[0]() is x and ent[1] == x._version and ent[2] == (n, d):
        return ent[3]
    need = _hip.call_nostream("u2_kmeans_shadow_floats", n, d)
    sh = ent[3] if ent is not None and ent[3].numel() >= need else None
    _shadow_cache.pop(key, None)
    if sh is None:
        ent = None                                          # frees the old shadow before the new one is allocated
        sh = torch.empty(need, dtype=torch.float32, device=x.device)
    _hip.call("u2_kmeans_prepare", x, sh, n, d)
    _shadow_cache[key] = (weakref.ref(x), x._version, (n, d), sh)
    return sh


def release_shadow():
    """Drops the cached shadows (half the bytes of the x they were made from)."""
    _shadow_cache.clear()


def assign(x, c, exact=False):
    """argmin_j sum_d (x_id - c_jd)^2 -> int64 [N] (u2_kmeans_assign_shadow: split-bf16 MFMA screening - the first pass over a bf16
    shadow of x that is made on the first call with a given x - and exact-fp32 MFMA products for the points it cannot decide;
    exact=True: the exact kernel for every point)."""
    n, d = x.shape
    k = c.shape[0]
    x = x.contiguous()
    labels = torch.empty(n, dtype=torch.int64, device=x.device)
    need = _hip.call_nostream("u2_kmeans_assign_workspace_floats", n, d, k)
    key = "assign:" + str(x.device)
    ent = _ws_cache.get(key)
    ws = ent[0] if ent is not None else None
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.float32, device=x.device)
    sh = None if exact or k < 2 or k > MAX_SCREENED_K else _shadow(x)
    # which counters of the workspace the call fills: "two-level" (K <= 320: both screening passes), "blocks" (K > 320 over the shadow:
    # the first pass per block of 320 centroids, then the exact kernel), None (the exact kernel alone)
    if exact or k < 2 or d % 32 != 0 or n < 256:
        mode = None
    elif k <= 320:
        mode = "two-level"
    else:
        mode = "blocks" if sh is not None else None
    _ws_cache[key] = (ws, k, mode)
    _hip.call("u2_kmeans_assign_shadow", x, sh, c.contiguous(), ws, labels, n, d, k, int(exact))
    return labels


def last_recheck_count(device):
    """How many points the last assign() on `device` sent to the exact kernel after screening (reads the workspace: one host sync);
    None when that call did not screen (exact=True, K > 1280, D % 32 != 0, fewer than 256 points)."""
    ent = _ws_cache.get("assign:" + str(device))
    if ent is None or ent[2] is None:
        return None
    ws, k, _ = ent
    return int(ws.view(torch.int32)[((k + 3) & ~3) + 1])


def last_coarse_undecided(device):
    """How many points the first (leading-bf16-piece) screening pass of the last assign() left undecided; 0 when that pass was
    switched off or skipped (reads the workspace: one host sync)."""
    ent = _ws_cache.get("assign:" + str(device))
    if ent is None or ent[2] is None:
        return None
    ws, k, mode = ent
    return int(ws.view(torch.int32)[((k + 3) & ~3) + (2 if mode == "two-level" else 1)])   # "blocks": its undecided ARE the re-checked


def update(x, labels, k):
    """c = scatter_add(x by label) / bincount(label) -> (centroids [K, D], counts [K])."""
    n, d = x.shape
    buf = torch.zeros(k * d + k, dtype=torch.float32, device=x.device)    # one fill for both
    csum, counts = buf[: k * d].view(k, d), buf[k * d:]
    _hip.call("u2_kmeans_update", x.contiguous(), labels, csum, counts, n, d, k, _update_workspace(n, d, k, x.device))
    c = torch.empty_like(csum)
    _hip.call("u2_kmeans_finalize", csum, counts, c, d, k)
    return c, counts


def kmeans(x, init_idx, niter):
    c = x[init_idx].clone()
    cl = None
    for _ in range(niter):
        cl = assign(x, c)
        c, _ = update(x, cl, c.shape[0])
    return cl, c


def update_sharded(x_local, labels_local, k, group=None):
    """The M step when the rows of x are sharded over the ranks of `group`: local partial sums and
    counts, one all-reduce of (K*D + K) floats (0.92 MB at K = 300, D = 768), then the division - every rank ends up
    with the same centroids, labels stay sharded."""
    n, d = x_local.shape
    buf = torch.zeros(k * d + k, dtype=torch.float32, device=x_local.device)
    csum, counts = buf[: k * d].view(k, d), buf[k * d :]
    if n:
        _hip.call("u2_kmeans_update", x_local.contiguous(), labels_local, csum, counts, n, d, k,
                  _update_workspace(n, d, k, x_local.device))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(buf, group=group)
    c = torch.empty((k, d), dtype=torch.float32, device=x_local.device)
    _hip.call("u2_kmeans_finalize", csum, counts, c, d, k)
    return c, counts


def kmeans_sharded(x_local, init_centroids, niter, group=None):
    """Lloyd iterations over a row shard; `init_centroids` [K, D] must be identical on every rank (e.g. rank 0 draws
    x[randperm(N)[:K]] and broadcasts it).  Returns (labels of the local rows, centroids)."""
    c = init_centroids.clone()
    cl = None
    for _ in range(niter):
        cl = assign(x_local, c)
        c, _ = update_sharded(x_local, cl, c.shape[0], group)
    return cl, c


def run_kmeans(x, num_centroids, niter=100, seed=0):
    """nn_utils.py:382-405 (run_kMeans): returns (cluster labels int64 [N], centroids fp32 [K, D])."""
    x = x.float().cuda() if not x.is_cuda else x.float()
    torch.manual_seed(seed)
    r = torch.randperm(x.shape[0])[:num_centroids]
    return kmeans(x, r.to(x.device), niter)
