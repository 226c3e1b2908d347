"""Multi-GPU sharding of the hot path: one process per GPU (torch.distributed; backend "nccl" == RCCL over
xGMI on ROCm, "gloo" on CPU for the tests).

Every target (light curve / TPF) is independent in the reference — callers loop over targets
(src/lightkurve/targetpixelfile.py:1967-1974, correctors/metrics.py:64-84) — so the batch is cut into
contiguous, cost-balanced blocks of targets, one per rank, and computed with NO data-path collective.
The only exchange is the optional all-gather of the per-shard results (power spectra, or just the per-target
(max power, argmax) summaries) so that every rank ends with the whole answer.
"""
import numpy as np

__all__ = ["partition_by_cost", "shard_bounds", "all_gather_equal", "all_gather_rows", "device_gather_available",
           "sharded_map", "sharded_map_ragged"]


def partition_by_cost(costs, world):
    """Contiguous partition of len(costs) items into ``world`` blocks with near-equal total cost.
    Returns int64 bounds[world + 1] (bounds[r]..bounds[r+1] is rank r's block; blocks may be empty)."""
    costs = np.asarray(costs, dtype=np.float64)
    n = len(costs)
    if world < 1:
        raise ValueError("world must be >= 1")
    if np.any(costs < 0):
        raise ValueError("costs must be non-negative")
    bounds = np.zeros(world + 1, dtype=np.int64)
    if n == 0:
        return bounds
    csum = np.concatenate([[0.0], np.cumsum(costs)])
    total = csum[-1]
    for r in range(1, world):
        # first index whose prefix cost reaches r/world of the total (ties -> nearest boundary)
        target = total * r / world
        j = int(np.searchsorted(csum, target, side="left"))
        if j > 0 and abs(csum[j - 1] - target) <= abs(csum[min(j, n)] - target):
            j -= 1
        bounds[r] = min(max(j, bounds[r - 1]), n)
    bounds[world] = n
    return bounds


def shard_bounds(n_items, world, costs=None):
    """bounds[world+1]: equal item counts (ceil) unless per-item ``costs`` (e.g. N_b * M) are given."""
    if costs is not None:
        return partition_by_cost(costs, world)
    per = -(-n_items // world) if n_items else 0
    return np.minimum(np.arange(world + 1, dtype=np.int64) * per, n_items)


def _dist():
    """torch.distributed, or None in an interpreter without torch (the conda + astropy environment the seams live
    in): no torch means no process group, and every caller treats that as "one rank"."""
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    return dist if dist.is_available() else None


def all_gather_equal(out, local, group=None, async_op=False):
    """The collective itself, on tensors that already live where the backend wants them (HBM for "nccl" = RCCL over xGMI,
    host memory for "gloo"): ``out[(world, n, ...)]`` <- every rank's ``local[(n, ...)]``.  No staging, no allocation —
    ``bench.py``'s multi-GPU step and ``all_gather_rows`` below both end here.  Returns the work handle when ``async_op``."""
    dist = _dist()
    world = dist.get_world_size(group)
    if out.shape[0] != world or tuple(out.shape[1:]) != tuple(local.shape):
        raise ValueError("out must be (world_size,) + local.shape, got %s for local %s" % (tuple(out.shape), tuple(local.shape)))
    return dist.all_gather_into_tensor(out.view((world * local.shape[0],) + tuple(local.shape[1:])), local.contiguous(),
                                       group=group, async_op=async_op)


def all_gather_rows(local, bounds, group=None):
    """All-gather row blocks of unequal height: rank r contributes ``local`` of shape (bounds[r+1]-bounds[r], ...);
    every rank gets the (bounds[-1], ...) concatenation.  torch.Tensor in -> torch.Tensor out on the same device: with
    RCCL a device tensor (what the ``*_batch_dev`` entry points fill) never touches the host.  numpy in -> numpy out, staged
    through the backend's device (one copy each way: use the tensor form to avoid them).  One padded all-gather."""
    dist = _dist()
    if dist is None or not dist.is_initialized():
        return local
    import torch
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = np.diff(np.asarray(bounds, dtype=np.int64))
    if len(counts) != world:
        raise ValueError("bounds must have world_size + 1 entries")
    is_np = isinstance(local, np.ndarray)
    backend = dist.get_backend(group)
    if backend == "nccl" and not torch.cuda.is_available():
        raise RuntimeError("torch sees no GPU: in a process that uses torch.distributed with RCCL, `import torch` must come "
                           "before the first lightkurve_amd compute call (torch bundles its own ROCm runtime; "
                           "liblkhip.so loads /opt/rocm's)")
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    x = torch.from_numpy(np.ascontiguousarray(local)).to(dev) if is_np else local.contiguous()
    if x.shape[0] != counts[rank]:
        raise ValueError("rank %d holds %d rows, bounds say %d" % (rank, x.shape[0], counts[rank]))
    cmax = int(counts.max()) if world else 0
    tail = tuple(x.shape[1:])
    if int(counts.min()) == cmax:
        pad = x                                      # equal shards: gather straight out of the caller's tensor
    else:
        pad = torch.zeros((cmax,) + tail, dtype=x.dtype, device=x.device)
        pad[: x.shape[0]] = x
    out = torch.empty((world, cmax) + tail, dtype=x.dtype, device=x.device)
    all_gather_equal(out, pad, group=group)
    full = out.view((world * cmax,) + tail) if int(counts.min()) == cmax else \
        torch.cat([out[r, : int(counts[r])] for r in range(world)], dim=0)
    return full.cpu().numpy() if is_np else full


def device_gather_available(group=None):
    """True when results can be gathered without leaving HBM: a RCCL ("nccl") process group and a GPU torch can see."""
    dist = _dist()
    if dist is None or not dist.is_initialized() or dist.get_backend(group) != "nccl":
        return False
    import torch
    return torch.cuda.is_available()


def sharded_map(items, fn, costs=None, gather=True, group=None):
    """Apply ``fn(list_of_local_items) -> array[(n_local, ...)]`` to this rank's contiguous block of ``items`` and
    (optionally) all-gather the rows so every rank returns the full (len(items), ...) result in input order.
    Without an initialised process group this is just ``fn(items)``."""
    dist = _dist()
    if dist is None or not dist.is_initialized():
        return fn(list(items))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    bounds = shard_bounds(len(items), world, costs)
    local = fn(list(items[int(bounds[rank]):int(bounds[rank + 1])]))
    if not gather:
        return local
    return all_gather_rows(local, bounds, group=group)


def sharded_map_ragged(items, fn, lengths, costs=None, gather=True, group=None, dtype=np.float64):
    """``sharded_map`` for results of unequal length: ``fn(local_items)`` returns one 1-D array per item (``lengths[i]``
    long — every rank can compute the lengths of all items, e.g. ``len(lc)``).  The local rows are packed into a
    NaN-padded block for the all-gather and unpacked again: every rank returns the list of all ``len(items)`` arrays."""
    items = list(items)
    lengths = [int(n) for n in lengths]
    if len(lengths) != len(items):
        raise ValueError("lengths must hold one entry per item")
    dist = _dist()
    if dist is None or not dist.is_initialized() or not gather:
        if dist is None or not dist.is_initialized():
            return list(fn(items))
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        bounds = shard_bounds(len(items), world, costs if costs is not None else lengths)
        return list(fn(items[int(bounds[rank]):int(bounds[rank + 1])]))
    lmax = max(lengths) if lengths else 0

    def padded(local):
        rows = fn(local)
        out = np.full((len(local), lmax), np.nan, dtype=dtype)
        for i, r in enumerate(rows):
            out[i, : len(r)] = r
        return out

    full = sharded_map(items, padded, costs=costs if costs is not None else lengths, gather=True, group=group)
    return [full[i, : lengths[i]] for i in range(len(items))]
