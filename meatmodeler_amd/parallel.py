"""Sharding of the hot path over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" in the CPU tests).  SURVEY.md §8e:

  * detection / matching: consecutive frame pairs are independent -> block partition of the pairs; a rank detects its
    frame block plus one halo frame (recomputed, not exchanged); match lists (KBs) are all-gathered so every rank
    links the same tracks;
  * bundle adjustment: points (with all their observations) are partitioned, cameras are replicated; the only data-path
    collective is the all-reduce(sum) of the camera-side blocks: B, g_c, the reduced camera system S, v and a few
    scalars per trust-region iteration.
"""
import numpy as np


def block_range(n, rank, world):
    """[lo, hi) of a balanced contiguous block partition of range(n)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pair_block(n_frames, rank, world):
    """Frame pairs (k, k+1) owned by `rank` and the frames it must detect (its pairs' frames incl. the halo)."""
    p_lo, p_hi = block_range(max(n_frames - 1, 0), rank, world)
    if p_hi <= p_lo:
        return (p_lo, p_lo), (0, 0)
    return (p_lo, p_hi), (p_lo, p_hi + 1)


def partition_points(fi, pi, n_points, rank, world):
    """Contiguous split of the points balanced by observation count.  Returns (p_lo, p_hi, obs_mask) where obs_mask
    selects the observations of the local points (point-major input keeps them contiguous)."""
    pi = np.asarray(pi)
    counts = np.bincount(pi, minlength=n_points)
    cum = np.concatenate([[0], np.cumsum(counts)])
    total = cum[-1]
    bounds = [int(np.searchsorted(cum, total * r / world, side="left")) for r in range(world)] + [n_points]
    bounds[0] = 0
    for r in range(1, world + 1):
        bounds[r] = max(bounds[r], bounds[r - 1])
    lo, hi = bounds[rank], bounds[rank + 1]
    mask = (pi >= lo) & (pi < hi)
    return lo, hi, mask


def split_by_weight(cum, world):
    """Contiguous split of items 0..n-1 into `world` parts balanced by weight.  cum [n+1] = exclusive running sum of the
    weights (non-decreasing integers, cum[0] = 0; numpy array or device tensor).  -> list of world + 1 item bounds.
    Part r starts at the first item whose running sum reaches total * r / world (integer arithmetic: ceil)."""
    n = len(cum) - 1
    try:
        import torch
        is_t = torch.is_tensor(cum)
    except ImportError:
        is_t = False
    if is_t:
        c = cum.to(torch.int64)
        total = int(c[-1].item())
        tg = torch.tensor([-(-total * r // world) for r in range(world)], dtype=torch.int64, device=c.device)
        bounds = torch.searchsorted(c, tg, right=False).tolist() + [n]
    else:
        c = np.asarray(cum).astype(np.int64)
        total = int(c[-1])
        bounds = [int(np.searchsorted(c, -(-total * r // world), side="left")) for r in range(world)] + [n]
    bounds[0] = 0
    for r in range(1, world + 1):
        bounds[r] = min(max(bounds[r], bounds[r - 1]), n)
    if n >= world:          # no empty part as long as there are enough items (one very heavy item cannot starve a rank)
        for r in range(1, world):
            bounds[r] = min(max(bounds[r], bounds[r - 1] + 1), n - (world - r))
    return bounds


def partition_tracks(track_ptr, rank, world):
    """Same split for CSR tracks (point-major observations): -> (p_lo, p_hi, o_lo, o_hi) — the rank's points and
    the contiguous observation range that belongs to them.  `track_ptr` may be a device tensor (two small read-backs,
    the CSR itself stays on the device)."""
    bounds = split_by_weight(track_ptr, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    return lo, hi, int(track_ptr[lo]), int(track_ptr[hi])


class AllReduce:
    """Callable all-reduce over a torch.distributed process group (in place)."""

    def __init__(self, group=None, force=False):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.force = force          # one-rank group: still go through the backend (trivial sums, real path)
        self.calls = 0

    def __call__(self, tensor, op="sum"):
        if self.world_size == 1 and not self.force:
            return tensor
        self.calls += 1
        d = self.dist
        d.all_reduce(tensor, op=d.ReduceOp.SUM if op == "sum" else d.ReduceOp.MAX, group=self.group)
        return tensor


def gather_varlen(local, world, dist, group=None):
    """All-gather of variable-length int32/float arrays (numpy, host) with a size prefix.  Returns the list of
    per-rank arrays.  Match lists are KBs, so this goes through host memory."""
    import torch
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    flat = np.ascontiguousarray(local).reshape(-1)
    n = torch.tensor([flat.size], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes + [1])
    buf = torch.zeros(m, dtype=torch.from_numpy(flat[:0]).dtype, device=dev)
    buf[:flat.size] = torch.from_numpy(flat).to(dev)
    outs = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return [o[:s].cpu().numpy() for o, s in zip(outs, sizes)]


def gather_blocks(local, counts, dist, group=None):
    """All-gather of per-rank blocks of rows whose row counts are known everywhere (block partitions are pure
    arithmetic on (n, world)): `local` [counts[rank], ...] device (or CPU) tensor -> [sum(counts), ...] on the same
    device, rank order.  Stays on the device for "nccl" (RCCL over xGMI); no size exchange, no host copy."""
    import torch
    world = len(counts)
    rank = dist.get_rank(group)
    assert local.shape[0] == counts[rank], (local.shape, counts, rank)
    m = max(max(counts), 1)
    tail = tuple(local.shape[1:])
    buf = torch.zeros((m,) + tail, dtype=local.dtype, device=local.device)
    buf[:counts[rank]] = local
    outs = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(outs, buf, group=group)
    return torch.cat([o[:c] for o, c in zip(outs, counts)], 0)
