"""Multi-GPU layer: images are independent, so a batch is SHARDED across ranks (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests) with no
collective on the data path.  The only exchange is the result hand-off the reference's consumer
needs in one place: the ragged line lists are gathered to one rank:

    1. all_gather of the int32 per-image line counts (tiny, fixed size);
    2. gather of each rank's compacted lines, padded to the largest rank total.

lsd_line is moved as 10 x int64 words per line (a byte copy: no float canonicalisation on the way).
"""
import numpy as np
import torch
import torch.distributed as dist

WORDS_PER_LINE = 10   # sizeof(lsd_line) == 80 bytes


def shard_range(n_items, world, rank):
    """Contiguous shard: image i goes to rank i*world//n_items.  Returns [lo, hi)."""
    lo = (rank * n_items + world - 1) // world
    hi = ((rank + 1) * n_items + world - 1) // world
    return lo, hi


def compact_lines(lines_i64, counts):
    """[n, max_lines, 10] int64 + [n] int32 -> dense [sum(min(count,max_lines)), 10] int64 (image-major)."""
    n, max_lines, _ = lines_i64.shape
    c = counts.clamp(max=max_lines).to(torch.int64)
    mask = torch.arange(max_lines, device=lines_i64.device)[None, :] < c[:, None]
    return lines_i64[mask], c


def gather_line_lists(lines_i64, counts, n_total, dst=0, group=None):
    """Gathers every rank's line lists to `dst`.

    lines_i64 [n_local, max_lines, 10] int64, counts [n_local] int32 for this rank's shard
    (shard_range(n_total, world, rank)).  Returns on dst: (offsets int64[n_total+1], lines int64[total,10])
    in global image order; on other ranks (None, None)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(n_total, world, rank)
    assert counts.numel() == hi - lo, (counts.numel(), lo, hi)
    dev = lines_i64.device
    per = max(shard_range(n_total, world, r)[1] - shard_range(n_total, world, r)[0] for r in range(world))
    dense, c = compact_lines(lines_i64, counts)
    # step 1: counts (padded to the largest shard so the all-gather is regular)
    cpad = torch.zeros(per, dtype=torch.int64, device=dev)
    cpad[:hi - lo] = c
    allc = torch.empty(world * per, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allc, cpad, group=group)
    allc = allc.view(world, per).cpu()
    totals = allc.sum(1)
    pad_rows = int(totals.max().item())
    # step 2: payload, padded to the largest rank total
    payload = torch.zeros((max(pad_rows, 1), WORDS_PER_LINE), dtype=torch.int64, device=dev)
    payload[:dense.shape[0]] = dense
    if rank == dst:
        bufs = [torch.empty_like(payload) for _ in range(world)]
        dist.gather(payload, bufs, dst=dst, group=group)
        parts, cnts = [], []
        for r in range(world):
            rlo, rhi = shard_range(n_total, world, r)
            parts.append(bufs[r][:int(totals[r])])
            cnts.append(allc[r, :rhi - rlo])
        lines = torch.cat(parts, 0)
        cnt = torch.cat(cnts)
        offsets = torch.zeros(n_total + 1, dtype=torch.int64)
        offsets[1:] = torch.cumsum(cnt, 0)
        return offsets, lines
    dist.gather(payload, None, dst=dst, group=group)
    return None, None


def lines_to_numpy(lines_i64, line_dtype):
    """int64[total,10] -> structured numpy array with the structLinesInfo fields."""
    a = lines_i64.cpu().numpy()
    return np.ascontiguousarray(a).view(np.uint8).reshape(-1, 80).view(line_dtype).reshape(-1)
