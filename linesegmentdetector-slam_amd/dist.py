"""Multi-GPU layer: images are independent, so a batch is SHARDED across ranks (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests) with no
collective on the data path.  The only exchange is the result hand-off the reference's consumer
needs in one place: the ragged line lists are gathered to one rank:

    1. all_gather of the per-image line counts (tiny, fixed size);
    2. gather of each rank's compacted lines in a fixed-capacity slab (regular collective, no host
       synchronisation in the per-step path).

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
    c = counts.clamp(min=0, max=max_lines).to(torch.int64)          # (-1: an image the region stage gave up -- no lines)
    mask = torch.arange(max_lines, device=lines_i64.device)[None, :] < c[:, None]
    return lines_i64[mask], c


def pack_lines(lines_i64, counts, cap_rows):
    """[n, max_lines, 10] int64 + [n] int32 -> ([cap_rows, 10] int64 with image i's lines at rows [off[i], off[i] + c[i]),
    c int64[n], overflow bool[1]) -- all on the device, no host synchronisation (rows that do not fit go to a dump row)."""
    n, max_lines, _ = lines_i64.shape
    dev = lines_i64.device
    c = counts.clamp(min=0, max=max_lines).to(torch.int64)          # (-1: an image the region stage gave up -- no lines, flagged below)
    off = torch.cumsum(c, 0) - c
    k = torch.arange(max_lines, device=dev)[None, :]
    dest = off[:, None] + k
    dest = torch.where((k < c[:, None]) & (dest < cap_rows), dest, torch.full_like(dest, cap_rows))
    payload = torch.zeros((cap_rows + 1, WORDS_PER_LINE), dtype=torch.int64, device=dev)
    payload[dest.reshape(-1)] = lines_i64.reshape(-1, WORDS_PER_LINE)
    return payload[:cap_rows], c, ((c.sum() > cap_rows) | (counts < 0).any() | (counts > max_lines).any()).reshape(1)


def gather_line_lists(lines_i64, counts, n_total, dst=0, group=None, cap_rows=None, dense=True):
    """Gathers every rank's line lists to `dst`.

    lines_i64 [n_local, max_lines, 10] int64, counts [n_local] int32 for this rank's shard
    (shard_range(n_total, world, rank)).  Two regular collectives and NO host synchronisation on any rank:
      1. all_gather of the per-image counts (padded to the largest shard);
      2. gather of a fixed-capacity slab of cap_rows lines per rank (default: the largest shard x max_lines, which cannot
         overflow; a smaller cap_rows keeps the slab near the payload -- an overflow is flagged, never silent).
    dense=False (the per-step path): returns on dst the device tensors (counts int64[world, per], lines int64[world, cap_rows, 10],
    overflow bool[world]); rank r's lines sit compacted at the head of lines[r] in image order.  dense=True additionally trims them
    on dst into (offsets int64[n_total+1], lines int64[total, 10]) in global image order -- this needs the totals on the
    host, i.e. one synchronisation on dst only, after both collectives.  Other ranks get (None, None[, None])."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = shard_range(n_total, world, rank)
    assert counts.numel() == hi - lo, (counts.numel(), lo, hi)
    dev = lines_i64.device
    per = max(shard_range(n_total, world, r)[1] - shard_range(n_total, world, r)[0] for r in range(world))
    if cap_rows is None:
        cap_rows = max(per * lines_i64.shape[1], 1)
    payload, c, over = pack_lines(lines_i64, counts, cap_rows)
    # step 1: counts (padded to the largest shard so the all-gather is regular) + the overflow flags
    cpad = torch.zeros(per + 1, dtype=torch.int64, device=dev)
    cpad[:hi - lo] = c
    cpad[per] = over.to(torch.int64)[0]
    allc = torch.empty(world * (per + 1), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allc, cpad, group=group)
    allc = allc.view(world, per + 1)
    # step 2: the fixed-capacity slabs
    if rank == dst:
        bufs = [torch.empty_like(payload) for _ in range(world)]
        dist.gather(payload, bufs, dst=dst, group=group)
        slabs = torch.stack(bufs, 0)
        if not dense:
            return allc[:, :per], slabs, allc[:, per] != 0
        hc = allc.cpu()                                    # (dst only, after the collectives)
        if bool((hc[:, per] != 0).any()):
            raise RuntimeError("gather_line_lists: a rank produced more than cap_rows=%d lines" % cap_rows)
        parts, cnts = [], []
        for r in range(world):
            rlo, rhi = shard_range(n_total, world, r)
            cr = hc[r, :rhi - rlo]
            parts.append(slabs[r, :int(cr.sum())])
            cnts.append(cr)
        lines = torch.cat(parts, 0)
        cnt = torch.cat(cnts)
        offsets = torch.zeros(n_total + 1, dtype=torch.int64)
        offsets[1:] = torch.cumsum(cnt, 0)
        return offsets, lines
    dist.gather(payload, None, dst=dst, group=group)
    return (None, None) if dense else (None, None, None)


class _RawDeviceBytes:
    """A raw device address as a __cuda_array_interface__ object: torch.as_tensor wraps it without a copy."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def torch_comm(group=None):
    """An lsd_comm (include/lsd_hip.h) over a torch.distributed process group: the callback the C ABI's lsd_gather_lines calls wraps
    the raw device buffers and issues dist.all_gather_into_tensor (backend nccl = RCCL over xGMI) on the stream it is given.
    Keep the returned object alive while the library may call it (it owns the ctypes callback)."""
    import importlib
    lsd = importlib.import_module(__package__)

    def _all_gather(user, d_send, d_recv, nbytes, stream):
        try:
            world = dist.get_world_size(group)
            send = torch.as_tensor(_RawDeviceBytes(d_send, nbytes), device="cuda")
            recv = torch.as_tensor(_RawDeviceBytes(d_recv, nbytes * world), device="cuda")
            st = torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream()
            with torch.cuda.stream(st):
                dist.all_gather_into_tensor(recv, send, group=group)
            return 0
        except Exception:                                  # (an exception must not unwind through the C frames)
            import traceback
            traceback.print_exc()
            return -1

    cb = lsd.ALL_GATHER_FN(_all_gather)
    comm = lsd.lsd_comm(dist.get_rank(group), dist.get_world_size(group), cb, None)
    comm._keepalive = cb
    return comm


def gather_lines_abi(ctx, comm, lines_i64, counts, n_total, cap_rows, stream=None):
    """The C ABI's hand-off (lsd_gather_lines) on torch tensors: lines_i64 [n_local, max_lines, 10] int64 and counts [n_local] int32 of
    this rank's shard -> (counts_all int32 [world, per + 2], slabs int64 [world, cap_rows, 10]) on every rank, enqueued on `stream`
    (a raw hipStream_t / torch stream .cuda_stream; None: the current torch stream), no host synchronisation.  Same packing as
    pack_lines() above; lsd.gather_unpack turns host copies of the two arrays into offsets + lines in global image order."""
    import importlib
    lsd = importlib.import_module(__package__)
    per, _ = lsd.gather_layout(n_total, comm.world)
    dev = lines_i64.device
    counts_all = torch.empty((comm.world, per + 2), dtype=torch.int32, device=dev)
    slabs = torch.empty((comm.world, cap_rows, WORDS_PER_LINE), dtype=torch.int64, device=dev)
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    ctx.gather_lines(comm, lines_i64.data_ptr(), counts.data_ptr(), counts.numel(), lines_i64.shape[1], n_total, cap_rows,
                     counts_all.data_ptr(), slabs.data_ptr(), stream)
    return counts_all, slabs


def lines_to_numpy(lines_i64, line_dtype):
    """int64[total,10] -> structured numpy array with the structLinesInfo fields."""
    a = lines_i64.cpu().numpy()
    return np.ascontiguousarray(a).view(np.uint8).reshape(-1, 80).view(line_dtype).reshape(-1)
