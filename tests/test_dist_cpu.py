"""The N>1 path on CPU: world_size-2 gloo run of the shard assignment and the line-list gather
(linesegmentdetector-slam_amd/dist.py) on synthetic per-rank line buffers, plus the bench's batch generator."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fake_shard(lo, hi, max_lines):
    """Deterministic fake results for global images lo..hi-1: image g has (g*7)%5 + (g%3) lines."""
    n = hi - lo
    lines = torch.zeros((n, max_lines, 10), dtype=torch.int64)
    counts = torch.zeros(n, dtype=torch.int32)
    for j in range(n):
        g = lo + j
        c = (g * 7) % 5 + (g % 3)
        counts[j] = c
        for k in range(c):
            lines[j, k] = torch.arange(10, dtype=torch.int64) + 1000 * g + 10 * k
        lines[j, c:] = -1                                         # garbage beyond the count must never travel
    return lines, counts


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    ldist = importlib.import_module("linesegmentdetector-slam_amd.dist")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = ldist.shard_range(n_total, world, rank)
    lines, counts = _fake_shard(lo, hi, 8)
    offsets, out = ldist.gather_line_lists(lines, counts, n_total, dst=0)
    if rank == 0:
        q.put((offsets.numpy(), out.numpy()))
    else:
        assert offsets is None and out is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [7, 8, 1])
def test_gather_line_lists_gloo_world2(n_total):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500) + n_total
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    offsets, out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    exp_lines, exp_counts = _fake_shard(0, n_total, 8)
    exp_off = np.concatenate([[0], np.cumsum(exp_counts.numpy())])
    assert np.array_equal(offsets, exp_off)
    dense = np.concatenate([exp_lines[g, :exp_counts[g]].numpy() for g in range(n_total)] + [np.zeros((0, 10), np.int64)])
    assert np.array_equal(out, dense)
    assert (out >= 0).all()


def _worker_padded(rank, world, port, n_total, cap_rows, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    ldist = importlib.import_module("linesegmentdetector-slam_amd.dist")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = ldist.shard_range(n_total, world, rank)
    lines, counts = _fake_shard(lo, hi, 8)
    cnts, slabs, over = ldist.gather_line_lists(lines, counts, n_total, dst=0, cap_rows=cap_rows, dense=False)
    if rank == 0:
        q.put((cnts.numpy(), slabs.numpy(), over.numpy()))
    else:
        assert cnts is None and slabs is None and over is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("cap_rows,expect_over", [(None, False), (16, False), (6, True)])
def test_gather_line_lists_padded_form_world2(cap_rows, expect_over):
    """The per-step form: fixed-capacity slabs, no host synchronisation; an undersized slab is flagged, never silent."""
    world, n_total = 2, 7
    ldist = importlib.import_module("linesegmentdetector-slam_amd.dist")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29100 + (os.getpid() % 500) + (cap_rows or 0)
    procs = [ctx.Process(target=_worker_padded, args=(r, world, port, n_total, cap_rows, q)) for r in range(world)]
    for p in procs:
        p.start()
    cnts, slabs, over = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    exp_lines, exp_counts = _fake_shard(0, n_total, 8)
    assert bool(over.any()) == expect_over
    for r in range(world):
        lo, hi = ldist.shard_range(n_total, world, r)
        assert np.array_equal(cnts[r, :hi - lo], exp_counts[lo:hi].numpy())
        dense = np.concatenate([exp_lines[g, :exp_counts[g]].numpy() for g in range(lo, hi)])
        fit = min(len(dense), slabs.shape[1])
        assert np.array_equal(slabs[r, :fit], dense[:fit])          # compacted at the head of the rank's slab, image order
        if not over[r]:
            assert len(dense) <= slabs.shape[1] and not slabs[r, len(dense):].any()


def _pack_like_the_device(lines, counts, per, cap_rows):
    """What lsd_gather_lines' device pack leaves on one rank (include/lsd_hip.h): cpad int32[per + 2] and a slab of cap_rows records
    (10 int64 words each), image-major, rows past cap_rows dropped and flagged."""
    n, max_lines, _ = lines.shape
    c = np.clip(counts, 0, max_lines).astype(np.int32)
    cpad = np.zeros(per + 2, np.int32)
    cpad[:n] = c
    dense = np.concatenate([lines[j, :c[j]] for j in range(n)] + [np.zeros((0, 10), np.int64)])
    slab = np.zeros((cap_rows, 10), np.int64)
    keep = min(len(dense), cap_rows)
    slab[:keep] = dense[:keep]
    cpad[per] = keep
    cpad[per + 1] = 1 if (len(dense) > cap_rows or (counts > max_lines).any() or (counts < 0).any()) else 0
    return cpad, slab


@pytest.mark.parametrize("n_total,world,cap_rows,expect_over", [(7, 2, 64, False), (8, 2, 64, False), (1, 2, 8, False), (13, 3, 64, False), (7, 2, 6, True)])
def test_c_abi_hand_off_host_side_with_a_host_memory_communicator(n_total, world, cap_rows, expect_over, lsdmod):
    """The host side of the C ABI's multi-GPU hand-off (include/lsd_hip.h: lsd_shard_range, lsd_gather_layout, lsd_gather_unpack) for
    world sizes 2 and 3 without a GPU: every rank's packed counts and slab (the documented layout of lsd_gather_lines' device pack)
    go through a host-memory communicator -- an lsd_comm whose all_gather callback copies between the ranks' host buffers, called
    through the same ctypes function-pointer type the library calls -- and lsd_gather_unpack must give the offsets and lines of the
    whole batch in global image order.  An undersized slab is flagged (LSD_ERR_CAPACITY), never silent."""
    import ctypes as C
    ldist = importlib.import_module("linesegmentdetector-slam_amd.dist")
    per, words = lsdmod.gather_layout(n_total, world)
    assert words == world * (per + 2)
    ranks = []
    for r in range(world):
        lo, hi = lsdmod.shard_range(n_total, world, r)
        assert (lo, hi) == ldist.shard_range(n_total, world, r)
        lines, counts = _fake_shard(lo, hi, 8)
        ranks.append(_pack_like_the_device(lines.numpy(), counts.numpy(), per, cap_rows))
    # the host-memory communicator: rank r's callback finds every rank's send buffer in a table keyed by the collective's number
    table = {}
    recv = {r: [np.zeros((world, per + 2), np.int32), np.zeros((world, cap_rows, 10), np.int64)] for r in range(world)}
    def make_comm(r):
        calls = [0]
        def all_gather(user, d_send, d_recv, nbytes, stream):
            k = calls[0]; calls[0] += 1
            for q in range(world):
                src = table[(k, q)]
                assert src.nbytes == nbytes
                C.memmove(d_recv + q * nbytes, src.ctypes.data, nbytes)
            return 0
        cb = lsdmod.ALL_GATHER_FN(all_gather)
        comm = lsdmod.lsd_comm(r, world, cb, None)
        comm._keep = cb
        return comm
    for r, (cpad, slab) in enumerate(ranks):
        table[(0, r)] = cpad; table[(1, r)] = slab
    for r in range(world):
        comm = make_comm(r)
        assert comm.all_gather(None, ranks[r][0].ctypes.data, recv[r][0].ctypes.data, ranks[r][0].nbytes, None) == 0
        assert comm.all_gather(None, ranks[r][1].ctypes.data, recv[r][1].ctypes.data, ranks[r][1].nbytes, None) == 0
    exp_lines, exp_counts = _fake_shard(0, n_total, 8)
    for r in range(world):                                   # every rank ends up with the same two arrays, and unpacks the same batch
        assert np.array_equal(recv[r][0], recv[0][0]) and np.array_equal(recv[r][1], recv[0][1])
        if expect_over:
            with pytest.raises(lsdmod.LsdError) as e:
                lsdmod.gather_unpack(recv[r][0], recv[r][1], n_total, world, cap_rows)
            assert e.value.status == lsdmod.LSD_ERR_CAPACITY
            continue
        offs, lines = lsdmod.gather_unpack(recv[r][0], recv[r][1], n_total, world, cap_rows)
        assert np.array_equal(offs, np.concatenate([[0], np.cumsum(exp_counts.numpy())]))
        dense = np.concatenate([exp_lines[g, :exp_counts[g]].numpy() for g in range(n_total)] + [np.zeros((0, 10), np.int64)])
        assert lines.view(np.int64).reshape(-1, 10).tobytes() == dense.tobytes()


def test_shard_range_partitions():
    ldist = importlib.import_module("linesegmentdetector-slam_amd.dist")
    for n in (1, 5, 8, 512, 513):
        for world in (1, 2, 3, 4, 8):
            cover = []
            for r in range(world):
                lo, hi = ldist.shard_range(n, world, r)
                assert 0 <= lo <= hi <= n
                cover += list(range(lo, hi))
            assert cover == list(range(n))
    assert ldist.shard_range(512, 8, 3) == (192, 256)


def test_compact_and_numpy_view(lsdmod):
    ldist = importlib.import_module("linesegmentdetector-slam_amd.dist")
    rec = np.zeros(3, lsdmod.LINE_DTYPE)
    rec["x1"] = [1.5, 2.5, 3.5]; rec["k"] = [np.inf, -0.0, np.nan]; rec["orient"] = [1, -1, 1]
    raw = torch.from_numpy(rec.view(np.int64).reshape(3, 10).copy())
    lines = torch.zeros((2, 4, 10), dtype=torch.int64)
    lines[0, :2] = raw[:2]; lines[1, :1] = raw[2:]
    dense, c = ldist.compact_lines(lines, torch.tensor([2, 1], dtype=torch.int32))
    back = ldist.lines_to_numpy(dense, lsdmod.LINE_DTYPE)
    assert back.tobytes() == rec.tobytes()                        # byte-exact, NaN/inf/-0.0 included
    dense, c = ldist.compact_lines(lines, torch.tensor([9, 0], dtype=torch.int32))   # overflowed image is clamped
    assert dense.shape[0] == 4 and c.tolist() == [4, 0]


def test_bench_batch_generator(maps):
    sys.path.insert(0, ROOT)
    import bench
    b = bench.make_batch(maps, 5, 512)
    assert b.shape == (5, 512, 512) and b.dtype == np.uint8
    assert np.array_equal(b[0], np.tile(maps["aisle1"], (1, 1))[:512, :512])     # image 0: unshifted, unflipped aisle1 tile
    assert np.array_equal(bench.make_image(maps, 3, 512), b[3])                   # seeded: reproducible
    assert np.array_equal(bench.make_batch(maps, 2, 512, first=3)[0], b[3])       # rank shards see the same images
    assert set(np.unique(b)) <= {0, 1, 255}
    flip = bench.make_image(maps, 4, 512)                                         # i//4%4 == 1 -> left-right flip
    assert not np.array_equal(flip, bench.make_image(maps, 0, 512))


def test_shard_balanced_deals_by_cost():
    """lsd_shard_balanced (C ABI, no GPU needed): a permutation; rank r's contiguous shard of it (lsd_shard_range) ascending inside, and
    the shards' costs far closer to each other than those of the plain contiguous split when the costs are skewed."""
    import ctypes as C
    import importlib
    lsd = importlib.import_module("linesegmentdetector-slam_amd")
    L = lsd.load_library()
    rng = np.random.default_rng(1)
    costs = (rng.pareto(2.0, 512) * 1e7 + 2e7).astype(np.int64)
    costs[:40] *= 4                                               # (the heavy images at the front: what the bench batch looks like)
    for world in (1, 2, 3, 4, 8):
        perm = lsd.shard_balanced(costs, world)
        assert sorted(perm.tolist()) == list(range(512))
        bal, con = [], []
        for r in range(world):
            lo, hi = C.c_int(), C.c_int()
            L.lsd_shard_range(512, world, r, C.byref(lo), C.byref(hi))
            assert np.all(np.diff(perm[lo.value:hi.value]) > 0)
            bal.append(int(costs[perm[lo.value:hi.value]].sum())); con.append(int(costs[lo.value:hi.value].sum()))
        assert max(bal) <= max(1.15 * np.mean(bal), np.mean(bal) + costs.max())      # (the greedy deal's guarantee: mean + the largest single cost)
        if world >= 4:
            assert max(bal) < max(con)
    with pytest.raises(lsd.LsdError):
        lsd.shard_balanced(np.zeros(0, np.int64), 2)
