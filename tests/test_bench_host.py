"""Host-side pieces of bench.py that the GPU-less suite can check."""
import numpy as np

import bench


def _simulate(slots, steps_per_slot, latency, crowded=()):
    """Completion times of a pipeline whose slot j finishes a step every latency[j] ms (slots in `crowded` run 1.5 x slower:
    three streams on a hardware queue instead of two), steps handed out round-robin up front."""
    done = np.zeros(slots * steps_per_slot)
    for j in range(slots):
        per = latency * (1.5 if j in crowded else 1.0)
        for k in range(steps_per_slot):
            done[k * slots + j] = (k + 1) * per
    return done


def test_steady_rate_of_an_even_pipeline():
    done = _simulate(16, 6, 160.0)
    assert abs(bench.steady_rate_ms(done, 16) - 160.0 / 16) < 1e-9


def test_steady_rate_does_not_wait_for_a_crowded_queue():
    # three of 32 slots share a crowded hardware queue: a fixed number of steps per slot lasts 1.5 x as long as the others need,
    # the rate over the window in which every slot still has work barely moves
    even = bench.steady_rate_ms(_simulate(32, 6, 160.0), 32)
    crowded = _simulate(32, 6, 160.0, crowded=(3, 11, 19))
    fixed = crowded.max() / len(crowded)
    rate = bench.steady_rate_ms(crowded, 32)
    assert fixed > 1.45 * even
    assert rate < 1.06 * even


def test_steady_rate_falls_back_on_a_window_that_is_too_short():
    done = _simulate(8, 1, 100.0)
    assert abs(bench.steady_rate_ms(done, 8) - 100.0 / 8) < 1e-9


def test_gpus_n_without_a_launcher_builds_the_drivers_command_line():
    """`python3 bench.py --gpus 2` started like `--gpus 1` (no torch.distributed.run around it, no WORLD_SIZE): bench.py starts the
    ranks itself as a child job before anything touches the GPU; --dry-launch prints the command it would run."""
    import json, os, subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(bench.__file__), "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--scaling", "strong", "--dry-launch"], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    rec = json.loads(p.stdout.strip().splitlines()[-1])
    cmd = rec["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    k = cmd.index(os.path.abspath(bench.__file__))
    assert cmd[k + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1", "--scaling", "strong"]      # the launcher-only flag is gone
    assert rec["env"] == {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "GPU_MAX_HW_QUEUES": "16"}


def test_launch_command_is_what_the_driver_runs():
    cmd = bench.launch_command(["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, port=29511)
    assert cmd[1:] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", "29511",
                       bench.__file__ if bench.__file__.startswith("/") else cmd[10], "--gpus", "8", "--steps", "20", "--warmup", "5"]


def test_steps_in_flight_follow_the_free_hbm():
    slot = 22.2e9
    assert bench.fit_depth(8, slot, 2e9, 280e9) == (8, None)
    d, note = bench.fit_depth(8, slot, 2e9, 100e9)
    assert d == 4 and "4 in flight" in note
    d, note = bench.fit_depth(8, slot, 2e9, 50e9)
    assert d == 2 and note
    assert bench.fit_depth(1, slot, 2e9, 1e9) == (1, None)
    d, _ = bench.fit_depth(8, slot, 2e9, 1e9)
    assert d == 1


def test_pasted_canvas_is_seamless_and_its_oracle_answer_is_pinned(maps, oracle):
    """bench.py's un-tiled 2048^2 workload (`real_maps.pasted2048`): three different reference maps side by side on their own background
    value, nothing wrapped -- the maps do not touch, the canvas holds exactly their cells, and the oracle's answer on it is the one
    recorded when the workload was added (164 lines, 13 563 raster pixels; the GPU test holds the HIP path to the oracle on it)."""
    c = bench.make_pasted(maps, 2048)
    assert c.shape == (2048, 2048) and c.dtype == np.uint8
    a, b, r = maps["f3key"], maps["f4key"], np.rot90(maps["aisle3"])
    assert int((c == 1).sum()) == int((a == 1).sum() + (b == 1).sum() + (r == 1).sum())
    assert int((c == 255).sum()) == int((a == 255).sum() + (b == 255).sum() + (r == 255).sum())
    assert np.array_equal(c[:a.shape[0], :a.shape[1]], a) and np.array_equal(c[990:990 + b.shape[0], :b.shape[1]], b)
    assert not c[a.shape[0]:990, :2048 - r.shape[1]].any() and not c[:, a.shape[1]:2048 - r.shape[1]].any()      # the gaps between the maps are background
    assert bench.make_pasted(maps, 1024) is None
    res = oracle.lsd(c.copy(), want_lineim=True)
    assert len(res["lines"]) == 164 and int((res["lineIm"] != 0).sum()) == 13563


def test_gpus_n_on_a_node_with_fewer_gpus_says_so_and_starts_nothing():
    """`python3 bench.py --gpus 2` where fewer than two GPUs are visible (this container has none, the GPU box one): the launcher
    refuses with a message and exit code 2 instead of starting ranks that would fail one by one."""
    import os, subprocess, sys
    import pytest, torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has the GPUs: the command would start a real two-rank job")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(bench.__file__), "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 2 and "--gpus 2 but this node shows" in p.stderr, (p.returncode, p.stderr[-500:])
