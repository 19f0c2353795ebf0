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
