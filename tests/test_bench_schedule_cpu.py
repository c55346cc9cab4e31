"""bench.py's control flow issues the same collectives, in the same order, on every rank.

Round 1's two-rank rehearsal hung because rank 0 alone ran the traced steps: its gradient all-reduces paired with the other
rank's final barrier.  ``bench.run_schedule`` is the whole control flow; here it is driven with recording fakes for rank 0
and rank 1 and the recorded collective sequences must be identical."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class FakeComm:
    def __init__(self, log):
        self.log = log

    def sync(self):
        self.log.append("barrier")

    def max(self, x):
        self.log.append("allreduce_max")
        return x


@pytest.mark.parametrize("warmup,steps,trace_steps", [(2, 3, 5), (0, 1, 0), (1, 4, 2)])
def test_every_rank_issues_the_same_collectives(bench, warmup, steps, trace_steps):
    logs = {}
    traced_on = {}
    for rank in (0, 1, 3):
        log = []
        seen = []

        def step(i, log=log, seen=seen):
            seen.append(i)
            log.append("grad_allreduce")          # every training step contains the gradient collectives

        def traced(step_j, n, log=log):
            traced_on[rank] = n
            for j in range(n):
                step_j(j)
            return [{"kernel": "k", "us_per_step": 1.0}]

        elapsed, rows = bench.run_schedule(step, FakeComm(log), rank, warmup, steps, trace_steps, traced)
        logs[rank] = log
        assert seen == list(range(warmup + steps + trace_steps))      # step indices (= batch rotation) identical on all ranks
        assert (rows is not None) == (rank == 0 and trace_steps > 0)
        assert elapsed >= 0.0
    assert logs[0] == logs[1] == logs[3]
    assert logs[0].count("grad_allreduce") == warmup + steps + trace_steps
    assert logs[0].count("barrier") == 2 and logs[0].count("allreduce_max") == 1
    assert set(traced_on) <= {0}
