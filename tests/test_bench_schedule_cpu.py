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


def test_bare_gpus_n_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 4` with no launcher around it (WORLD_SIZE unset) starts `<launcher> --nnodes=1 --nproc-per-node 4
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 4 ...` as a CHILD process before it touches a GPU, passes rank 0's JSON line
    through on stdout and exits with the child's code -- what Lightning's strategy="ddp" does for the reference
    (runner/run_experiment.py:437-454).  The launcher here is a recording fake (TF_BENCH_LAUNCHER)."""
    import json
    import subprocess
    fake = tmp_path / "fake_launcher.py"
    fake.write_text(
        "import json, os, sys\n"
        "json.dump({'argv': sys.argv[1:], 'ipc': os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'), 'world': os.environ.get('WORLD_SIZE')},"
        f" open({str(tmp_path / 'seen.json')!r}, 'w'))\n"
        "print(json.dumps({'metric': 'fake', 'n_gpus': 4}))\n"
        "sys.exit(int(os.environ.get('FAKE_RC', '0')))\n")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["TF_BENCH_LAUNCHER"] = f"{sys.executable} {fake}"
    for rc in (0, 3):
        env["FAKE_RC"] = str(rc)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "2"], env=env,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == rc, r.stderr[-2000:]
        assert json.loads(r.stdout.strip().splitlines()[-1]) == {"metric": "fake", "n_gpus": 4}
        seen = json.load(open(tmp_path / "seen.json"))
        a = seen["argv"]
        assert a[:3] == ["--nnodes=1", "--nproc-per-node", "4"] and a[3:5] == ["--master-addr", "127.0.0.1"] and a[5] == "--master-port"
        assert a[7].endswith("bench.py") and a[8:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]
        assert seen["ipc"] == "0" and seen["world"] is None


def test_a_probe_that_never_returns_cannot_take_the_bench_line_with_it(tmp_path):
    """bench.probe_under_timer: the own-RCCL probe of the N > 1 line runs last and under a timer.  A probe that hangs (a communicator
    that never forms) must leave rank 0's JSON line printed -- without the probe's part -- and exit code 0; a probe that returns is
    merged into the line."""
    import json
    import subprocess
    script = tmp_path / "probe.py"
    script.write_text(
        "import importlib.util, json, os, sys, time\n"
        f"spec = importlib.util.spec_from_file_location('b', {os.path.join(ROOT, 'bench.py')!r}); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "res = {'metric': 'm', 'value': 1.0, 'allreduce': {'backend': 'nccl', 'group_world': 8}}\n"
        "if sys.argv[1] == 'hang':\n"
        "    b.probe_under_timer(res, lambda: time.sleep(3600), 0, timeout_s=1.0)\n"
        "else:\n"
        "    b.probe_under_timer(res, lambda: {'rccl': {'world': 8, 'calls': 11}}, 0, timeout_s=30.0)\n"
        "print(json.dumps(res))\n")
    r = subprocess.run([sys.executable, str(script), "hang"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] == 1.0 and line["allreduce"]["group_world"] == 8 and "error" in line["allreduce"]["rccl"]
    r = subprocess.run([sys.executable, str(script), "ok"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["allreduce"]["rccl"] == {"world": 8, "calls": 11} and line["allreduce"]["backend"] == "nccl"


def test_an_extra_leg_that_hangs_or_raises_cannot_take_the_bench_line_with_it(tmp_path):
    """bench.extra_under_timer (the real module under data parallelism as an extra of the N > 1 line): a hang costs the entry -- the line
    is printed without it, exit code 0 --, an exception becomes the entry's error text, a result is merged under its key."""
    import json
    import subprocess
    script = tmp_path / "extra.py"
    script.write_text(
        "import importlib.util, json, os, sys, time\n"
        f"spec = importlib.util.spec_from_file_location('b', {os.path.join(ROOT, 'bench.py')!r}); b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)\n"
        "res = {'metric': 'm', 'value': 2.0}\n"
        "def boom():\n"
        "    raise SystemExit('non-finite loss')\n"
        "fn = {'hang': lambda: time.sleep(3600), 'raise': boom, 'ok': lambda: {'ms_per_step': 5.5}}[sys.argv[1]]\n"
        "b.extra_under_timer(res, 'wrapper_b4_real_dp', fn, 0, 1.0 if sys.argv[1] == 'hang' else 30.0)\n"
        "print(json.dumps(res))\n")
    want = {"hang": lambda e: "did not return" in e["error"], "raise": lambda e: "non-finite" in e["error"], "ok": lambda e: e == {"ms_per_step": 5.5}}
    for mode, check in want.items():
        r = subprocess.run([sys.executable, str(script), mode], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, (mode, r.stderr[-2000:])
        line = json.loads(r.stdout.strip().splitlines()[-1])
        assert line["value"] == 2.0 and check(line["wrapper_b4_real_dp"]), (mode, line)


def _live_sections_worker(rank, world, port, out, bench_path, skew):
    import importlib.util
    import json
    import time

    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("bench_under_test", bench_path)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    seq = []                                             # the collectives this rank issued, in order
    real_ar, real_bar = dist.all_reduce, dist.barrier

    def ar(t, *a, **k):
        seq.append(("all_reduce", int(t.numel()), str(k.get("op", a[0] if a else "SUM"))))
        return real_ar(t, *a, **k)

    def bar(*a, **k):
        seq.append(("barrier",))
        return real_bar(*a, **k)
    b.dist.all_reduce, b.dist.barrier = ar, bar
    if skew:
        time.sleep(0.05 * ((rank * 5) % world))          # ranks arrive at every section at different times

    def busbw():                                         # stand-in for allreduce_busbw: the same shape of work -- barrier, collectives, barrier, max
        g = torch.ones(1000)
        dist.barrier()
        for _ in range(3):
            dist.all_reduce(g)
        dist.barrier()
        t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return dict(bytes=4000, collectives_per_step=4, ms=1.0, algbw_gbs=4.0, busbw_gbs=round(4.0 * 2 * (world - 1) / world, 2))

    def leg(ms):
        def run():                                       # a leg = warm-up, barrier, timed steps with gradient collectives, barrier, max over ranks
            g = torch.ones(64)
            dist.barrier()
            for _ in range(2):
                dist.all_reduce(g)
            dist.barrier()
            t = torch.tensor([ms * (1 + 0.01 * rank)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return {"samples_s": round(4 * world / (float(t) * 1e-3), 1), "ms_per_step": round(float(t), 3), "batch_per_gpu": 32 // world,
                    "reducer": "OrderedRangeReducer"}
        return run

    result = {"metric": "m", "value": 1.0, "n_gpus": world}
    b.live_sections(result, world=world, rank=rank, checksum=lambda: torch.tensor(42.5, dtype=torch.float64), busbw=busbw,
                    strong=leg(1.5), wrapper_dp=leg(5.0), probe=lambda: {"rccl": {"world": world, "calls": 1, "matches_process_group": True}})
    seqs = [None] * world
    dist.all_gather_object(seqs, seq)
    if rank == 0:
        json.dump({"result": result, "seqs": seqs}, open(out, "w"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("skew", [False, True])
def test_the_n8_line_carries_its_sections_and_every_rank_walks_them_alike(tmp_path, skew):
    """bench.live_sections with EIGHT gloo ranks (what `python bench.py --gpus 8` runs after the headline): rank 0's line carries
    ``rank_sync`` (identical on 8 ranks), ``allreduce.busbw_gbs`` + backend + group size, ``strong`` (global batch 32 = 4 per rank,
    run_experiment.py:373-374), ``wrapper_b4_real_dp`` and ``allreduce.rccl``; all eight ranks issue the SAME sequence of collectives,
    also when they reach the sections at different times.  The measurements are stand-ins (no GPU here) that issue real collectives in the
    shape the real ones do; the order, the keys and the timers are the product code."""
    import json
    import torch.multiprocessing as mp
    out = str(tmp_path / "n8.json")
    from test_ddp_cpu import _free_port
    mp.spawn(_live_sections_worker, args=(8, _free_port(), out, os.path.join(ROOT, "bench.py"), skew), nprocs=8, join=True)
    got = json.load(open(out))
    r = got["result"]
    assert r["rank_sync"] == {"parameter_checksum": 42.5, "identical_on_ranks": 8}
    assert r["allreduce"]["busbw_gbs"] == 7.0 and r["allreduce"]["backend"] == "gloo" and r["allreduce"]["group_world"] == 8
    assert r["allreduce"]["rccl"] == {"world": 8, "calls": 1, "matches_process_group": True}
    assert r["strong"]["global_batch"] == 32 and r["strong"]["batch_per_gpu"] == 4 and r["strong"]["samples_s"] > 0
    assert "error" not in r["wrapper_b4_real_dp"] and r["wrapper_b4_real_dp"]["reducer"] == "OrderedRangeReducer"
    assert all(s == got["seqs"][0] for s in got["seqs"]) and len(got["seqs"][0]) > 10


def _mismatch_worker(rank, world, port, bench_path):
    import importlib.util

    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = importlib.util.spec_from_file_location("bench_under_test", bench_path)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    try:
        b.live_sections({}, world=world, rank=rank, checksum=lambda: torch.tensor(1.0 + (rank == 2), dtype=torch.float64),
                        busbw=lambda: {})
    except AssertionError:
        dist.destroy_process_group()
        return
    raise SystemExit("ranks with different parameters passed the check")


def test_ranks_that_drifted_apart_fail_the_line_on_every_rank():
    import torch.multiprocessing as mp
    from test_ddp_cpu import _free_port
    mp.spawn(_mismatch_worker, args=(4, _free_port(), os.path.join(ROOT, "bench.py")), nprocs=4, join=True)
