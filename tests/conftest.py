import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------------------------------------------------------------
# Collection order of the GPU suite.  All GPU tests share one process, so a device fault ends the run where it happens
# (round 5: test 151 of 347 aborted and 197 results were lost).  The tests that compare the HIP path with the oracle /
# the reference-generated fixtures -- one per row of SURVEY.md section 8 -- therefore run FIRST, the wide kernel-shape
# sweeps and the multi-process rehearsals last: whatever faults in a sweep, every row's parity result is already
# reported.  (tests/test_collection_order_cpu.py asserts the property.)
# ------------------------------------------------------------------------------------------------------------------------
_FILE_RANK = {
    "test_c_abi_consumer.py": 0,        # the plain-C host over the ABI
    "test_gpu_encoder.py": 1,           # a4 - a9, a11: encoder fixtures, outputs + every gradient
    "test_gpu_wrapper.py": 2,           # a1 - a3, a10: level loop, K1, K9 fixtures
    "test_gpu_heads.py": 3,             # f2
    "test_gpu_lm_head.py": 4,           # f3
    "test_gpu_asymmetric.py": 5,        # f4
    "test_gpu_packed.py": 6,            # packed rows against the oracle
    "test_gpu_fp32_mode.py": 7,         # cfg 3
    "test_gpu_grouped.py": 8,
    "test_gpu_train_shadows.py": 9,
    "test_gpu_graph.py": 10,
    "test_gpu_kernels.py": 20,          # kernel-level sweeps (its fixture tests are pulled forward below)
    "test_gpu_wgrad_multi.py": 21,
    "test_gpu_random_shapes.py": 22,
    "test_gpu_ddp.py": 30,              # in-process / two-process rehearsals: communicators, helper threads
    "test_gpu_comm.py": 31,
}
# tests of test_gpu_kernels.py that ARE a scope row's parity test (reference traces / oracle): run with the fixture files
_EARLY_TESTS = {"test_radam_matches_reference_optimizer": 4.5}


def gpu_order_key(path_basename: str, test_name: str):
    base = test_name.split("[")[0]
    if base in _EARLY_TESTS:
        return _EARLY_TESTS[base]
    return _FILE_RANK.get(path_basename, 15)


def pytest_collection_modifyitems(session, config, items):
    def key(ix_item):
        ix, item = ix_item
        if item.get_closest_marker("gpu") is None:
            return (-1, ix)                                   # CPU tests keep their order, in front
        return (gpu_order_key(os.path.basename(str(item.fspath)), item.name), ix)
    items[:] = [it for _, it in sorted(enumerate(items), key=key)]


# ------------------------------------------------------------------------------------------------------------------------
# Fault attribution.  (i) pytest.ini runs with --capture=sys, so what NATIVE code prints on fd 2 -- the HSA runtime's
# "Memory access fault by GPU node ..." line above all -- reaches the log instead of dying in pytest's fd-level capture
# buffer with the process; (ii) the id of every GPU test is written to fd 2 before it starts, so that line (and an abort)
# is preceded by the test that launched the kernel; (iii) every GPU test ends with a device synchronise: an asynchronous
# error surfaces in ITS OWN test, not in the next test's first sync.
# ------------------------------------------------------------------------------------------------------------------------
_LEAK_ENV = ("TF_REHEARSE_COLLECTIVES", "TF_REHEARSE_PHANTOM_PEERS", "TF_TEST_WGRAD_DELAY_US", "TF_TEST_BREAK_EDGE", "TF_FORCE_LAYERWISE")


@pytest.fixture(autouse=True)
def _gpu_fault_attribution(request):
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    if os.environ.get("TF_GPU_TEST_TRACE", "1") != "0":
        try:
            os.write(2, f"[gpu-test] {request.node.nodeid}\n".encode())
        except OSError:
            pass
    env_before = {k: os.environ.get(k) for k in _LEAK_ENV}
    yield
    import torch
    if torch.cuda.is_available():
        try:
            torch.cuda.synchronize()
        except Exception as e:       # noqa: BLE001
            pytest.fail(f"device error surfaced at the end of {request.node.nodeid}: {e}")
        # test-only hooks must not outlive their test: the weight-gradient delay probe (a process-wide word inside the library) and the
        # rehearsal switches of the reducers (environment; the worker scripts get them through their own environment, never this process's)
        ops = sys.modules.get("transfusion_amd.ops")
        if ops is not None and getattr(ops, "_debug_wgrad_delay_us", 0):
            ops.debug_delay_wgrad(0)
            pytest.fail(f"{request.node.nodeid} left ops.debug_delay_wgrad set")
    leaked = [k for k in _LEAK_ENV if os.environ.get(k) != env_before[k]]
    if leaked:
        for k in leaked:
            if env_before[k] is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = env_before[k]
        pytest.fail(f"{request.node.nodeid} changed {leaked} in the test process's environment")


# ------------------------------------------------------------------------------------------------------------------------
# Guard-page operands for the kernel tests (tests/guard_alloc.py): g(tensor) -> the tensor in device memory whose end is
# the last mapped byte.  TF_GUARD_ALLOC=0 hands tensors back on the ordinary allocator (A/B); a runtime without the
# virtual-memory API does the same with a warning (the tests still run, the over-read check is what is lost).
# ------------------------------------------------------------------------------------------------------------------------
class _Guard:
    def __init__(self):
        self.pool = None
        self.why = None
        if os.environ.get("TF_GUARD_ALLOC", "1") == "0":
            self.why = "TF_GUARD_ALLOC=0"
            return
        try:
            from guard_alloc import GuardPool, GuardUnavailable
            try:
                self.pool = GuardPool()
            except GuardUnavailable as e:
                self.why = str(e)
        except Exception as e:       # noqa: BLE001
            self.why = f"{type(e).__name__}: {e}"
        if self.pool is None:
            os.write(2, f"[guard] guard pages unavailable: {self.why} -- kernel tests run on the ordinary allocator\n".encode())

    @property
    def active(self):
        return self.pool is not None

    def __call__(self, t):
        import torch
        if t is None:
            return None
        if self.pool is None:
            return t.to("cuda:0") if isinstance(t, torch.Tensor) and not t.is_cuda else t
        return self.pool.place(t)

    def close(self):
        if self.pool is not None:
            self.pool.close()


@pytest.fixture
def guard():
    g = _Guard()
    yield g
    g.close()
