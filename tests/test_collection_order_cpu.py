"""The GPU suite's collection order (tests/conftest.py): every test that checks the HIP path against the oracle or a reference-generated
fixture is collected BEFORE the wide kernel-shape sweeps and the multi-process rehearsals, so that one faulting kernel test cannot erase a
scope row's parity result (round 5: a fault in test 151 of 347 left 197 tests unreported)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# files whose tests are the parity tests of SURVEY.md section 8's rows (oracle / golden fixtures), and the sweep / rehearsal files
PARITY_FILES = ("test_c_abi_consumer.py", "test_gpu_encoder.py", "test_gpu_wrapper.py", "test_gpu_heads.py", "test_gpu_lm_head.py",
                "test_gpu_asymmetric.py", "test_gpu_packed.py", "test_gpu_fp32_mode.py")
SWEEP_FILES = ("test_gpu_kernels.py", "test_gpu_wgrad_multi.py", "test_gpu_random_shapes.py", "test_gpu_ddp.py", "test_gpu_comm.py")
EARLY_IN_SWEEP_FILES = ("test_radam_matches_reference_optimizer",)


def test_parity_tests_are_collected_before_the_kernel_sweeps():
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "--collect-only", "-q", "-m", "gpu", "-p", "no:cacheprovider"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    ids = [ln.strip() for ln in r.stdout.splitlines() if "::" in ln]
    assert len(ids) > 300, r.stdout[-2000:] + r.stderr[-2000:]
    fname = lambda i: os.path.basename(i.split("::")[0])
    tname = lambda i: i.split("::")[1].split("[")[0]
    first_sweep = min(k for k, i in enumerate(ids) if fname(i) in SWEEP_FILES and tname(i) not in EARLY_IN_SWEEP_FILES)
    late = [i for k, i in enumerate(ids) if k > first_sweep and (fname(i) in PARITY_FILES or tname(i) in EARLY_IN_SWEEP_FILES)]
    assert not late, f"parity tests collected after the first sweep test ({ids[first_sweep]}): {late[:5]}"
    # every parity file is present at all (a renamed file would silently fall to the default rank)
    for f in PARITY_FILES:
        assert any(fname(i) == f for i in ids), f
    # the communicator / multi-process rehearsals close the run
    assert fname(ids[-1]) in ("test_gpu_comm.py", "test_gpu_ddp.py"), ids[-1]
