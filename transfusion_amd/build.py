"""Builds libtfusion_hip.so (gfx950 only) in-tree with hipcc.  Used by __graft_entry__.build() and
`python -m transfusion_amd.build`.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libtfusion_hip.so")
SOURCES = ["gemm_bf16.hip", "wgrad_multi.hip", "attn_bf16.hip", "attn_x3.hip", "rowops.hip", "heads.hip", "comm.hip", "tf_api.hip"]
HEADERS = ["tf_common.h", "tf_kernels.h", "attn_common.h", os.path.join("..", "..", "include", "tfusion.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


EXP_DIR = os.path.join(os.path.dirname(HERE), "build", "variants", "exp")


def build_lib(force: bool = False, verbose: bool = True, experiments: bool = False) -> str:
    """experiments=True: a SECOND library with -DTF_EXPERIMENTS (the TF_* environment switches of the kernels' launch planners and
    the ablation blocks are compiled in) under build/variants/exp/, selected with TFUSION_LIB for A/B runs.  The shipped library reads
    no experiment switch."""
    lib_dir = EXP_DIR if experiments else LIB_DIR
    lib_path = os.path.join(lib_dir, "libtfusion_hip.so")
    flags = FLAGS + (["-DTF_EXPERIMENTS"] if experiments else [])
    os.makedirs(lib_dir, exist_ok=True)
    obj_dir = os.path.join(EXP_DIR, "_obj") if experiments else os.path.join(HERE, "csrc", "_obj")
    os.makedirs(obj_dir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return src

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for done in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[transfusion_amd.build] compiled {os.path.basename(done)}", file=sys.stderr)
    objs = [os.path.join(obj_dir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(lib_path, objs + [os.path.join(CSRC, HEADERS[-1])]):
        # exported = exactly the entries include/tfusion.h declares; the launchers the translation units call each other through
        # (tf_launch_*, tf_tu_*) stay internal to the library
        from transfusion_amd import _lib
        vmap = os.path.join(obj_dir, "exports.map")
        with open(vmap, "w") as f:
            f.write("{\n  global:\n" + "".join(f"    {name};\n" for name in _lib.FUNCTIONS) + "  local:\n    *;\n};\n")
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", f"-Wl,--version-script={vmap}", "-o", lib_path] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[transfusion_amd.build] linked {lib_path}", file=sys.stderr)
    return lib_path


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, experiments="--exp" in sys.argv))
