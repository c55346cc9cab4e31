"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/tfusion.h declares,
and the ctypes mirrors generated from the header have the sizes a C compiler gives them.  No compute calls."""
import ctypes as C
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from transfusion_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build_lib()
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    from transfusion_amd import _lib
    assert len(_lib.FUNCTIONS) >= 28
    for f in _lib.FUNCTIONS:
        assert hasattr(lib, f), f
    assert lib.tf_version() == _lib.CONSTS["TF_ABI_VERSION"]


def test_exports_nothing_the_header_does_not_declare(lib):
    """The dynamic symbol table of the library holds the header's entries and no other function of ours (the tf_launch_* / tf_tu_*
    launchers the translation units call each other through are internal: a consumer cannot bind what the ABI does not promise)."""
    from transfusion_amd import _lib
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-2] in ("T", "t", "W")}
    ours = {s for s in exported if s.startswith("tf_") or s.startswith("_Z")}
    assert ours == set(_lib.FUNCTIONS), sorted(ours ^ set(_lib.FUNCTIONS))[:20]


def test_struct_mirrors_match_c_layout(lib, tmp_path):
    from transfusion_amd import _lib
    names = sorted(_lib.STRUCTS)
    src = tmp_path / "sz.c"
    src.write_text('#include "tfusion.h"\n#include <stdio.h>\nint main(void){' +
                   "".join(f'printf("{n} %zu\\n", sizeof({n}));' for n in names) + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    for n in names:
        assert int(out[n]) == C.sizeof(_lib.STRUCTS[n]), n


def test_plan_and_dropout_helpers_are_host_side(lib):
    from transfusion_amd import _lib
    plan = _lib.TfEncoderPlan()
    assert lib.tf_encoder_plan(32, 196, 512, 768, 4, 4, 1536, C.byref(plan)) == 0
    assert (plan.hd, plan.hdp, plan.dp, plan.ffp, plan.ldq, plan.S, plan.M) == (192, 192, 768, 1536, 2304, 708, 22656)
    plan0_w, plan0_k = plan.wpack_bytes, plan.work_bytes
    assert lib.tf_encoder_plan(2, 196, 128, 712, 4, 4, 1424, C.byref(plan)) == 0
    assert (plan.hd, plan.hdp, plan.dp, plan.ffp) == (178, 192, 768, 1472)
    assert lib.tf_encoder_plan(2, 196, 128, 770, 4, 4, 1424, C.byref(plan)) != 0     # d % H != 0
    assert b"tf_encoder_plan" in lib.tf_last_error()
    # fp32-accuracy mode: every bf16 tensor gains a lo plane -- the workspace and the weight shadows roughly double
    e = _lib.TfEncoderDesc()
    e.B, e.Nv, e.Nl, e.d, e.H, e.L, e.ff = 32, 196, 512, 768, 4, 4, 1536
    p0, p1 = _lib.TfEncoderPlan(), _lib.TfEncoderPlan()
    assert lib.tf_encoder_plan_ex(C.addressof(e), C.addressof(p0)) == 0
    e.precision = 1
    assert lib.tf_encoder_plan_ex(C.addressof(e), C.addressof(p1)) == 0
    assert (p0.wpack_bytes, p0.work_bytes) == (plan0_w, plan0_k) and 1.8 * p0.work_bytes < p1.work_bytes < 2.2 * p0.work_bytes      # plane pairs of everything; FOUR planes of attention workspace (dS and Pd) against one
    assert 1.5 * p0.wpack_bytes < p1.wpack_bytes < 2.0 * p0.wpack_bytes and p1.hdp == 192      # the fp8 shadows and fp32 biases have no lo plane
    e.d, e.H, e.ff = 1024, 4, 2048                       # head dim 256: bf16 only
    assert lib.tf_encoder_plan_ex(C.addressof(e), C.addressof(p1)) != 0
    e.precision = 2
    assert lib.tf_encoder_plan_ex(C.addressof(e), C.addressof(p1)) != 0
    assert lib.tf_drop_threshold(0.0) == 0
    assert abs(lib.tf_drop_threshold(0.15) / 2**16 - 0.15) < 1e-5
    assert abs(lib.tf_drop_scale(0.15) - 1 / 0.85) < 1e-4
    assert lib.tf_drop_key(42, 1) != lib.tf_drop_key(42, 2) != lib.tf_drop_key(43, 2)


def test_no_cpu_fallback():
    import torch
    from transfusion_amd import _lib, ops
    with pytest.raises(_lib.TfError):
        ops.linear(torch.randn(4, 8), torch.randn(8, 8))


def test_integration_md_stub_matches_header(lib, tmp_path):
    """The ctypes stub INTEGRATION.md tells a maintainer to paste must be the struct the library reads: extract the class from the
    document, and compare its fields with the header-generated mirror and its size with what gcc gives the C struct."""
    import re
    from transfusion_amd import _lib
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    src = next(b for b in blocks if "class TfGemmArgs(ctypes.Structure)" in b)
    cls_src = src[src.index("class TfGemmArgs"):src.index("def linear_bf16")]
    ns = {"ctypes": C}
    exec(cls_src, ns)
    stub = ns["TfGemmArgs"]
    mirror = _lib.STRUCTS["TfGemmArgs"]
    assert [n for n, _ in stub._fields_] == [n for n, _ in mirror._fields_]
    assert [C.sizeof(t) for _, t in stub._fields_] == [C.sizeof(t) for _, t in mirror._fields_]
    assert C.sizeof(stub) == C.sizeof(mirror)
    for n, _ in stub._fields_:
        assert getattr(stub, n).offset == getattr(mirror, n).offset, n
    csrc = tmp_path / "g.c"
    csrc.write_text('#include "tfusion.h"\n#include <stdio.h>\nint main(void){printf("%zu\\n", sizeof(TfGemmArgs));return 0;}\n')
    exe = tmp_path / "g"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(csrc), "-o", str(exe)])
    assert int(subprocess.check_output([str(exe)], text=True)) == C.sizeof(stub)


def test_comm_entries_validate_arguments_without_a_gpu(lib):
    """tf_comm_* / tf_allreduce_bucket: argument errors come back as -1 with a message, and drawing a unique id
    either works (RCCL resident: torch ships it) or reports TF_ERR_NO_RCCL -- no compute, no device needed."""
    import ctypes as C
    from transfusion_amd import _lib as L
    lib.tf_comm_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int]
    lib.tf_allreduce_bucket.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]
    handle = C.c_void_p()
    ident = bytes(L.CONSTS["TF_COMM_ID_BYTES"])
    assert lib.tf_comm_create(C.byref(handle), ident, 2, 2) == -1          # rank outside [0, world)
    assert b"tf_comm_create" in lib.tf_last_error()
    assert lib.tf_comm_create(C.byref(handle), None, 1, 0) == -1           # no id
    assert lib.tf_allreduce_bucket(None, None, 16, None) == -1
    assert lib.tf_comm_destroy(None) == -1
    buf = C.create_string_buffer(L.CONSTS["TF_COMM_ID_BYTES"])
    rc = lib.tf_comm_unique_id(buf)
    assert rc in (0, L.CONSTS["TF_ERR_NO_RCCL"])
    if rc == 0:
        assert any(buf.raw)                                                # an id was drawn
    else:
        assert b"librccl" in lib.tf_last_error()


def test_shipped_library_reads_no_experiment_switch(lib):
    """The TF_* environment switches of the launch planners and the ablation blocks exist in EXPERIMENTS builds only
    (-DTF_EXPERIMENTS: `python -m transfusion_amd.build --exp`, tools/build_variant.sh): the shipped library does not reference
    getenv and none of the switch names is in the binary, so no environment variable can change what training computes."""
    import re
    from transfusion_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    names = set(m.group(0).decode() for m in re.finditer(rb"TF_[A-Z][A-Z0-9_]{3,}", blob))
    assert not names, sorted(names)
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", _lib.LIB_PATH], text=True)
    assert " getenv" not in syms
    # the switches are still in the sources (behind the macro), i.e. this test would notice one that bypasses it
    src = "".join(open(os.path.join(ROOT, "transfusion_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "transfusion_amd", "csrc"))
                  if f.endswith((".hip", ".h")))
    assert "TF_ENV_INT(\"TF_" in src
    assert not re.search(r"[^_a-z]getenv\(\"TF_", src)


def test_dense_rows_warning_is_said_once():
    import warnings
    from transfusion_amd.modeling.cross_fusion.ego_fusion import cross_f_box_layers as m
    m._warned_dense_rows = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        m._warn_dense_rows_once()
        m._warn_dense_rows_once()
    assert len(w) == 1 and "lang_valid_rows" in str(w[0].message)
