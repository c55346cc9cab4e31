"""What hipcc did with the kernels, checked on the CPU (no GPU needed): the gfx950 code objects are pulled out of
libtfusion_hip.so (transfusion_amd/codeobj.py: .hip_fatbin -> __CLANG_OFFLOAD_BUNDLE__ -> ELF -> NT_AMDGPU_METADATA) and their
``.vgpr_spill_count`` / ``.private_segment_fixed_size`` are asserted.

A VGPR spill is not just lost speed here.  Several kernels issue LDS-DMA rings behind COUNTED ``s_waitcnt vmcnt(N)`` and transposed LDS
reads from inline asm whose destination registers are live before their ``lgkmcnt`` wait (tf_common.h ``tr_read_asm``): a compiler
spill reload inside such a loop waits ``vmcnt(0)`` (drains the ring) and a spilled asm destination is stored before its data has
arrived.  So: every kernel of the default paths (head dims <= 192, all GEMM / weight-gradient / row kernels) must be spill-free and
scratch-free; the few wide-head instantiations that still spill are listed with their CURRENT counts -- a regression shows up as a
count above the listed one, an improvement as a stale entry (both fail)."""
import os
import re

import pytest

from transfusion_amd import build as tb
from transfusion_amd import codeobj

# kernel (demangled, without parameter list) -> VGPRs spilled today.  hd 224 / 256 only, and only FALLBACK forms: the one-wave-per-SIMD
# 32-row dK / dV kernel keeps dK^T + dV^T + K + V fragments resident and overflows even the 512-register file; the encoder runtime
# no longer runs it (with its dS workspace the wide heads take the spill-free two-pass 16-row kernels) -- it serves stand-alone
# tf_attn_bwd calls without a workspace and cross attention.  The fp32-accuracy dQ kernel at 224 (d = 896 in run.precision 32) is the one
# training-path entry left (31 VGPRs, prologue fragments; DESIGN.md "Register audit").
ALLOWED_SPILLS = {
    "attn_bwd_dkv_kernel<224, true>": 48,
    "attn_bwd_dkv_kernel<256, false>": 46,
    "attn_bwd_dkv_kernel<256, true>": 89,
    "attn_bwd_dq_x3_kernel<224>": 31,
}
COUNT_SLACK = 4        # compiler noise between builds of unrelated edits


@pytest.fixture(scope="module")
def table():
    if not os.path.exists(tb.LIB_PATH):
        tb.build_lib()
    return codeobj.kernel_table(tb.LIB_PATH)


def _head_dim(name):
    m = re.match(r"attn_\w+_kernel<(\d+)", name)
    return int(m.group(1)) if m else None


def test_library_holds_the_expected_kernels(table):
    assert len(table) > 150
    for must in ("gemm_nt_big_kernel<1, 9, false, false, 2>", "wgrad_tn2_kernel<false, 3>", "attn_fwd_kernel<192, false>",
                 "attn_bwd_dkv16_kernel<192, false, 64, true, 2>", "attn_bwd_dq_ds_kernel<192>", "attn_delta_kernel", "row_map_kernel",
                 "ln_bwd_kernel<2, 8, false, false>", "ln_bwd2_kernel<2, 8, 1, 0, 4>", "ln_bwd2_kernel<2, 8, 2, 1, 4>", "radam_kernel",
                 "wgrad_multi_kernel<false, 3, 1, true>", "wgrad_multi_kernel<true, 3, 1, true>",
                 "attn_bwd_dkv_pair_kernel<192, false>", "attn_bwd_dkv_pair_kernel<192, true>"):
        assert must in table, must
    # every kernel was compiled for wave64 workgroups of at most 1024 threads and declares its registers
    assert all(0 < v["max_flat_workgroup_size"] <= 1024 and v["vgpr_count"] > 0 for v in table.values())


def test_no_spills_outside_the_allow_list(table):
    bad = {k: v for k, v in table.items() if (v["vgpr_spill_count"] or v["private_segment_fixed_size"]) and k not in ALLOWED_SPILLS}
    assert not bad, {k: (v["vgpr_spill_count"], v["private_segment_fixed_size"]) for k, v in bad.items()}


def test_default_paths_are_spill_free(table):
    """Explicitly, whatever the allow-list says: head dims <= 192, and every kernel with hand-placed DMA rings / asm LDS reads."""
    for k, v in table.items():
        hd = _head_dim(k)
        critical = (hd is not None and hd <= 192) or k.startswith(("wgrad_tn", "wgrad_multi", "gemm_nt_big", "gemm_nt_kernel", "ln_bwd"))
        if critical:
            assert v["vgpr_spill_count"] == 0 and v["private_segment_fixed_size"] == 0, (k, v)
            assert k not in ALLOWED_SPILLS, k


def test_allow_list_is_current(table):
    for k, n in ALLOWED_SPILLS.items():
        assert k in table, f"{k} is no longer in the library: drop it from ALLOWED_SPILLS"
        got = table[k]["vgpr_spill_count"]
        assert 0 < got <= n + COUNT_SLACK, f"{k}: {got} spilled VGPRs, the list says {n}"


def test_dead_wide_head_instantiations_stay_out(table):
    """Head dims above 192: the one-pass 16-row kernels (dQ with S / dP recompute; dK + dV together) do not fit 256 registers and are
    never dispatched there (attn_bf16.hip launch_bwd) -- they must not be instantiated either (they were: 70 to 177 spilled VGPRs of dead
    code).  What IS live at 224 / 256 is the two-pass form (dV pass, dK + dS pass) and the thin dQ kernel, spill-free."""
    for k in table:
        hd = _head_dim(k)
        if hd is not None and hd > 192:
            assert not k.startswith("attn_bwd_dq16_kernel"), k
            if k.startswith("attn_bwd_dkv16_kernel"):
                assert k.rstrip(">").endswith((", 0", ", 1")), k                  # WHICH = 0 (dV) or 1 (dK + dS), never the one-pass form
    for hd in (224, 256):
        for k in (f"attn_bwd_dkv16_kernel<{hd}, false, 64, false, 0>", f"attn_bwd_dkv16_kernel<{hd}, false, 64, true, 1>", f"attn_bwd_dq_ds_kernel<{hd}>"):
            assert k in table and table[k]["vgpr_spill_count"] == 0 and table[k]["private_segment_fixed_size"] == 0, k


def test_occupancy_assumptions(table):
    """Register budgets the launch geometry relies on (waves per SIMD = 512 / allocated registers, 8-register granules)."""
    alloc = lambda k: (table[k]["vgpr_count"] + table[k]["agpr_count"] + 7) // 8 * 8
    assert alloc("attn_fwd_kernel<192, false>") <= 256               # two 4-wave workgroups per CU
    assert alloc("attn_bwd_dq_ds_kernel<192>") <= 256
    assert alloc("attn_bwd_dkv16_kernel<192, false, 64, true, 2>") <= 256    # 8 waves = two per SIMD
    assert alloc("wgrad_tn2_kernel<false, 3>") <= 256                # leaves half of every SIMD's file to the backward chain
    assert alloc("wgrad_multi_kernel<false, 3, 1, true>") <= 256     # two 4-wave workgroups per CU
    assert alloc("wgrad_multi_kernel<true, 3, 1, true>") <= 256
    assert alloc("attn_bwd_dkv_pair_kernel<192, false>") <= 256      # the wave pair of a SIMD
    assert alloc("ln_bwd_kernel<2, 8, false, false>") <= 128                # four waves per SIMD at d <= 1024
    assert alloc("ln_bwd2_kernel<2, 8, 1, 0, 4>") <= 128 and alloc("ln_bwd2_kernel<2, 8, 2, 1, 4>") <= 128
    assert alloc("gemm_nt_big_kernel<1, 9, false, false, 1>") <= 256 # the two-workgroups-per-CU form
