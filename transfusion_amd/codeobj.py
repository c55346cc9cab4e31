"""What hipcc did with the kernels: per-kernel register / scratch facts read out of the gfx950 code objects inside
libtfusion_hip.so (no GPU needed).

The shared library carries one ``__CLANG_OFFLOAD_BUNDLE__`` per translation unit in its ``.hip_fatbin`` section; each bundle
holds an AMDGPU ELF whose ``NT_AMDGPU_METADATA`` note lists, per kernel, ``.vgpr_count``, ``.agpr_count``, ``.sgpr_count``,
``.vgpr_spill_count``, ``.sgpr_spill_count``, ``.private_segment_fixed_size`` (scratch bytes per lane) and
``.group_segment_fixed_size`` (static LDS).  ``tests/test_codeobj_cpu.py`` asserts on these: a spill inside a kernel that issues
hand-placed LDS-DMA / counted ``s_waitcnt vmcnt`` sequences is not a performance detail there, it is a correctness hazard (a reload's
compiler-inserted ``vmcnt(0)`` drains the ring; a spilled inline-asm destination is read before its ``lgkmcnt`` wait).
"""
from __future__ import annotations

import os
import re
import struct
import subprocess
import tempfile
from typing import Dict, List

LLVM_BIN = os.environ.get("TF_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
_MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
_FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
           "group_segment_fixed_size", "max_flat_workgroup_size")


def _tool(name: str) -> str:
    p = os.path.join(LLVM_BIN, name)
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found (set TF_LLVM_BIN)")
    return p


def extract_code_objects(lib_path: str, out_dir: str, arch: str = "gfx950") -> List[str]:
    """Writes every device ELF of ``arch`` found in ``lib_path``'s .hip_fatbin to ``out_dir``; returns the paths."""
    fat = os.path.join(out_dir, "fatbin.bin")
    subprocess.run([_tool("llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib_path, os.path.join(out_dir, "discard.so")],
                   check=True, capture_output=True)
    data = open(fat, "rb").read()
    paths = []
    pos = data.find(_MAGIC)
    while pos >= 0:
        (n,) = struct.unpack_from("<Q", data, pos + len(_MAGIC))
        off = pos + len(_MAGIC) + 8
        for _ in range(n):
            eoff, esize, tlen = struct.unpack_from("<QQQ", data, off)
            off += 24
            triple = data[off:off + tlen].decode()
            off += tlen
            if esize and triple.endswith(arch):
                p = os.path.join(out_dir, f"co_{len(paths)}.elf")
                with open(p, "wb") as f:
                    f.write(data[pos + eoff:pos + eoff + esize])
                paths.append(p)
        pos = data.find(_MAGIC, pos + 1)
    return paths


def _demangle(names: List[str]) -> Dict[str, str]:
    import shutil
    filt = os.path.join(LLVM_BIN, "llvm-cxxfilt")
    if not os.path.exists(filt):
        filt = shutil.which("c++filt")
    if not filt or not names:
        return {n: n for n in names}
    out = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
    return {n: d for n, d in zip(names, out)}


def kernel_table(lib_path: str) -> Dict[str, Dict[str, int]]:
    """demangled kernel name (without its argument list) -> {field: value} for every gfx950 kernel in the library."""
    table: Dict[str, Dict[str, int]] = {}
    with tempfile.TemporaryDirectory() as td:
        for co in extract_code_objects(lib_path, td):
            txt = subprocess.run([_tool("llvm-readelf"), "--notes", co], capture_output=True, text=True, check=True).stdout
            # the metadata note prints as YAML; one "- .agpr_count: ..." item per kernel inside amdhsa.kernels
            kern = txt.split("amdhsa.kernels:", 1)[1] if "amdhsa.kernels:" in txt else ""
            kern = kern.split("amdhsa.target:", 1)[0]
            items = re.split(r"\n\s+- \.(?=[a-z_]+:)", "\n" + kern)
            recs = []
            for it in items:
                m = re.search(r"\.name:\s+(\S+)", it)
                if not m or ".vgpr_count:" not in it:
                    continue
                rec = {}
                for f in _FIELDS:
                    mm = re.search(rf"\.{f}:\s+(\d+)", it)
                    rec[f] = int(mm.group(1)) if mm else 0
                recs.append((m.group(1).strip("'\""), rec))
            dm = _demangle([n for n, _ in recs])
            for n, rec in recs:
                name = dm[n]
                name = re.sub(r"^void\s+", "", name)
                name = re.sub(r"\((?:[^()]|\([^()]*\))*\)\s*(\[clone[^\]]*\])?$", "", name).strip()     # drop the parameter list
                name = name.replace("(anonymous namespace)::", "")
                table[name] = rec
    return table


def spilling(table: Dict[str, Dict[str, int]]) -> Dict[str, Dict[str, int]]:
    return {k: v for k, v in table.items() if v["vgpr_spill_count"] or v["sgpr_spill_count"] or v["private_segment_fixed_size"]}


if __name__ == "__main__":
    import sys
    from transfusion_amd.build import LIB_PATH
    paths = [x for x in sys.argv[1:] if not x.startswith("-")]
    t = kernel_table(paths[0] if paths else LIB_PATH)
    bad = spilling(t)
    print(f"{len(t)} kernels, {len(bad)} with spills / scratch")
    for k in sorted(t, key=lambda k: (-t[k]["vgpr_spill_count"], k)):
        v = t[k]
        if "-a" in sys.argv or k in bad:
            print(f"{v['vgpr_count']:4d} v {v['agpr_count']:4d} a {v['sgpr_count']:4d} s  spill v{v['vgpr_spill_count']:4d} s{v['sgpr_spill_count']:3d}  "
                  f"scratch {v['private_segment_fixed_size']:5d} B  lds {v['group_segment_fixed_size']:6d}  {k}")
