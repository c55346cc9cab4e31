#!/bin/bash
# SQ counter pass over tools/kernel_bench.py wgrad for both wgrad kernels; usage: tools/wgrad_pmc.sh
out=/root/repo/gpurun_out/pmc_wgrad
cd /tmp && export TMPDIR=/tmp
for v in 1 2; do
TF_WGRAD_V=$v timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/v$v -o w -- python /root/repo/tools/kernel_bench.py wgrad 3 > $out.v$v.log 2>&1
f=$(find $out/v$v -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "wgrad" not in k: continue
    k = k.split("::")[-1].split("(")[0] + " grid=" + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in acc.items():
    w = c["SQ_WAVE_CYCLES"]
    print(k, "launches", n[k], " mfma_util(2 waves/SIMD if full) = busy/(wave_cycles*4) =", round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (w * 4), 3))
    for name, v in sorted(c.items()):
        print(f"   {name:28s} {v / n[k]:14.0f}  {v / w:7.3f} of WAVE_CYCLES")
PY
done
