#!/bin/bash
# SQ counter pass over tools/kernel_bench.py attn; prints per-kernel sums.  usage: tools/attn_pmc.sh <tag>
tag=${1:-pmc}
out=/root/repo/gpurun_out/pmc_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv -d $out -o $tag -- python /root/repo/tools/kernel_bench.py attn 3 > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
if [ -z "$f" ]; then echo "no counter file"; tail -5 $out.log; exit 1; fi
python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "attn_" not in k or "dropmask" in k: continue
    k = k.split("::")[-1].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in acc.items():
    w = c["SQ_WAVE_CYCLES"]
    print(k, "launches", n[k])
    for name, v in sorted(c.items()):
        print(f"   {name:28s} {v / n[k]:14.0f}  {v / w:7.3f} of WAVE_CYCLES")
PY
