#!/bin/bash
# second SQ counter pass over tools/kernel_bench.py attn (LDS / issue-class split); usage: tools/attn_pmc2.sh <tag>
tag=${1:-pmc2}
out=/root/repo/gpurun_out/pmc2_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $out -o $tag -- python /root/repo/tools/kernel_bench.py attn 3 > $out.log 2>&1
f=$(find $out -name "*counter_collection.csv" | head -1)
if [ -z "$f" ]; then echo "no counter file"; tail -15 $out.log; exit 1; fi
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
