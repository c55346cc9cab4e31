#!/bin/bash
# SQ counters of the training step's MFMA kernels (GEMM, wgrad, attention), IN the benchmark's own launch sequence:
# two rocprofv3 --pmc passes (8 SQ slots each) over `python bench.py` (program directly after `--`), per-kernel sums.
# Counter mode serialises the kernels, so the side-stream wgrads are measured without the chain beside them.
# usage (on the GPU box): tools/sq_pmc_bench.sh <tag> [extra bench args]   -> gpurun_out/sq_<tag>.txt
tag=${1:-r02}; shift
out=/root/repo/gpurun_out/sq_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS"
P2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $out/p$i -o t -- python /root/repo/bench.py --steps 2 --warmup 1 --no-census --no-cpu-baseline --no-legs "$@" > $out/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/p$i.log; exit 1; }
  echo "pass $i done"
done
python3 - $out <<'PY' > /root/repo/gpurun_out/sq_$tag.txt
import csv, sys, glob, collections, re
out = sys.argv[1]
print("# rocprofv3 --pmc (two passes) over `python bench.py --steps 2 --warmup 1 --no-census --no-cpu-baseline`; mean per launch of each kernel symbol.")
print("# SQ_* cycle counters are summed over all waves (quad-cycles); SQ_VALU_MFMA_BUSY_CYCLES in cycles; MFMA pipe utilisation =")
print("# MFMA_BUSY / (4 * BUSY_CYCLES_per_SE-summed ...) is not portable across SE counts, so the table gives the ratio the guide uses:")
print("# MFMA_BUSY_CYCLES / (WAVE_CYCLES * 4 / waves_per_SIMD) is left to the reader; columns are raw means and fractions of WAVE_CYCLES.")
for i in (1, 2):
    f = glob.glob(f"{out}/p{i}/**/*counter_collection.csv", recursive=True)
    if not f:
        print(f"pass {i}: no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"]
        if not any(t in k for t in ("gemm_nt", "wgrad_", "attn_fwd", "attn_bwd", "ln_bwd")): continue
        k = re.sub(r"^void ", "", k).replace("(anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
    print(f"\n==== pass {i} ====")
    for k in sorted(acc, key=lambda k: -acc[k]["SQ_WAVE_CYCLES"]):
        c = acc[k]; w = c["SQ_WAVE_CYCLES"]
        print(f"{k}  launches {n[k]}")
        for name, v in sorted(c.items()):
            print(f"   {name:28s} {v / max(n[k], 1):16.0f}  {v / w:7.3f} of WAVE_CYCLES")
PY
echo "wrote gpurun_out/sq_$tag.txt"; head -50 /root/repo/gpurun_out/sq_$tag.txt
