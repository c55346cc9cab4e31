#!/bin/bash
# The round's closing measurements in one go, ON THE GPU BOX (one gpurun call; ~6 GPU-minutes):
#   1. PMC traffic of every kernel (tools/traffic_pmc.sh)            -> profiles/traffic.json (stamped with the kernel sources' hash)
#   2. SQ counters of the MFMA kernels (tools/sq_pmc_bench.sh)       -> profiles/<tag>_sq_counters.txt
#   3. python bench.py (defaults; reads the fresh traffic.json)      -> profiles/<tag>_bench.json
#   4. rocprofv3 --kernel-trace --stats of the headline              -> profiles/<tag>_bench_kernel_stats.csv
#   5. ... and of the real module's step (tools/wrapper_time.py)     -> profiles/<tag>_wrapper_real_kernel_stats.csv
# usage: /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/final_profiles.sh r06_v1'
# The files land under gpurun_out/final_<tag>/ (merged back by gpurun); copy them into profiles/ and add the README rows.
set -o pipefail
tag=${1:?usage: tools/final_profiles.sh <tag>}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/final_$tag
mkdir -p $out $root/gpurun_out/pmc_traffic
cd $root
bash tools/traffic_pmc.sh > $out/traffic.txt 2>&1 || { echo "traffic pass failed"; tail -5 $out/traffic.txt; exit 1; }
cp gpurun_out/traffic.json $out/traffic.json && cp gpurun_out/traffic.json profiles/traffic.json
bash tools/sq_pmc_bench.sh $tag > $out/sq.log 2>&1 || { echo "SQ passes failed"; tail -5 $out/sq.log; exit 1; }
cp gpurun_out/sq_$tag.txt $out/${tag}_sq_counters.txt
timeout -k 10 600 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { echo "bench failed"; tail -5 $out/${tag}_bench.err; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd $root
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_b -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-census --no-legs > $out/prof_b.log 2>&1 \
  && cp "$(find $out/prof_b -name '*kernel_stats.csv' | head -1)" $out/${tag}_bench_kernel_stats.csv
export B=4 REAL=1 TRAINER=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_w -o w -- python3 tools/wrapper_time.py > $out/prof_w.log 2>&1 \
  && cp "$(find $out/prof_w -name '*kernel_stats.csv' | head -1)" $out/${tag}_wrapper_real_kernel_stats.csv
rm -rf $out/prof_b $out/prof_w gpurun_out/pmc_traffic gpurun_out/sq_$tag
python3 - $out/${tag}_bench.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print("headline", r["ms_per_step"], "ms/step", r["value"], r["unit"], "| roofline frac", r["roofline"]["frac"], "traffic_ratio", r["roofline"].get("traffic_ratio"),
      "stale" if r["roofline"].get("traffic_stale") else "fresh")
print({k: v["ms_per_step"] for k, v in r.get("legs", {}).items()})
PY
ls $out
