#!/bin/bash
# per-kernel attention timings (rocprofv3 kernel trace of tools/kernel_bench.py attn); usage: tools/attn_prof.sh <tag>
tag=${1:-attn}
out=/root/repo/gpurun_out/prof_$tag
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o $tag -- python /root/repo/tools/kernel_bench.py attn 20 > $out.log 2>&1
f=$(find $out -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then grep "attn_" "$f" | cut -d, -f1-6; else echo "no stats file"; tail -5 $out.log; fi
