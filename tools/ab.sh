#!/bin/bash
# same-box A/B of environment switches: tools/ab.sh "VAR=1" "VAR=2 OTHER=3" ...   (first run = defaults)
# prints ms/step of `bench.py --steps 30 --warmup 5 --no-census --no-cpu-baseline` for each setting, twice (run-to-run noise)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for v in "" "$@"; do
    ms=$(env $v timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-census --no-cpu-baseline --no-legs ${AB_ARGS} 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "rep $rep  ${v:-<defaults>}  ->  $ms ms/step"
  done
done
