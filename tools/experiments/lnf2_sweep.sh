#!/bin/bash
# LayerNorm forward, form 2 (ln_fwd2_kernel): rows in flight per wave x grid cap against the one-row-per-wave kernel (TF_LNF_V=0).
# Needs build/variants/lnf2 (tools/build_variant.sh lnf2 rowops.hip)
export TFUSION_LIB=$PWD/build/variants/lnf2/libtfusion_hip.so
for rows in 16640 2080; do
  export KB_ROWS=$rows
  echo "== rows $rows: former kernel"; TF_LNF_V=0 python3 tools/kernel_bench.py ln 20 2>/dev/null | grep ln_fwd
  for r in 1 2 4; do for g in 512 768 1024 1536; do
    echo "== rows $rows: v2 ROWS=$r GRID=$g"; TF_LNF_V=1 TF_LNF_ROWS=$r TF_LNF_GRID=$g python3 tools/kernel_bench.py ln 20 2>/dev/null | grep ln_fwd
  done; done
done
