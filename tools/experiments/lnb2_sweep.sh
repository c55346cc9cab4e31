#!/bin/bash
# LayerNorm backward, form 2 (ln_bwd2_kernel): alone-time sweep over its configurations (TF_LNB_CFG: waves per workgroup, rows in
# flight per wave, column partials in LDS or registers, waves per SIMD) against the former kernel (TF_LNB_V=0).
# Needs build/variants/lnb2 (tools/build_variant.sh lnb2 rowops.hip).
# usage (GPU box): bash tools/experiments/lnb2_sweep.sh > gpurun_out/lnb2_sweep.txt 2>&1
export TFUSION_LIB=$PWD/build/variants/lnb2/libtfusion_hip.so
for rows in 16640 2080; do
  export KB_ROWS=$rows
  echo "== rows $rows: former kernel"; TF_LNB_V=0 python3 tools/kernel_bench.py ln 20 | grep ln_bwd
  for c in 0 1 2 3 4 5 6 7 8; do
    echo "== rows $rows: v2 CFG=$c"
    TF_LNB_V=1 TF_LNB_CFG=$c python3 tools/kernel_bench.py ln 20 | grep ln_bwd
  done
done
