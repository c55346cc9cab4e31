#!/bin/bash
# wgrad_multi_kernel: v_mfma_f32_16x16x32_bf16 (TF_WGM_MF16=1) against v_mfma_f32_32x32x16_bf16 (0), alone and in the step.
# Needs build/variants/wgm2 (tools/build_variant.sh wgm2 wgrad_multi.hip)
export TFUSION_LIB=$PWD/build/variants/wgm2/libtfusion_hip.so
for rep in 1 2 3; do for v in 0 1; do echo "== MF16=$v rep $rep"; TF_WGM_MF16=$v KB_BLOCKS=432,216 python3 tools/kernel_bench.py wgradm 20 2>/dev/null | grep "merged, blocks"; done; done
for rep in 1 2 3; do for v in 0 1; do ms=$(TF_WGM_MF16=$v timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-census --no-cpu-baseline --no-legs ${AB_ARGS} 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); echo "step MF16=$v rep $rep -> $ms ms"; done; done
