#!/bin/bash
# wgrad_multi192_kernel: quad 1 half a step behind quad 0 (TF_WGM_STAG=1) against lockstep (0); alone (each in its own process: first
# position) and in the step with the form beside the chain too (TF_WGM_192=2) and under the shipped rule (-1)
export TFUSION_LIB=$PWD/build/variants/wgm3/libtfusion_hip.so
for rep in 1 2 3; do for v in 0 1; do echo "== 192 form STAG=$v rep $rep"; TF_WGM_STAG=$v TF_WGM_192=1 KB_BLOCKS=-1 python3 tools/kernel_bench.py wgradm 20 2>/dev/null | grep "merged, blocks"; done; echo "== 256x128 form rep $rep"; TF_WGM_192=0 KB_BLOCKS=-1 python3 tools/kernel_bench.py wgradm 20 2>/dev/null | grep "merged, blocks"; done
for rep in 1 2 3; do for v in "TF_WGM_192=0" "TF_WGM_192=2 TF_WGM_STAG=0" "TF_WGM_192=2 TF_WGM_STAG=1" "TF_WGM_192=-1 TF_WGM_STAG=1"; do ms=$(env $v timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-census --no-cpu-baseline --no-legs 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); echo "step $v rep $rep -> $ms ms"; done; done
