#!/bin/bash
# wgrad_multi192_kernel (192 x 192 tiles, two quads, in-workgroup reduction) against the 256 x 128 form: alone and in the step.
# Needs build/variants/wgm3 (tools/build_variant.sh wgm3 wgrad_multi.hip)
export TFUSION_LIB=$PWD/build/variants/wgm3/libtfusion_hip.so
for rep in 1 2 3; do for v in 0 -1; do echo "== TF_WGM_192=$v rep $rep"; TF_WGM_192=$v KB_BLOCKS=-1,0 python3 tools/kernel_bench.py wgradm 20 2>/dev/null | grep "merged, blocks"; done; done
for rep in 1 2 3; do for v in 0 -1; do ms=$(TF_WGM_192=$v timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-census --no-cpu-baseline --no-legs ${AB_ARGS} 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); echo "step TF_WGM_192=$v rep $rep -> $ms ms"; done; done
