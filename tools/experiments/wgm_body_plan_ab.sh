#!/bin/bash
# where a body layer's weight-gradient products start (TF_WGM_BODY, tf_api.hip wgrad_plan): all after the attention backward (0x000,
# shipped), the FFN pair behind the FFN-down dgrad (0x003), FFN pair + out-proj behind the LN1 backward (0x070), the tail layer's plan (0x043)
export TFUSION_LIB=$PWD/build/variants/exp_api/libtfusion_hip.so
for rep in 1 2; do for v in 0x000 0x003 0x070 0x043; do
  ms=$(TF_WGM_BODY=$v timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-census --no-cpu-baseline --no-legs 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); echo "TF_WGM_BODY=$v rep $rep -> $ms ms"
done; done
