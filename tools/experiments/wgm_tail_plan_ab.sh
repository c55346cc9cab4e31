#!/bin/bash
# where the LAST layer's weight-gradient products start (TF_WGM_TAIL, tf_api.hip wgrad_plan): 0x843 shipped (FFN pair behind the FFN-down
# dgrad, out-proj behind the LN1 backward, in-proj behind the attention backward), 0x070 (FFN pair + out-proj together behind the LN1
# backward: one fork fewer), 0x003 (FFN pair early, out-proj + in-proj together at the end)
export TFUSION_LIB=$PWD/build/variants/exp_api/libtfusion_hip.so
for rep in 1 2 3; do for v in 0x843 0x070 0x003; do
  ms=$(TF_WGM_TAIL=$v timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-census --no-cpu-baseline --no-legs 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); echo "TF_WGM_TAIL=$v rep $rep -> $ms ms"
done; done
