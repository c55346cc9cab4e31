#!/bin/bash
# the chain <-> side-stream events with a device-scope release / without the system-scope fence against the default flags:
# build/variants/ev_base, ev_dev, ev_nofence (tools/build_variant.sh <name> tf_api.hip "-DTF_EVENT_FLAGS=(...)")
for rep in 1 2 3; do for v in ev_base ev_dev ev_nofence; do
  ms=$(TFUSION_LIB=$PWD/build/variants/$v/libtfusion_hip.so timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-census --no-cpu-baseline --no-legs ${AB_ARGS} 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "step $v rep $rep -> $ms ms"
done; done
