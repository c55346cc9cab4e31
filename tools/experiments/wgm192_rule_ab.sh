export TFUSION_LIB=$PWD/build/variants/wgm3/libtfusion_hip.so
for rep in 1 2 3; do for v in 0 -1 2; do ms=$(TF_WGM_192=$v timeout -k 10 200 python3 bench.py --steps 30 --warmup 5 --no-census --no-cpu-baseline --no-legs ${AB_ARGS} 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"); echo "step TF_WGM_192=$v rep $rep -> $ms ms"; done; done
