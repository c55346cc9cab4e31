export TFUSION_LIB=$PWD/build/variants/exp_api/libtfusion_hip.so
for rep in 1 2; do for v in "TF_FORK_LATE=0" "TF_FORK_LATE=1"; do
  env $v timeout -k 10 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-legs 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', d['ms_per_step'], 'wgrad in situ', r['avg_launch_us'], 'frac', r['frac'], 'alone', r['alone']['frac'], r['alone'].get('chip_to_itself',{}).get('frac'))"
done; done
