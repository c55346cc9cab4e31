python -m pytest tests/test_gpu_kernels.py tests/test_gpu_packed.py -x -q -k "attn or attention or packed" > gpurun_out/t2.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/t2.log
python tools/kernel_bench.py attn 10 2>/dev/null | grep -v amdgpu
