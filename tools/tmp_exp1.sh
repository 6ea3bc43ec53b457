cd /root/repo
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -s -k "c3" --durations=10 > gpurun_out/r06/c3_tests.log 2>&1
tail -30 gpurun_out/r06/c3_tests.log
