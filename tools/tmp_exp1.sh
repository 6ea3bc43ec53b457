cd /root/repo
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -s -k "bench_ranks or one_rank_rccl" --durations=5 > gpurun_out/r06/bench_rank_tests.log 2>&1
tail -15 gpurun_out/r06/bench_rank_tests.log
