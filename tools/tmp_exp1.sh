cd /root/repo
mkdir -p gpurun_out/r06
timeout 1500 python -m pytest tests/test_cpp_dropin.py -m gpu -q -x -s > gpurun_out/r06/cpp_tests.log 2>&1
grep -n 'visible device\|failed\|passed\|FAIL\|dealt' gpurun_out/r06/cpp_tests.log | tail -20
