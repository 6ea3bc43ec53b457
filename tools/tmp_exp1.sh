cd /root/repo
mkdir -p gpurun_out/r06
for r in 1 2; do for d in 0 2 4; do echo "== GPE_BATCH_DLEAD=$d"; GPE_BATCH_DLEAD=$d timeout 300 python tools/c4bench.py 8 4 2; done; done > gpurun_out/r06/batch_dlead.log 2>&1
cat gpurun_out/r06/batch_dlead.log
