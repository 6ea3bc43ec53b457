cd /root/repo
mkdir -p gpurun_out/r06
for r in 1 2; do
for w in 0 1 2; do echo "== GPE_TAIL_XCD=$w"; GPE_TAIL_XCD=$w timeout 300 python tools/tail_try.py 2048 3072 4096; done
done > gpurun_out/r06/xcd_rows.log 2>&1
cat gpurun_out/r06/xcd_rows.log
