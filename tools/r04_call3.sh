#!/bin/bash
# round 4, lease 3: GPU tier with the XCD-aware chunk map and graph mode; A/B of the map; host overhead eager vs graph
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_graphed.py -m gpu -x -q > gpurun_out/r04_graphed.log 2>&1; echo "graphed rc=$?"; tail -15 gpurun_out/r04_graphed.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputier_lease3.log 2>&1; echo "gputier rc=$?" | tee -a gpurun_out/r04_gputier_lease3.log
tail -4 gpurun_out/r04_gputier_lease3.log
AB_CFGS="cfg2 cfg2dyn cfg3 cfg4" timeout -k 10 400 python tools/ab_libs.py libhbvx_plainmap.so libhbvx_base.so > gpurun_out/r04_ab_xcdmap.txt 2>&1
cat gpurun_out/r04_ab_xcdmap.txt
for i in 1 2 3; do
  timeout -k 10 120 python tools/host_overhead.py 400 dmg 2>/dev/null | grep enqueue
  timeout -k 10 120 python tools/host_overhead.py 400 dmggraph 2>/dev/null | grep enqueue
done > gpurun_out/r04_host_overhead.txt 2>&1
cat gpurun_out/r04_host_overhead.txt
