#!/bin/bash
# round 4, lease 4: graph mode tests + host overhead eager vs graph; cfg4 with the new soil-moisture solve (A/B + probe)
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_graphed.py -m gpu -x -q > gpurun_out/r04_graphed.log 2>&1; echo "graphed rc=$?"; tail -15 gpurun_out/r04_graphed.log
for i in 1 2 3; do
  timeout -k 10 120 python tools/host_overhead.py 400 dmg 2>&1 | grep enqueue
  timeout -k 10 120 python tools/host_overhead.py 400 dmggraph 2>&1 | grep -E "enqueue|Error" 
done > gpurun_out/r04_host_overhead.txt 2>&1
cat gpurun_out/r04_host_overhead.txt
timeout -k 10 600 python -m pytest tests/test_hbv_adj.py -m gpu -x -q > gpurun_out/r04_adj_tests.log 2>&1; echo "adj tests rc=$?"; tail -5 gpurun_out/r04_adj_tests.log
AB_CFGS="cfg4 dmg dmggraph" timeout -k 10 400 python tools/ab_libs.py libhbvx_oldsoil.so libhbvx_base.so > gpurun_out/r04_ab_soil.txt 2>&1
cat gpurun_out/r04_ab_soil.txt
