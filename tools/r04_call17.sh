#!/bin/bash
# round 4, lease 17: who stores the trajectory in the pipelined forward, by ensemble size (HBVX_PIPE_DIRECT pins the form)
mkdir -p gpurun_out
for rnd in 1 2; do for dflag in 0 1; do
  echo "== HBVX_PIPE_DIRECT=$dflag round $rnd"
  HBVX_PIPE_DIRECT=$dflag timeout -k 10 300 python tools/bench_configs.py grid:hbv:672:7300:16 grid:hbv:1344:7300:8 grid:hbv:2688:7300:4 grid:hbv:5376:3650:2 grid:hbv_2:1344:7300:8 grid:hbv_2:2688:7300:4 hourly cfg3 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms']; print(d['config'], 'M', d['M'], d['ms_per_step'], 'fwd', k.get('hbvx_forward'))"
done; done > gpurun_out/r04_direct_by_members.txt 2>&1
cat gpurun_out/r04_direct_by_members.txt
