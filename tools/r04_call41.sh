#!/bin/bash
# lease 41: days per thread / loads in flight of the routing kernels
set -e
mkdir -p gpurun_out
AB_ALL=1 AB_CFGS="cfg2 dmg" timeout -k 10 600 python tools/ab_libs.py libhbvx_base.so libhbvx_rt16.so libhbvx_rt64.so libhbvx_rt128.so > gpurun_out/r04_c41_ab.txt 2>&1
python - <<'PY'
import re
for l in open('gpurun_out/r04_c41_ab.txt'):
    print(l[:400])
PY
