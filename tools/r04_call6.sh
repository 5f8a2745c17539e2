#!/bin/bash
# round 4, lease 6: the soil-moisture solve with free Halley updates (cfg4 forward) against round 3's
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_hbv_adj.py -m gpu -x -q > gpurun_out/r04_adj_tests.log 2>&1; echo "adj tests rc=$?"; tail -3 gpurun_out/r04_adj_tests.log
AB_CFGS="cfg4" timeout -k 10 400 python tools/ab_libs.py libhbvx_oldsoil.so libhbvx_base.so libhbvx_vfree1.so > gpurun_out/r04_ab_soil_free.txt 2>&1
cat gpurun_out/r04_ab_soil_free.txt
