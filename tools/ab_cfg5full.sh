#!/bin/bash
# Dev tool: cfg5 at full size (100 000 basins x 16 x 730 on one GPU) with two builds of the library, alternating.
#   tools/ab_cfg5full.sh libhbvx_A.so libhbvx_B.so   (looked up in hydrodl2_amd/csrc/; the shipped library is restored)
set -e
C=$(dirname $(readlink -f $0))/../hydrodl2_amd/csrc
cp $C/libhbvx.so $C/libhbvx_base.so
for rnd in 1 2; do
  for l in "$@"; do
    cp $C/$l $C/libhbvx.so
    echo "== $l round $rnd"
    python3 $(dirname $0)/../bench.py --config cfg5 --gpus 1 --steps 5 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'], 3), d['roofline']['whole_step']['kernel_ms'])"
  done
done
cp $C/libhbvx_base.so $C/libhbvx.so
