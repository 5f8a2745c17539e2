#!/bin/bash
# lease 35: which of the two adjoint changes breaks config 3 at full size
mkdir -p gpurun_out
cd hydrodl2_amd/csrc; cp libhbvx.so libhbvx_keep.so; cd ../..
for v in keep gld; do
  cp hydrodl2_amd/csrc/libhbvx_$v.so hydrodl2_amd/csrc/libhbvx.so
  echo "== $v"
  timeout -k 10 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k cfg3 2>&1 | tail -3 | cut -c1-200
done
cp hydrodl2_amd/csrc/libhbvx_keep.so hydrodl2_amd/csrc/libhbvx.so
