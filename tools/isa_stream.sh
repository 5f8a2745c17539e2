#!/bin/bash
# Instruction mix of the streaming kernels' day loops for a set of extra compiler flags (no GPU needed):
#   bash tools/isa_stream.sh <tag> [hipcc flags...]
TAG=$1; shift
cd "$(dirname "$0")/../hydrodl2_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize --cuda-device-only -I../../include "$@" \
    -S launch_stream.hip -o /tmp/launch_stream_$TAG.s 2>/dev/null
cd ../..
python tools/isa_loop_stats.py /tmp/launch_stream_$TAG.s "k_bwd_stream2<2, true, 2, 2, false, true, true>" \
    "k_fwd_stream2<2, true, 2, 2, true>" "k_fwd_stream2<0, true, 2, 1, true>" "k_bwd_stream2<0, true, 2, 1, false, true, false>" \
    "k_fwd_stream2<0, true, 2, 0, true>" "k_bwd_stream2<0, true, 2, 0, false, true, false>"
