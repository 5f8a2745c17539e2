#!/usr/bin/env python3
"""Timeline of ONE bench step from a rocprofv3 kernel trace: every kernel with its start offset,
duration and the idle gap before it.  Shows what a step spends outside the hbvx kernels (torch's
elementwise kernels around the loss, launch gaps).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python3 bench.py --steps 3 --warmup 2 --no-secondary --no-cpu-baseline
    python tools/step_timeline.py gpurun_out/tl
"""
import csv
import glob
import os
import sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last step: from the last k_fwd_pipe / k_fwd_stream2 launch to the end of the trace
starts = [i for i, r in enumerate(rows) if "k_fwd_pipe" in r[2] or "k_fwd_stream" in r[2]]
i0 = starts[-1]
# include the fill that runs on the side stream just before / beside the forward
while i0 > 0 and ("k_zero" in rows[i0 - 1][2] or rows[i0 - 1][0] > rows[i0][0] - 20000):
    i0 -= 1
t0 = rows[i0][0]
prev_end = t0
busy = 0
for s, e, name in rows[i0:]:
    gap = s - prev_end
    short = name.split("(")[0].replace("void ", "").replace("hbvx::", "")[:70]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  gap {gap / 1e3:7.1f}  {short}")
    prev_end = max(prev_end, e)
print(f"step span {(prev_end - t0) / 1e3:.1f} us")
