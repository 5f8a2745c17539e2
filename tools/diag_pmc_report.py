#!/usr/bin/env python3
"""Per-kernel means of the counter passes collected by tools/diag_pmc.sh (values in millions)."""
import collections, csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}", "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("hbvx::", "")
        if not k.startswith("k_"):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, v in acc.items():
    d = sorted(dur[k])[len(dur[k]) // 2]
    if d < 0.3:
        continue
    print(f"{k}  (median {d:.3f} ms under the profiler)")
    for c in sorted(v):
        print(f"     {c:32s} {sum(v[c]) / len(v[c]) / 1e6:12.3f} M")
