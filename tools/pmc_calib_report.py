#!/usr/bin/env python3
"""profiles/r02_pmc_calibration.csv from the two counter passes of tools/micro/pmc_calib (run by
tools/diag_r02.sh): per access shape, bytes the kernel moved by construction vs bytes FETCH_SIZE /
WRITE_SIZE report (KiB x 1024), and the factor to apply to the counter."""
import csv, glob, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "gpurun_out", "diag")
truth = dict(l.split() for l in open(os.path.join(D, "calib_bytes.txt")) if l.startswith("calib"))
rows = []
for kind, counter in (("f", "FETCH_SIZE"), ("w", "WRITE_SIZE")):
    f = glob.glob(os.path.join(D, f"calib_{kind}", "**", "*counter_collection.csv"), recursive=True)[0]
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k in truth and r["Counter_Name"] == counter:
            is_read = "read" in k
            if is_read != (counter == "FETCH_SIZE"):
                continue
            actual = int(truth[k])
            counted = float(r["Counter_Value"]) * 1024.0
            ms = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
            rows.append((k, counter, actual, int(counted), round(actual / counted, 3), round(ms, 4),
                         round(actual / ms / 1e6, 1)))
out = os.path.join(ROOT, "profiles", "r02_pmc_calibration.csv")
with open(out, "w") as f:
    f.write("kernel,counter,bytes_by_construction,bytes_counted,multiply_counter_by,kernel_ms_under_profiler,GBps\n")
    for r in rows:
        f.write(",".join(str(v) for v in r) + "\n")
print(open(out).read())
