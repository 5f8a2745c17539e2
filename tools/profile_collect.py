#!/usr/bin/env python3
"""Turn the raw rocprofv3 output of tools/profile_round.sh into the committed summaries:

    python tools/profile_collect.py <tag>

    profiles/<tag>_kernel_stats.csv      per-kernel time (rocprofv3 --kernel-trace --stats)
    profiles/<tag>_pmc_fetch_size.csv    per-kernel FETCH_SIZE (KiB) of the hbvx kernels, averaged
    profiles/<tag>_pmc_write_size.csv    per-kernel WRITE_SIZE (KiB)
    profiles/<tag>_bench.json            the bench line of the profiled run
    profiles/<tag>_lstm_kernel_stats.csv, <tag>_lstm.json   tools/bench_lstm.py (sequence LSTM beside torch's)
    profiles/<tag>_dpl_kernel_stats.csv, <tag>_dpl.json     examples/train_dpl.py --lstm fused
    profiles/<tag>_sq_counters_cfg{2,3,5}.txt   per-kernel SQ counter means (tools/diag_pmc.sh) + vector instructions per SIMD and cycle
    profiles/pmc_traffic.json            HBM bytes per ABI call (read by bench.py for roofline.traffic)
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALLS = {
    "hbvx_forward": ["k_fwd_pipe", "k_fwd_tiled", "k_fwd_stream"],
    "hbvx_backward": ["k_bwd_chunk_phi", "k_bwd_chunk_scan", "k_bwd_chunk_sweep", "k_bwd_chunk_reduce",
                      "k_bwd_tiled", "k_bwd_stream", "k_bwd_pipe"],
    "hbvx_route_forward": ["k_route_fwd", "k_uh_gamma"],
    "hbvx_route_backward": ["k_route_bwd"],
    "hbvx_bfi": ["k_bfi"],
    "hbvx_zero": ["k_zero_nt", "k_zero_gaps", "k_zero_bytes"],
}


def short(name):
    return name.split("(")[0].replace("void ", "").strip()


def counter_table(path):
    """kernel -> (dispatches, mean counter value) from a rocprofv3 counter_collection csv."""
    acc = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            acc[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in acc.items()}


def sq_counters(tag, out):
    """Per-kernel means of the SQ counter passes (tools/diag_pmc.sh <tag>_<cfg>) and the derived figures
    DESIGN.md quotes: vector instructions per SIMD and cycle = SQ_INSTS_VALU / (SIMDs the kernel can use) /
    (GRBM_GUI_ACTIVE / 8 XCDs).  (Until round 4 this printed "VALU busy" = SQ_ACTIVE_INST_VALU x 4 cycles / SIMD-cycles,
    on the assumption that a wave64 instruction holds the pipe four cycles; after the prefetch fix that came out at
    128 % for the transfer-map kernel, and round 3's micro-benchmark had already read 0.70 fma per SIMD and cycle at
    four waves per SIMD: the assumption was wrong, the old percentages are relative figures only.)"""
    for cfg in ("cfg2", "cfg3", "cfg5"):
        acc = defaultdict(lambda: defaultdict(list))
        dur = defaultdict(list)
        grids = {}
        for f in glob.glob(os.path.join(ROOT, "gpurun_out", f"pmc_{tag}_{cfg}", "p*", "**", "*counter_collection.csv"),
                           recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"]).replace("hbvx::", "")
                if not k.startswith("k_"):
                    continue
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
                grids[k] = (int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0))
        if not acc:
            continue
        with open(os.path.join(out, f"{tag}_sq_counters_{cfg}.txt"), "w") as f:
            f.write(f"# rocprofv3 --pmc passes of tools/bench_configs.py {cfg} (tools/diag_pmc.sh), per-launch means\n")
            for k, v in sorted(acc.items()):
                d = sorted(dur[k])[len(dur[k]) // 2]
                if d < 0.2:
                    continue
                m = {c: sum(x) / len(x) for c, x in v.items()}
                f.write(f"{k}  (median {d:.3f} ms under the profiler)\n")
                for c in sorted(m):
                    f.write(f"     {c:32s} {m[c] / 1e6:12.3f} M\n")
                if "GRBM_GUI_ACTIVE" in m and "SQ_INSTS_VALU" in m:
                    gs, ws = grids.get(k, (0, 0))
                    wgs = gs // ws if ws else 0
                    cus = min(256, wgs) if wgs else 256          # one workgroup per CU at most when the grid is small
                    rate = m["SQ_INSTS_VALU"] / (cus * 4) / (m["GRBM_GUI_ACTIVE"] / 8)
                    f.write(f"     -> {rate:.2f} vector instructions per SIMD and cycle on the {cus} CUs the grid covers "
                            f"(a pure fma loop at four waves per SIMD: 0.70, profiles/r03_issue_rate.txt)\n")


def main(tag):
    """(see the module docstring)"""
    raw = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    out = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(raw, "kt", "**", "*kernel_stats.csv"), recursive=True)
    shutil.copy(stats[0], os.path.join(out, f"{tag}_kernel_stats.csv"))
    with open(os.path.join(ROOT, "gpurun_out", f"prof_{tag}.bench.json")) as f:
        line = [l for l in f if l.startswith("{")][-1]
    with open(os.path.join(out, f"{tag}_bench.json"), "w") as f:
        f.write(line)
    st = glob.glob(os.path.join(raw, "cfgs", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        shutil.copy(st[0], os.path.join(out, f"{tag}_configs_kernel_stats.csv"))
        js = os.path.join(ROOT, "gpurun_out", f"prof_{tag}.cfgs.jsonl")
        if os.path.exists(js):
            shutil.copy(js, os.path.join(out, f"{tag}_bench_configs.jsonl"))
    for extra in ("lstm", "dpl"):
        st = glob.glob(os.path.join(raw, extra, "**", "*kernel_stats.csv"), recursive=True)
        js = os.path.join(ROOT, "gpurun_out", f"prof_{tag}.{extra}.json")
        if st and os.path.exists(js):
            shutil.copy(st[0], os.path.join(out, f"{tag}_{extra}_kernel_stats.csv"))
            with open(js) as f:
                lines = [l for l in f if l.startswith("{")]
            with open(os.path.join(out, f"{tag}_{extra}.json"), "w") as f:
                f.write(lines[-1])
    NOTE = ("per ABI call (sum over its kernels), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in "
            "separate passes, KiB x1024.  hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE: the guide's x2 for "
            "FETCH_SIZE also holds for 4 B/lane row loads (profiles/r02_pmc_calibration.csv); "
            "WRITE_SIZE is exact for row and 16-byte streams and counts 16-byte pieces of a line twice")

    def traffic_of(suffix, csv_tag):
        tables = {}
        for kind in ("fetch", "write"):
            found = glob.glob(os.path.join(raw, kind + suffix, "**", "*counter_collection.csv"), recursive=True)
            if not found:
                return None
            tab = {k: v for k, v in counter_table(found[0]).items() if "hbvx" in k or k.startswith("k_")}
            tables[kind] = tab
            with open(os.path.join(out, f"{tag}_pmc_{kind}_size{csv_tag}.csv"), "w") as f:
                f.write("kernel,dispatches,mean_%s_SIZE_KiB\n" % kind.upper())
                for k, (n, v) in sorted(tab.items()):
                    f.write(f"\"{k}\",{n},{v:.3f}\n")
        traffic = {}
        for call, pats in CALLS.items():
            ks = sorted({k for kind in tables.values() for k in kind if any(p in k for p in pats)})
            if not ks:
                continue
            fb = sum(tables["fetch"].get(k, (0, 0.0))[1] for k in ks) * 1024
            wb = sum(tables["write"].get(k, (0, 0.0))[1] for k in ks) * 1024
            traffic[call] = {"kernels": ks, "fetch_bytes_raw": fb, "write_bytes": wb, "hbm_bytes_raw": fb + wb,
                             "hbm_bytes": 2.0 * fb + wb}
        return traffic

    # top level: the headline configuration (bench.py, cfg2); "configs": the others (tools/bench_configs.py)
    traffic = traffic_of("", "") or {}
    traffic["note"] = NOTE
    traffic["configs"] = {}
    for cfg, name in (("cfg3", "cfg3"), ("cfg5", "cfg5share")):
        t = traffic_of("_" + cfg, "_" + cfg)
        if t:
            traffic["configs"][name] = t
    with open(os.path.join(out, "pmc_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    sq_counters(tag, out)
    print(json.dumps({k: (round(v["fetch_bytes_raw"] / 1e9, 3), round(v["write_bytes"] / 1e9, 3))
                      for k, v in traffic.items() if isinstance(v, dict) and "write_bytes" in v}))
    for name, t in traffic["configs"].items():
        print(name, json.dumps({k: (round(v["fetch_bytes_raw"] / 1e9, 3), round(v["write_bytes"] / 1e9, 3)) for k, v in t.items()}))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02")
