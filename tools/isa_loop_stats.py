#!/usr/bin/env python3
"""Instruction mix of the loops of each kernel in a gfx950 assembly listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S x.hip -o x.s
    python tools/isa_loop_stats.py x.s [kernel-substring]

For every kernel: each innermost-to-outermost loop body (label .. backward branch) with its
counts of VALU / transcendental / SALU / VMEM / LDS / lane-spill instructions.  The time loops of
the steppers are issue-bound (a wave issues one instruction every ~6-8 cycles whatever it is), so instructions and
s_waitcnt per day are the figures of merit (a vmcnt wait right behind a burst of loads means the
prefetch does not fly, profiles/r04_ab_chunk_prefetch.txt)."""
from __future__ import annotations

import re
import subprocess
import sys

TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")


def classify(op: str) -> str:
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernels(path: str):
    cur, body = None, []
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if cur:
                yield cur, body
            cur, body = m.group(1), []
            continue
        if cur is not None:
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                yield cur, body
                cur, body = None, []
                continue
            body.append(line.rstrip("\n"))
    if cur:
        yield cur, body


def loops(body):
    labels = {}
    insts = []
    for ln in body:
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        s = ln.strip()
        if not s or s.startswith((";", ".")):
            continue
        insts.append(s.split(";")[0].split())
    out = []
    for i, ins in enumerate(insts):
        if ins and ins[0].startswith("s_cbranch") or (ins and ins[0] == "s_branch"):
            tgt = ins[-1]
            if tgt in labels and labels[tgt] <= i:
                out.append((labels[tgt], i))
    return insts, out


def main():
    path = sys.argv[1]
    sel = sys.argv[2:] or [""]
    for sym, body in kernels(path):
        name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip().replace("hbvx::", "")
        if not any(s in name for s in sel):
            continue
        insts, lps = loops(body)
        print(f"{name[:120]}: {len(insts)} instructions")
        for a, b in sorted(lps, key=lambda ab: ab[0] - ab[1])[:3]:
            c = {}
            for ins in insts[a:b + 1]:
                k = classify(ins[0])
                c[k] = c.get(k, 0) + 1
            print(f"   loop [{a}:{b}] {b - a + 1:5d} instr  " +
                  " ".join(f"{k}={c[k]}" for k in ("valu", "trans", "salu", "vmem", "lds", "lane", "wait", "other") if k in c))


if __name__ == "__main__":
    main()
