#!/usr/bin/env python3
"""Per-kernel register / spill / LDS figures of a built HIP library, read from the gfx950 code
object's metadata notes (no GPU needed).

    python tools/kernel_resources.py [lib.so] [substring ...]

Used by tests/test_code_object.py (no kernel may spill) and when tuning occupancy: waves per SIMD
allowed by registers = min(8, 512 // (ceil((vgpr + agpr) / 8) * 8))  (MI355X_MICROARCH.md, register
files)."""
from __future__ import annotations

import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "hydrodl2_amd", "csrc", "libhbvx.so")


def code_objects(lib: str, out_dir: str) -> list[str]:
    """Extract the gfx950 code objects of `lib` (one offload bundle per translation unit) into out_dir."""
    fat = os.path.join(out_dir, "fatbin")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    out = []
    for k, st in enumerate(starts):
        end = starts[k + 1] if k + 1 < len(starts) else len(blob)
        part = os.path.join(out_dir, f"bundle{k}")
        with open(part, "wb") as f:
            f.write(blob[st:end])
        co = os.path.join(out_dir, f"gfx950_{k}.co")
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}",
                               f"--output={co}"])
        out.append(co)
    return out


def kernel_table(lib: str = DEFAULT_LIB) -> list[dict]:
    with tempfile.TemporaryDirectory() as td:
        notes = "".join(subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True,
                                       capture_output=True, text=True).stdout
                        for co in code_objects(lib, td))
    rows = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        def field(key, _b=blk):
            m = re.search(r"\.%s:\s+(\S+)" % key, _b)
            return m.group(1) if m else "0"
        rows.append({"symbol": field("name"), "agpr": int(blk.split()[0]), "vgpr": int(field("vgpr_count")),
                     "sgpr": int(field("sgpr_count")), "vgpr_spill": int(field("vgpr_spill_count")),
                     "sgpr_spill": int(field("sgpr_spill_count")), "lds": int(field("group_segment_fixed_size")),
                     "scratch": int(field("private_segment_fixed_size")),
                     "max_threads": int(field("max_flat_workgroup_size"))})
    names = subprocess.run(["c++filt"], input="\n".join(r["symbol"] for r in rows), capture_output=True,
                           text=True, check=True).stdout.split("\n")
    for r, n in zip(rows, names):
        r["name"] = n.replace("hbvx::", "")
        alloc = -(-(r["vgpr"] + r["agpr"]) // 8) * 8
        r["waves_per_simd"] = min(8, 512 // max(alloc, 8))
    return rows


def main():
    args = sys.argv[1:]
    lib = DEFAULT_LIB
    if args and args[0].endswith(".so"):
        lib = args.pop(0)
    rows = kernel_table(lib)
    for r in rows:
        if args and not any(a in r["name"] for a in args):
            continue
        print(f"{r['name'][:110]:110s} v={r['vgpr']:3d} a={r['agpr']:3d} w/simd={r['waves_per_simd']} "
              f"spill={r['vgpr_spill']:3d} sgpr={r['sgpr']:3d} scratch={r['scratch']}")
    print(f"{len(rows)} kernels, {sum(1 for r in rows if r['vgpr_spill'] or r['sgpr_spill'])} with spills")


if __name__ == "__main__":
    main()


def disassemble(lib: str, symbol_substrings: list[str]) -> dict[str, list[str]]:
    """Instruction mnemonics + operands of the kernels whose mangled name contains one of the substrings:
    {symbol: ["v_add_f32 ...", ...]} from llvm-objdump of the library's gfx950 code objects."""
    out: dict[str, list[str]] = {}
    with tempfile.TemporaryDirectory() as td:
        for co in code_objects(lib, td):
            syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", co], capture_output=True, text=True,
                                  check=True).stdout
            want = sorted({ln.split()[-1] for ln in syms.splitlines()
                           if " FUNC " in ln and any(s in ln.split()[-1] for s in symbol_substrings)})
            for sym in want:
                txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f"--disassemble-symbols={sym}", co],
                                     capture_output=True, text=True, check=True).stdout
                ins = []
                for ln in txt.splitlines():
                    m = re.match(r"^\s+([a-z_0-9]+(?:\s+[^/]*)?)\s*//", ln)
                    if m:
                        ins.append(" ".join(m.group(1).split()))
                out[sym] = ins
    return out


def disassemble_addr(lib: str, symbol_substrings: list[str]) -> dict[str, list[tuple[int, str]]]:
    """Like disassemble(), with each instruction's address: {symbol: [(addr, "s_waitcnt vmcnt(0)"), ...]}."""
    out: dict[str, list[tuple[int, str]]] = {}
    with tempfile.TemporaryDirectory() as td:
        for co in code_objects(lib, td):
            syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", co], capture_output=True, text=True,
                                  check=True).stdout
            want = sorted({ln.split()[-1] for ln in syms.splitlines()
                           if " FUNC " in ln and any(s in ln.split()[-1] for s in symbol_substrings)})
            for sym in want:
                txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f"--disassemble-symbols={sym}", co],
                                     capture_output=True, text=True, check=True).stdout
                ins = []
                for ln in txt.splitlines():
                    m = re.match(r"^\s+([a-z_0-9]+(?:\s+[^/]*)?)\s*//\s*([0-9A-Fa-f]+):", ln)
                    if m:
                        ins.append((int(m.group(2), 16), " ".join(m.group(1).split())))
                out[sym] = ins
    return out


def loops_of(ins: list[tuple[int, str]]) -> list[tuple[int, int]]:
    """(head index, branch index) of every backward branch of a disassemble_addr() listing.  SOPP branches carry a
    signed 16-bit dword offset relative to the next instruction (printed unsigned by llvm-objdump)."""
    index = {a: i for i, (a, _) in enumerate(ins)}
    out = []
    for i, (a, x) in enumerate(ins):
        if x.startswith(("s_cbranch", "s_branch")):
            try:
                off = int(x.split()[-1])
            except ValueError:
                continue
            off = off - 65536 if off >= 32768 else off
            tgt = a + 4 + 4 * off
            if tgt in index and index[tgt] <= i:
                out.append((index[tgt], i))
    return out
