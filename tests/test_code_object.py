"""CPU tier: properties of the BUILT library's gfx950 code objects (no GPU needed).

No kernel may spill vector registers to scratch -- a spill on a time-stepper wave is memory traffic on
the critical chain -- with one documented exception: the `W4` instances of k_bwd_stream2, which are
compiled for four waves per SIMD on purpose and measured faster than the spill-free three-wave form
on grids that overflow three waves per SIMD by a little (hbv_stream2.h).  Scalar-register spills go
to VGPR lanes (no memory) and are reported, not failed."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "hydrodl2_amd", "csrc", "libhbvx.so")


@pytest.fixture(scope="module")
def table():
    if not os.path.exists(LIB):
        pytest.skip("libhbvx.so not built")
    import kernel_resources
    return kernel_resources.kernel_table(LIB)


def _is_w4(name: str) -> bool:
    return name.startswith("void k_bwd_stream2<") and name.split("(")[0].rstrip(">").endswith(", true")


def test_no_vector_register_spills(table):
    assert len(table) > 250, "expected every kernel family in the library"
    bad = [(r["name"], r["vgpr_spill"]) for r in table if r["vgpr_spill"] and not _is_w4(r["name"])
           and not r["name"].startswith("void k_bwd_stream2_ckpt<")]
    # (k_bwd_stream2_ckpt, the memory-lean adjoint with its K-day segment in LDS: compiled for three waves per SIMD
    # like the streaming adjoint it stands in for; its one-item-ahead loads across the phase boundaries cost 10-40
    # spilled values in the capillary / hourly instances.  Measured 1.15 x the saved-trajectory pair at config 5 --
    # it is the path for records that do not fit otherwise, not the fast path; profiles/r05_ckpt_ab.jsonl)
    assert not bad, f"kernels spilling VGPRs: {bad}"
    w4 = [r for r in table if _is_w4(r["name"])]
    assert w4 and all(r["waves_per_simd"] >= 4 for r in w4)
    assert max(r["vgpr_spill"] for r in w4) <= 32


def test_time_steppers_keep_their_occupancy(table):
    """Occupancy the dispatch logic relies on: the streaming forward fits four waves per SIMD (a 12 500-
    basin share is 3 125 waves), the streaming adjoint three."""
    by = {r["name"].split("(")[0]: r for r in table}
    for name, r in by.items():
        if name.startswith("void k_fwd_stream2<"):
            six = name.split("<")[1].split(", ")[3] == "4"      # SC == 4: the six-slot run-time list (three waves for 2.0 / hourly)
            assert r["waves_per_simd"] >= (3 if six else 4), (name, r["vgpr"])
        if name.startswith("void k_bwd_stream2<") and not _is_w4(r["name"]):
            assert r["waves_per_simd"] >= 2, (name, r["vgpr"])
    assert by["void k_bwd_stream2<2, true, 2, 2, false, true, false>"]["waves_per_simd"] >= 3


def _load_bursts(ins, min_loads=8, max_gap=12):
    """[(first, last)] instruction index ranges of bursts of at least `min_loads` register loads (LDS-DMA excluded)."""
    loads = [i for i, x in enumerate(ins)
             if x.startswith(("global_load_dword", "buffer_load_dword")) and " lds" not in x]
    bursts, cur = [], []
    for i in loads:
        if cur and i - cur[-1] > max_gap:
            if len(cur) >= min_loads:
                bursts.append((cur[0], cur[-1]))
            cur = []
        cur.append(i)
    if len(cur) >= min_loads:
        bursts.append((cur[0], cur[-1]))
    return bursts


def test_the_adjoints_prefetch_is_not_waited_for_where_it_is_issued():
    """The time-parallel adjoint kernels issue a day's loads one day ahead.  Any arithmetic on a value among them --
    round 4 found `0.0f + grad_flux4[...]` in the issue step -- makes the compiler wait for ALL of them on the spot
    (vector-memory loads return in order): a memory round trip per day (profiles/r04_ab_chunk_prefetch.txt).  So: no
    s_waitcnt vmcnt inside, or within a few instructions behind, a burst of loads in these kernels."""
    if not os.path.exists(LIB):
        pytest.skip("libhbvx.so not built")
    import kernel_resources
    # (trailing ...ELb0EE: the instances without learned ensemble weights, MU = false)
    kernels = ["k_bwd_chunk_phiILi0ELb0ELi0ELb0ELi0ELb0EE", "k_bwd_chunk_sweepILi0ELb0ELi0ELb0ELi0ELb0ELb0EE",
               "k_bwd_chunk_phiILi0ELb1ELi1ELb0ELi1ELb0EE", "k_bwd_chunk_sweepILi0ELb1ELi1ELb0ELi1ELb0ELb0EE",
               "k_bwd_chunk_phiILi4ELb1ELi1ELb0ELi2ELb0EE", "k_adj_chunk_phiILb1ELb1E"]
    dis = kernel_resources.disassemble(LIB, kernels)
    assert len(dis) == len(kernels), sorted(dis)
    for sym, ins in dis.items():
        bursts = _load_bursts(ins)
        assert bursts, f"{sym}: no load burst found (pattern changed?)"
        for a, b in bursts:
            waits = [ins[i] for i in range(a, min(b + 6, len(ins))) if ins[i].startswith("s_waitcnt") and "vmcnt" in ins[i]]
            assert not waits, f"{sym}: {waits} between / right behind the loads at instructions {a}..{b}"


def _is_dma(x: str) -> bool:
    return x.startswith("global_load_lds") or (x.startswith("buffer_load") and x.endswith(" lds"))


def test_lds_dma_rows_are_waited_for_before_they_are_read():
    """Kernels that bring a day's rows HBM -> LDS by LDS-DMA (no registers while in flight) read them back with
    ds_read at the top of the day's loop.  Nothing but a `s_waitcnt vmcnt` orders the two: round 4 had a build whose
    LDS reads ran ahead of the DMA and read STALE rows at config 3's full size.  The chunk kernels now say it in the
    source (an explicit vmcnt(0): everything in flight there is the day's own input), the streaming adjoint relies on
    the compiler's own tracking of `buffer_load ... lds` (counted vmcnt(n), one per DMA'd row).  Either way the ISA
    must show, in the day loop (the widest loop that holds the DMA), a vmcnt wait between the loop head and its first ds_read."""
    if not os.path.exists(LIB):
        pytest.skip("libhbvx.so not built")
    import kernel_resources
    dis = kernel_resources.disassemble_addr(LIB, ["k_bwd_chunk_phiILi1ELb1ELi3E", "k_bwd_chunk_sweepILi1ELb1ELi3E",
                                                  "k_bwd_stream2I"])
    assert len(dis) >= 6 + 40, sorted(dis)
    checked = 0
    for sym, ins in dis.items():
        loops = [(h, b) for h, b in kernel_resources.loops_of(ins) if any(_is_dma(x) for _, x in ins[h:b + 1])]
        if not loops:
            assert "k_bwd_stream2" in sym, f"{sym}: no loop with an LDS-DMA"    # (some streaming forms load to registers)
            continue
        h, b = max(loops, key=lambda hb: hb[1] - hb[0])     # the day loop (inner backward branches: its conditionals)
        body = [x for _, x in ins[h:b + 1]]
        first_read = next((i for i, x in enumerate(body) if x.startswith("ds_read")), None)
        assert first_read is not None, f"{sym}: DMA loop without a ds_read"
        first_dma = next(i for i, x in enumerate(body) if _is_dma(x))
        if "k_bwd_stream2" in sym and first_read > first_dma:
            # the run-time slot-list instances (SC == 3) are full of scalar branch trees and the block placement no
            # longer starts the loop at the read-back: the positional check does not apply to them (same source lines,
            # same compiler mechanism as the compiled-set instances checked here)
            assert "ELi3ELb" in sym or "ELi4ELb" in sym, f"{sym}: the loop does not start with the read-back of the previous DMA"
            continue
        assert first_read < first_dma, f"{sym}: the loop does not start with the read-back of the previous DMA"
        waits = [x for x in body[:first_read] if x.startswith("s_waitcnt") and "vmcnt" in x]
        assert waits, f"{sym}: no vmcnt wait between the loop head and {body[first_read]!r}"
        if "chunk" in sym:
            assert any("vmcnt(0)" in w for w in waits), f"{sym}: the explicit vmcnt(0) is gone: {waits}"
        checked += 1
    assert checked >= 6 + 20
