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
    bad = [(r["name"], r["vgpr_spill"]) for r in table if r["vgpr_spill"] and not _is_w4(r["name"])]
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
            assert r["waves_per_simd"] >= 4, (name, r["vgpr"])
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
    kernels = ["k_bwd_chunk_phiILi0ELb0ELi0ELb0ELi0E", "k_bwd_chunk_sweepILi0ELb0ELi0ELb0ELi0ELb0E",
               "k_bwd_chunk_phiILi0ELb1ELi1ELb0ELi1E", "k_bwd_chunk_sweepILi0ELb1ELi1ELb0ELi1ELb0E",
               "k_bwd_chunk_phiILi4ELb1ELi1ELb0ELi2E", "k_adj_chunk_phiILb1ELb1E"]
    dis = kernel_resources.disassemble(LIB, kernels)
    assert len(dis) == len(kernels), sorted(dis)
    for sym, ins in dis.items():
        bursts = _load_bursts(ins)
        assert bursts, f"{sym}: no load burst found (pattern changed?)"
        for a, b in bursts:
            waits = [ins[i] for i in range(a, min(b + 6, len(ins))) if ins[i].startswith("s_waitcnt") and "vmcnt" in ins[i]]
            assert not waits, f"{sym}: {waits} between / right behind the loads at instructions {a}..{b}"
