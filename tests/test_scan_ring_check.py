"""tools/check_scan_ring.py proves the scan kernel's hand-counted load ring on the emitted machine code (run by build()).
Here: the built library passes, and the analysis is not vacuous — planted violations of each kind are found."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import check_scan_ring as C          # noqa: E402


@pytest.fixture(scope="module")
def code():
    from coral_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    objs = [(d, n) for d, n in C.device_code_objects(_lib.LIB_PATH) if "k_cigar_scan_v4" in d]
    assert len(objs) == 1
    return objs[0]


@pytest.mark.parametrize("ring", [6, 4, 8, 12])
def test_built_kernel_keeps_its_ring_intact(code, ring):
    assert C.verify(code[0], code[1], ring, verbose=False) == []


def _kernel(code, ring=6):
    return C.parse_kernel(code[0], "k_cigar_scan_v4ILi%dEE" % ring)


def _loop_ring_loads(ins):
    """Indices of the ring loads that are directly followed by a counted wait (the unrolled round)."""
    return [k for k, (a, mn, ops) in enumerate(ins) if mn == "global_load_dwordx4" and ops.endswith(" nt")
            and ins[k + 1][1] == "s_waitcnt" and "vmcnt(5)" in ins[k + 1][2]]


def test_a_copy_of_an_in_flight_ring_register_is_found(code):
    ins = _kernel(code)
    k = _loop_ring_loads(ins)[2]
    dest = min(C.vregs(ins[k][2].split(",")[0]))
    planted = ins[:k + 2] + [(ins[k + 1][0] + 1, "v_mov_b32_e32", "v200, v%d" % dest)] + ins[k + 2:]       # the copy a register allocator might make
    errs = C.check(planted, 6)
    assert errs and "touches v[%d]" % dest in errs[0]


def test_a_clobbered_in_flight_ring_register_is_found(code):
    ins = _kernel(code)
    k = _loop_ring_loads(ins)[4]
    dest = max(C.vregs(ins[k][2].split(",")[0]))
    planted = ins[:k + 2] + [(ins[k + 1][0] + 1, "v_add_u32_e32", "v%d, v201, v202" % dest)] + ins[k + 2:]
    assert any("touches v[%d]" % dest in e for e in C.check(planted, 6))


def test_a_wait_that_retires_too_little_is_found(code):
    ins = _kernel(code)
    k = _loop_ring_loads(ins)[1] + 1
    planted = list(ins)
    planted[k] = (ins[k][0], "s_waitcnt", "vmcnt(6)")          # one load too many left in flight: the chunk read next is not there yet
    assert C.check(planted, 6)
    text = re.sub(r"s_waitcnt vmcnt\(5\)", "s_waitcnt vmcnt(6)", code[0], count=0)
    assert any("counted waits" in e for e in C.verify(text, code[1], 6, verbose=False))


def test_a_missing_drain_is_found(code):
    """Without the vmcnt(0) waits behind the loop the exact path would run (and the kernel could end) with ring loads in flight."""
    ins = _kernel(code)
    planted = [(a, "s_nop", "0") if (mn == "s_waitcnt" and "vmcnt(0)" in ops) else (a, mn, ops) for a, mn, ops in ins]
    assert C.check(planted, 6)
