"""__graft_entry__.check_priorities -- the build-time check that the issue priorities of the 4-wave kernels (csrc/pagk_prio.h: a
workgroup that is behind its neighbours runs at s_setprio 3, the others by phase below it) are in the compiler's assembly.  The
checker is tested here on a hand-written skeleton; the real assembly is checked by build_hip() on every build."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

CLAIMS = {"k_track_blockILi2E": (3, 2, 1, 0), "k_track_block5": (3, 2, 0)}
GOOD = """
_ZN4pagk13k_track_blockILi2ELi25EEEvNS_9TrackArgsE: ; @kernel
	s_setprio 3
	v_add_f32 v1, v2, v3
	s_setprio 2
	s_setprio 1
	s_setprio 0
	s_endpgm
.Lfunc_end0:
_ZN4pagk14k_track_block5ILi2ELi25ELb1EEEvNS_9TrackArgsE: ; @kernel
	s_setprio 3
	s_setprio 2
	s_setprio 0
	s_endpgm
.Lfunc_end1:
_ZN4pagk9k_unrelatedEv:
	s_setprio 1
	s_endpgm
.Lfunc_end2:
"""


def test_the_good_shape_passes():
    assert g.check_priorities(GOOD, CLAIMS) == []


def test_a_kernel_that_lost_its_top_priority_is_reported():
    bad = GOOD.replace("\ts_setprio 3\n\tv_add_f32", "\tv_add_f32")
    msgs = g.check_priorities(bad, CLAIMS)
    assert len(msgs) == 1 and "k_track_blockILi2E" in msgs[0] and "[3]" in msgs[0]


def test_priorities_of_another_kernel_do_not_count():
    # block5 needs no priority 1; the unrelated kernel's s_setprio 1 must not satisfy anything either way
    bad = GOOD.replace("\ts_setprio 2\n\ts_setprio 0\n\ts_endpgm\n.Lfunc_end1", "\ts_setprio 0\n\ts_endpgm\n.Lfunc_end1")
    msgs = g.check_priorities(bad, CLAIMS)
    assert len(msgs) == 1 and "k_track_block5" in msgs[0] and "[2]" in msgs[0]


def test_a_missing_kernel_is_reported():
    msgs = g.check_priorities(GOOD, {"k_track_wave": (3,)})
    assert msgs and "missing" in msgs[0]


def test_the_products_claims_name_the_headline_kernel():
    assert any(k.startswith("k_track_blockILi2ELi25ELi4ELb0ELb0ELb1") for k in g.PRIORITY_CLAIMS)
