"""tools/isa_handoff.py -- the build-time check that the fence-free level-to-level hand-off of k_track_quad<.., LEVELS>
(variant 7; csrc/pagk_quad_kernel.h) still has, in the compiler's assembly, the shape its correctness argument rests
on.  The checker itself is tested here on a hand-written assembly skeleton: the good shape passes, each way of losing
it is reported.  (The real assembly is checked by __graft_entry__.build_hip() on every build.)"""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import isa_handoff  # noqa: E402

GOOD = """
_ZN4pagk12k_track_quadILi7ELb1ELb1ELb0EEEvNS_9TrackArgsE: ; @kernel
	global_atomic_add v3, v3, v4, s[34:35] sc0
	s_waitcnt vmcnt(0)
.LBB0_1:
	global_load_dword v1, v3, s[8:9] sc1
	s_waitcnt vmcnt(0)
	s_sleep 0x7f
	global_load_dword v1, v3, s[8:9] sc1
	;;#ASMSTART
	; pagk-handoff: take-over begin
	;;#ASMEND
	global_load_dword v40, v[8:9], off sc1
	global_load_dword v41, v[8:9], off offset:4 sc1
	global_load_dword v93, v[8:9], off offset:8 sc1
	global_load_dword v10, v[8:9], off offset:12 sc1
	s_waitcnt vmcnt(0)
	;;#ASMSTART
	; pagk-handoff: take-over end
	;;#ASMEND
	v_add_f32 v1, v2, v3
	;;#ASMSTART
	; pagk-handoff: state begin
	;;#ASMEND
	global_store_dword v[2:3], v0, off sc1
	global_store_dword v[2:3], v1, off offset:4 sc1
	global_store_dword v[2:3], v93, off offset:8 sc1
	global_store_dword v[2:3], v4, off offset:12 sc1
	;;#ASMSTART
	; pagk-handoff: state end
	;;#ASMEND
	;;#ASMSTART
	; pagk-handoff: publish begin
	;;#ASMEND
	s_waitcnt vmcnt(0)
	v_mov_b32 v2, 1
	global_atomic_add v1, v1, v2, s[10:11] offset:256 sc0
	s_waitcnt vmcnt(0)
	v_add_u32 v0, v1, v5
	global_store_dword v[0:1], v2, off sc1
	;;#ASMSTART
	; pagk-handoff: publish end
	;;#ASMEND
	s_endpgm
.Lfunc_end0:
"""


def test_the_good_shape_passes():
    assert isa_handoff.check(GOOD) == []


@pytest.mark.parametrize("what, old, new, expect", [
    ("a state store lost its agent scope", "global_store_dword v[2:3], v1, off offset:4 sc1", "global_store_dword v[2:3], v1, off offset:4",
     "state store without agent scope"),
    ("the wait in front of the slot atomic is gone", "\ts_waitcnt vmcnt(0)\n\tv_mov_b32 v2, 1\n", "\tv_mov_b32 v2, 1\n", "publish sequence"),
    ("the wait between the atomic and the entry store is gone", "offset:256 sc0\n\ts_waitcnt vmcnt(0)\n", "offset:256 sc0\n", "publish sequence"),
    ("the publishing store lost its agent scope", "global_store_dword v[0:1], v2, off sc1", "global_store_dword v[0:1], v2, off", "publishing store is not agent scope"),
    ("a state load lost its agent scope", "global_load_dword v93, v[8:9], off offset:8 sc1", "global_load_dword v93, v[8:9], off offset:8", "state load without agent scope"),
    ("the polling load lost its agent scope", "s_sleep 0x7f\n\tglobal_load_dword v1, v3, s[8:9] sc1", "s_sleep 0x7f\n\tglobal_load_dword v1, v3, s[8:9]", "polling load without agent scope"),
    ("a fence crept into the take-over", "global_load_dword v40, v[8:9], off sc1", "buffer_inv sc1\n\tglobal_load_dword v40, v[8:9], off sc1", "cache fence"),
    ("the slot atomic no longer returns", "global_atomic_add v1, v1, v2, s[10:11] offset:256 sc0", "global_atomic_add v1, v2, s[10:11] offset:256", "does not return"),
    ("the markers are gone", "; pagk-handoff: state begin", "; something else", "no `state` region"),
])
def test_each_way_of_losing_the_shape_is_reported(what, old, new, expect):
    assert old in GOOD, what
    problems = isa_handoff.check(GOOD.replace(old, new, 1))
    assert any(expect in p for p in problems), (what, problems)


def test_an_assembly_without_the_kernel_is_reported():
    assert isa_handoff.check("_ZN4pagk13k_track_blockILi2E: \n s_endpgm\n") != []
