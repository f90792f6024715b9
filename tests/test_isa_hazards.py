"""The built library's ISA holds no unguarded wide store (tools/check_store_hazard.py): a VALU write to the data
registers of a > 8-byte VMEM store inside its next two wait states. The compiler guards its own stores; inline-asm
stores (the self-validating 16-byte granules of lp_chain.hip.h and lp_fused_r32.hip.h) must carry their own s_nop --
round 4's fused Rational loop shipped without one."""
import os
import shutil

import pytest

from tools import check_store_hazard as chk

BAD = """
0000000000001000 <k_demo>:
	global_store_dwordx4 v[8:9], v[4:7], off sc1               // 000000028844: DE7C8000 007F0408
	s_cmp_eq_u64 vcc, 0                                        // 000000028850: BF12806A
	v_or_b32_e32 v5, v3, v47                                   // 000000028854: 280A5F03
	s_endpgm
"""
GOOD = BAD.replace("\ts_cmp_eq_u64 vcc, 0", "\ts_nop 3\n\ts_cmp_eq_u64 vcc, 0")
FAR = BAD.replace("\ts_cmp_eq_u64 vcc, 0", "\ts_cmp_eq_u64 vcc, 0\n\ts_mov_b32 s0, 0")


def test_scanner_sees_the_round4_pattern_and_accepts_the_guarded_forms():
    seen, bad = chk.scan(BAD)
    assert seen == 1 and len(bad) == 1 and "v_or_b32_e32 v5" in bad[0][3]
    assert chk.scan(GOOD) == (1, [])
    assert chk.scan(FAR) == (1, [])                     # two other instructions are two wait states


@pytest.mark.skipif(not os.path.exists(os.path.join(chk.LLVM, "llvm-objdump")) or shutil.which("hipcc") is None
                    and not os.path.exists(chk.DEFAULT_LIB), reason="needs the ROCm LLVM tools")
def test_built_library_has_no_unguarded_wide_store():
    if not os.path.exists(chk.DEFAULT_LIB):
        pytest.skip("library not built")
    seen, bad = chk.scan(chk.disassemble(chk.DEFAULT_LIB))
    assert seen > 100, seen                             # (all four parts were found)
    assert not bad, bad
