"""The reference's own call sites, UNEDITED, must compile against the adapters.

INTEGRATION.md promises a type substitution: `SIX` / `MIP` / `Lineq` at the reference's call sites become
`xpoly_amd::SIX` / `xpoly_amd::MIP` / `xpoly_amd::Lineq<RMat>` and nothing else changes. This test holds the
promise to the letter. It cuts the text of the call sites out of the reference where it lies (read at run
time -- no reference text is stored in this repository), puts it into functions whose only additions are the
parameter lists the surrounding member functions would have provided, with

    using xpoly_amd::SIX;  using xpoly_amd::MIP;  typedef xpoly_amd::Lineq<RMat> Lineq;

in scope, and runs `g++ -fsyntax-only` on the result against the REAL `src/com` headers plus
`include/xpoly_amd/{six,lineq}.hpp`.

Sites (reference file:line):
  * `Lineq::has_solution`          src/com/linsys.cpp:836-906  (reviseTargetFunc, MIP::maxm(..., false, NULL, rhs_idx))
  * `DepPoly::is_empty`            src/eng/poly.cpp:530-573    (Lineq(NULL), move2var, reduce, has_solution)
  * `PolyTran::FeaSchedule` solver src/eng/poly.cpp:5110-5140  (MIP::reviseTargetFunc, six-argument maxm / minm)
  * `PolyTran` bound computation   src/eng/poly.cpp:4801-4812  (appendEquation, fme)
  * `LoopTran` new loop limits     src/eng/ldtran.cpp:178-193  (Lineq(A, rhs), fme)
  * `formatBound` call             src/eng/ldtran.cpp:1529-1534
  * `initVarConstraint` call       src/eng/poly.cpp:1601-1604
Runs only where /root/reference exists (the authoring container), like the `@ref` oracle tests.
"""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"

pytestmark = pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("g++") is None,
                                reason="needs the reference sources and g++")


def _lines(path):
    with open(os.path.join(REF, path), encoding="latin-1") as f:
        return f.read().split("\n")


def body_of(path, header_regex):
    """The text from the `{` that opens the function whose header matches, through its matching `}`."""
    text = "\n".join(_lines(path))
    m = re.search(header_regex, text)
    assert m, (path, header_regex)
    start = text.index("{", m.end())
    depth, i = 0, start
    while True:
        c = text[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return text[start:i + 1]
        i += 1


def between(path, first_marker, last_marker, after=None, include_last=True):
    """Whole lines from the first line containing `first_marker` (searched after the line containing `after`)
    through the next line containing `last_marker`."""
    ls = _lines(path)
    k = 0
    if after is not None:
        k = next(i for i, l in enumerate(ls) if after in l)
    a = next(i for i in range(k, len(ls)) if first_marker in ls[i])
    b = next(i for i in range(a, len(ls)) if last_marker in ls[i])
    return "\n".join(ls[a:b + 1 if include_last else b])


PRELUDE = r"""
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include "ltype.h"
#include "comf.h"
#include "smempool.h"
#include "strbuf.h"
#include "rational.h"
#include "flty.h"
#include "sstl.h"
#include "matt.h"
#include "xmat.h"
#include "bs.h"
#include "sbs.h"
#include "sgraph.h"
#include "lpsol.h"
#include "linsys.h"
#include "xpoly_amd/six.hpp"
#include "xpoly_amd/lineq.hpp"
namespace xpoly_amd {
template <> struct scalar_kind<xcom::Float> { static const int value = 0; };
template <> struct scalar_kind<xcom::Rational> { static const int value = 1; };
}
namespace site {
using namespace xcom;
// THE substitution of INTEGRATION.md section 2 -- nothing else below is ours except parameter lists
using xpoly_amd::SIX;
using xpoly_amd::MIP;
typedef xpoly_amd::Lineq<RMat> Lineq;
#ifndef UNREACH
#define UNREACH() ASSERT0(0)
#endif
"""


def build_source():
    src = [PRELUDE]
    # 1. Lineq::has_solution: the member's own parameter list, its body as it stands
    src.append("bool has_solution(RMat const& leq, RMat const& eq, RMat & vc, UINT rhs_idx, bool is_int_sol, bool is_unique_sol)\n"
               + body_of("com/linsys.cpp", r"bool\s+Lineq::has_solution\s*\("))
    # 2. DepPoly::is_empty: a shim for the class it is a member of (src/eng/poly.h:43-58, :456), the body as it stands
    src.append("#define DEP_POLY_rhs_idx(d) ((d).rhs_idx)\n"
               "class VarConstraintMat : public INTMat {};\n"
               "class DepPoly : public RMat { public: UINT id; UINT flag; UINT rhs_idx; bool is_empty(bool keepit, VarConstraintMat const* vc); };\n"
               "bool DepPoly::is_empty(bool keepit, VarConstraintMat const* vc)\n"
               + body_of("eng/poly.cpp", r"bool\s+DepPoly::is_empty\s*\("))
    # 3. PolyTran::FeaSchedule's solver block
    blk = between("eng/poly.cpp", "//Prepare data for SIX solver.", "goto FAIL;", after="bool PolyTran::FeaSchedule(")
    src.append("bool fea_schedule_solver_block(RMat & sys, UINT u_count, UINT lam_count)\n{\n" + blk +
               "\n    }\n    (void)h;\nFAIL:\n    return st;\n}\n")
    # 4. PolyTran's bound computation: appendEquation + fme chain
    blk = between("eng/poly.cpp", "RMat tub(lub);", "tub = res;", after="//Append equations to inequality system.")
    src.append("void poly_bounds_block(RMat & lub, RMat & eq, INT new_rhs_idx, RMat & ub)\n{\n    INT i;\n" + blk +
               "\n        }\n    Lineq lin2(NULL);\n    if (!lin2.reduce(ub, new_rhs_idx, false)) { UNREACH(); }\n}\n")
    # 5. LoopTran's new loop limits
    blk = between("eng/ldtran.cpp", "Lineq lineq(A, m_rhs_idx);", "*newb = *A;", after="//Computes new loop limits")
    blk2 = between("eng/ldtran.cpp", "Lineq lineq(A, m_rhs_idx);", "//Record outermost loop bound.",
                   after="//Computes new loop limits", include_last=False)
    src.append("void looptran_limits_block(RMat * A, INT m_rhs_idx, List<RMat*> & aux_limits)\n{\n    INT i;\n" + blk2 + "\n}\n")
    # 6. formatBound / set_param call
    blk = between("eng/ldtran.cpp", "lineq.set_param(ineq, m_rhs_idx);", "lineq.formatBound(ivar, formed);")
    src.append("void format_bound_block(Lineq & lineq, RMat * ineq, INT m_rhs_idx, UINT ivar)\n{\n" + blk + "\n}\n")
    # 7. initVarConstraint call
    blk = between("eng/poly.cpp", "Lineq lin(NULL);", "vc.copy(tmp);", after="buildMapIVCoeff(from, to, coeff);")
    src.append("struct SM { UINT get_num_of_var() const; };\n"
               "void init_vc_block(Vector<INT> & coeff, SM * sm_from, RMat & vc)\n{\n" + blk + "\n}\n")
    # 8. Lineq::is_consistent through the substituted class
    src.append("bool consistent(RMat & m) { Lineq lin(&m); return lin.is_consistent(); }\n")
    src.append("} // namespace site\nint main() { return 0; }\n")
    return "\n".join(src)


def test_reference_call_sites_compile_unedited_against_the_adapters(tmp_path):
    src = build_source()
    # the bodies must really be the reference's: the spellings the adapter used to reject are in there
    assert "mip.reviseTargetFunc(tgtf, eq, leq, num_of_var);" in src
    assert "six.reviseTargetFunc(tgtf, eq, leq, num_of_var);" in src
    assert re.search(r"mip\.maxm\(v, res, tgtf, vc, eq, leq,\s*false, NULL, rhs_idx\)", src)
    assert "six.reviseTargetFunc(tgtf, sys, leq, num_of_var);" in src
    assert "lin.appendEquation(eq);" in src and "lineq.formatBound(ivar, formed);" in src
    assert "lin.initVarConstraint(&coeff, tmp," in src
    f = tmp_path / "callsites.cpp"
    f.write_text(src, encoding="latin-1")
    cmd = ["g++", "-fsyntax-only", "-D_LINUX_", "-Wno-write-strings", "-w", "-I", os.path.join(REF, "com"),
           "-I", os.path.join(REPO, "include"), str(f)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-6000:]
