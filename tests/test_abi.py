"""CPU suite, part 2: the C-ABI library builds, loads and exports every symbol
include/xpoly_amd.h declares; without a GPU it must fail loudly, never fall back."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from xpoly_amd import build, _capi
    build.build()
    return _capi.lib()


def test_library_exports_every_declared_symbol():
    lib = _lib()
    hdr = open(os.path.join(ROOT, "include", "xpoly_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(xpg_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 30
    from xpoly_amd._capi import SYMBOLS
    assert sorted(SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name), name


def test_no_gpu_means_loud_failure():
    lib = _lib()
    if lib.xpg_device_count() > 0:
        pytest.skip("a GPU is present")
    import xpoly_amd
    with pytest.raises(xpoly_amd.XpgError):
        xpoly_amd.Context(0)


def test_product_never_touches_the_oracle():
    """The shipped package must not import, link or read anything under oracle/."""
    pkg = os.path.join(ROOT, "xpoly_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.replace("the CPU oracle", ""), os.path.join(dirpath, f)
    hdr = open(os.path.join(ROOT, "include", "xpoly_amd.h")).read()
    assert "oracle" not in hdr


def test_status_codes_match_reference():
    hdr = open(os.path.join(ROOT, "include", "xpoly_amd.h")).read()
    want = {"XPG_SIX_SUCC": 0, "XPG_SIX_UNBOUND": 1, "XPG_SIX_NO_PRI_FEASIBLE_SOL": 2,
            "XPG_SIX_OPTIMAL_IS_INFEASIBLE": 3, "XPG_SIX_TIME_OUT": 4, "XPG_IP_SUCC": 0,
            "XPG_IP_UNBOUND": 1, "XPG_IP_NO_PRI_FEASIBLE_SOL": 2, "XPG_IP_NO_BETTER_THAN_BEST_SOL": 3}
    for k, v in want.items():
        m = re.search(r"#define\s+%s\s+(-?\d+)" % k, hdr)
        assert m and int(m.group(1)) == v, k


def test_header_is_plain_c99_and_links():
    """The drop-in boundary is a C ABI: a C99 translation unit includes the header, links the
    library and (without a GPU) gets XPG_ERR_NO_DEVICE back instead of a crash."""
    import subprocess
    _lib()
    exe = os.path.join(ROOT, "tests", "cxx", "abi_c99")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx", "abi_c99.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "xpoly_amd"), "-lxpoly_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "xpoly_amd")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr


def test_fma_canon_equals_the_two_reference_operations():
    """The rational sweep's fused a + k*e (scalar.hip.h, fma_canon / mul_canon / add_canon) against add(a, mul(k, e)),
    the reference's two operations (src/com/rational.cpp:273-310, :363-397), on 2 M canonical triples that include
    the float32 `appro` rescue: host code only, no GPU."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cxx", "fma_canon_fuzz")
    subprocess.check_call(["hipcc", "-O2", "-ffp-contract=off", "-w", "-o", exe, os.path.join(ROOT, "tests", "cxx", "fma_canon_fuzz.cpp")])
    r = subprocess.run([exe, "2000000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fma_canon == add(mul)" in r.stdout, r.stdout + r.stderr


def test_rational_inputs_outside_int32_are_refused():
    """xpoly's Rational is int32 / int32 (rational.h:66-67): the Python mirror must not wrap wider integers silently."""
    import numpy as np
    import pytest
    from xpoly_amd.six import RAT, as_kind
    ok = as_kind(np.array([[1, -2], [2147483647, -2147483648]], dtype=np.int64), RAT, 2)
    assert ok.dtype == np.int32 and ok.shape == (2, 2, 2) and ok[1, 0, 0] == 2147483647
    with pytest.raises(ValueError):
        as_kind(np.array([[1, 2147483648]], dtype=np.int64), RAT, 2)
    with pytest.raises(ValueError):
        as_kind(np.array([[[1, 1], [-2147483649, 1]]], dtype=np.int64), RAT, 2)
    with pytest.raises(ValueError):
        as_kind(np.array([[0.5, 1.0]]), RAT, 2)
