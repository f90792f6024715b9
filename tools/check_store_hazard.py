"""ISA scan of the built library for the one hazard the compiler cannot see into inline asm for: on gfx950 the data
registers of a VMEM store wider than 8 bytes are read AFTER issue, so a VALU write to one of them within the next two
wait states can land in the store (lp_chain.hip.h `ch_store_u32x4`, lp_fused_r32.hip.h `fr_store`: both found the
hard way). The compiler inserts `s_nop 1` behind its own wide stores; an `asm volatile("global_store_dwordx4 ...")`
has to carry its own.

    python tools/check_store_hazard.py [library]      -> prints offenders, exit code 1 if any

Every `global_store_dwordx3/x4` / `buffer_store_dwordx3/x4` / `flat_store_dwordx3/x4` in the gfx950 code object is
followed through its next two wait states (an `s_nop N` is N + 1 of them, any other instruction one); a VALU instruction
whose destination overlaps the store's data registers inside that window is reported (loads write their destination
when the data returns, hundreds of cycles later: not this hazard, and the compiler does not guard them either).
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(HERE, "..", "xpoly_amd", "libxpoly_amd.so")

_reg = re.compile(r"^v(\d+)$|^v\[(\d+):(\d+)\]$")


def _regs(tok):
    m = _reg.match(tok.strip())
    if not m:
        return set()
    if m.group(1) is not None:
        return {int(m.group(1))}
    return set(range(int(m.group(2)), int(m.group(3)) + 1))


def disassemble(lib):
    with tempfile.TemporaryDirectory() as d:
        out = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--list", "--input=" + lib],
                             capture_output=True, text=True).stdout
        targets = [t for t in out.split() if "gfx950" in t]
        co = os.path.join(d, "dev.co")
        if targets:
            subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", "--input=" + lib,
                                   "--targets=" + targets[0], "--output=" + co])
        else:
            # a linked shared object keeps its code objects in .hip_fatbin: llvm-objdump extracts them next to a copy
            tmp = os.path.join(d, os.path.basename(lib))
            with open(lib, "rb") as f, open(tmp, "wb") as g:
                g.write(f.read())
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", tmp], capture_output=True, text=True, cwd=d)
            cands = sorted(os.path.join(d, f) for f in os.listdir(d) if "gfx950" in f)
            if not cands:
                raise RuntimeError("no gfx950 code object in " + lib)
            # (one code object per translation unit: the library is linked from four parts)
            return "\n".join(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", c], capture_output=True, text=True).stdout
                             for c in cands)
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True).stdout


def scan(asm):
    """-> (wide stores seen, [(function, line number, store, offender)])"""
    fn = "?"
    lines = asm.split("\n")
    insns = []                                   # (function, line number, mnemonic, operand tokens, text)
    for k, l in enumerate(lines):
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", l)
        if m:
            fn = m.group(1)
            continue
        t = l.split("//")[0].strip()
        if not t or t.endswith(":"):
            continue
        parts = t.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        insns.append((fn, k + 1, parts[0], ops, t))
    bad, seen = [], 0
    for i, (f, ln, mn, ops, text) in enumerate(insns):
        if not re.match(r"^(global|buffer|flat|scratch)_store_dwordx[34]$", mn):
            continue
        seen += 1
        # global_store: addr, data, saddr | buffer_store: data, ... | flat_store: addr, data
        data = _regs(ops[0]) if mn.startswith("buffer") else (_regs(ops[1]) if len(ops) > 1 else set())
        states, j = 0, i + 1
        while states < 2 and j < len(insns) and insns[j][0] == f:
            _, ln2, mn2, ops2, text2 = insns[j]
            if mn2 == "s_nop":
                states += int(ops2[0], 0) + 1
                j += 1
                continue
            if mn2.startswith("s_branch") or mn2.startswith("s_cbranch") or mn2 in ("s_endpgm", "s_setpc_b64"):
                break                                                # (a taken branch costs more than the window)
            writes = mn2.startswith("v_") and not mn2.startswith("v_cmp") and not mn2.startswith("v_cmpx")
            if writes and ops2 and (_regs(ops2[0]) & data):
                bad.append((f, ln, text, text2))
                break
            states += 1
            j += 1
    return seen, bad


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else DEFAULT_LIB
    seen, bad = scan(disassemble(lib))
    print("%d wide VMEM stores scanned, %d with a VALU write to their data registers inside two wait states" % (seen, len(bad)))
    for f, ln, st, off in bad:
        print("  %s (line %d): %s  ->  %s" % (f, ln, st, off))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
