#!/usr/bin/env python3
"""Scan gfx950 assembly (hipcc --save-temps *.s, or llvm-objdump -d output) for the sequence behind DESIGN.md section 7's
load-dependent results: a packed-fp32 VALU instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose destination pair is read by
the IMMEDIATELY following VALU instruction with no wait state in between.  The hazard recogniser of this toolchain pads such a pair
with `s_nop 0` only when the producer's src0 has op_sel_hi = 1 (its VOP3 dst-op_sel test shares that modifier bit), so producers written
`op_sel_hi:[0,...]` - the low-half broadcast forms the SLP vectoriser emits for sum / sum-of-squares chains - go unpadded.
usage: scan_pk_hazard.py file.s|file.o [...]   -> per kernel: unpadded dependent pairs / padded pairs; exit status 1 if any pair is unpadded.
A host object (.o) is unbundled in a temporary directory (llvm-objdump --offloading) and its gfx950 code object disassembled."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def disassemble(obj):
    """device ISA text of a hipcc host object"""
    with tempfile.TemporaryDirectory() as td:
        o = os.path.join(td, os.path.basename(obj))
        shutil.copy(obj, o)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", o], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(td) if "amdgcn" in f]
        if not co:
            return ""
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", os.path.join(td, co[0])], check=True,
                              capture_output=True, text=True).stdout

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = set()
    for m in REG.finditer(tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def scan(path):
    kern, res = None, {}
    prev = None          # (mnemonic, dest regs, text) of the previous instruction if it was a packed-fp32 op
    nop_since = False
    text = disassemble(path) if path.endswith(".o") else open(path, errors="replace").read()
    for line in text.splitlines():
        s = line.split(";")[0].split("//")[0].strip()
        if not s:
            continue
        m = re.match(r"^(?:[0-9a-f]+ <)?([A-Za-z_][\w$.]*)>?:", s)
        if m and not s.startswith(".L"):
            kern, prev = m.group(1), None
            continue
        if s.startswith(".") or s.endswith(":"):
            if s.startswith(".LBB") or s.endswith(":"):
                prev = None      # a branch target: the fall-through pair is still adjacent, but keep the scan conservative and simple
            continue
        mn = s.split()[0]
        if mn.startswith("s_nop"):
            nop_since = True
            continue
        ops = s[len(mn):].split(",")
        if prev is not None and mn.startswith("v_"):
            srcs = set()
            for o in ops[1:]:
                srcs |= regs(o)
            if mn.startswith(("v_fmac", "v_pk_fma", "v_mac")) or len(ops) >= 1:
                pass
            # accumulating forms read their destination too
            if mn.startswith(("v_fmac_", "v_mac_", "v_dot2c")):
                srcs |= regs(ops[0])
            if srcs & prev[1]:
                d = res.setdefault(kern, [0, 0, []])
                if nop_since:
                    d[1] += 1
                else:
                    d[0] += 1
                    if len(d[2]) < 3:
                        d[2].append(prev[2] + "  ->  " + s)
        if mn in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32"):
            prev, nop_since = (mn, regs(ops[0]), s), False
        elif not mn.startswith("s_nop"):
            prev = None
    return res


if __name__ == "__main__":
    total = 0
    for p in sys.argv[1:]:
        r = scan(p)
        bad = {k: v for k, v in r.items() if v[0]}
        print(f"{p}: {sum(v[0] for v in r.values())} unpadded dependent packed-fp32 pairs in {len(bad)} kernels, {sum(v[1] for v in r.values())} padded")
        for k, v in sorted(bad.items(), key=lambda kv: -kv[1][0])[:40]:
            print(f"   {v[0]:4d} unpadded {v[1]:4d} padded  {k}")
            for ex in v[2][:2]:
                print("         ", ex)
        total += sum(v[0] for v in r.values())
    sys.exit(1 if total else 0)
