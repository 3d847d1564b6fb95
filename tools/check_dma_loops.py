#!/usr/bin/env python3
"""Build-time check of the kernels whose K loops order an LDS-DMA stage ring by hand (csrc/wgrad.hip wgrad_dma_kernel: the DMA pieces are
inline assembly - `s_mov_b32 m0` + `buffer_load_dwordx4 ... lds` - that the compiler's vmcnt bookkeeping does not see, and the loop's own
`s_waitcnt vmcnt(pieces of the younger stages)` + `s_barrier` are the only waits).  The scheme holds only while the compiler emits NO vector
memory load of its own between the first and the last MFMA of the kernel (it would be counted by the hand-written vmcnt waits and could be
waited for too early or too late) - nothing in the source enforces that, so the build does:
  every load between the first and the last v_mfma of a checked kernel must be `buffer_load_dwordx4 ... lds`.
usage: check_dma_loops.py file.o [...]   (exit status 1 on a violation)"""
import os
import re
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scan_pk_hazard as S

CHECKED = ("wgrad_dma_kernel",)


def check(path):
    text = S.disassemble(path)
    bad, seen = [], 0
    for name in re.findall(r"<(\S+)>:", text):
        if not any(c in name for c in CHECKED):
            continue
        i = text.index("<" + name + ">:")
        j = text.find("\n\n", i)
        lines = [l.split("//")[0].strip() for l in text[i:j if j > 0 else len(text)].splitlines()]
        mf = [k for k, l in enumerate(lines) if l.startswith("v_mfma")]
        if not mf:
            continue
        seen += 1
        for l in lines[mf[0]:mf[-1] + 1]:
            if re.match(r"(global_load|flat_load|scratch_load|buffer_load)", l) and not (l.startswith("buffer_load_dwordx4") and l.rstrip().endswith("lds")):
                bad.append((name, "a compiler-emitted load inside the hand-ordered K loop: " + l))
    return seen, bad


if __name__ == "__main__":
    total, violations = 0, []
    for p in sys.argv[1:]:
        n, bad = check(p)
        total += n
        violations += bad
    for name, why in violations:
        print(f"{name}: {why}")
    print(f"dma-loop check: {total} kernels, {len(violations)} violations")
    sys.exit(1 if violations else 0)
