#!/usr/bin/env python3
"""Build-time check of the instruction sequence `grid_ordered_sum` (csrc/common.h) relies on, in every kernel that uses it.
The ordered grid-wide sum publishes a workgroup's partial with a RETURNING agent-scope atomic exchange, takes its ticket with an
agent-scope atomic add whose operand depends on the exchange's return value, and the last workgroup reads the partials with
agent-scope atomic loads.  All three are relaxed atomics (a release / acquire pair would write back and invalidate the XCD's L2:
36 -> 74 us on the loss kernel), so the ordering is an ISA-level property, not a C++-memory-model one:
  (1) the exchange must be a returning atomic (`global_atomic_swap ... sc0`): it completes at the memory side before its value returns;
  (2) an `s_waitcnt vmcnt(0)` must sit between the exchange and the ticket add (the wave waits for that return value);
  (3) the partial reads of the last workgroup must be `global_load_dword ... sc1` (agent scope: they bypass the non-coherent levels).
A toolchain that compiles the source differently fails the build here instead of silently breaking bitwise reproducibility.
usage: check_ordered_sum.py file.o [...]   (exit status 1 on a violation)"""
import re
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scan_pk_hazard as S


def check(path):
    text = S.disassemble(path)
    bad, seen = [], 0
    for name in re.findall(r"<(\S+)>:", text):
        i = text.index("<" + name + ">:")
        j = text.find("\n\n", i)
        lines = [l.split("//")[0].strip() for l in text[i:j].splitlines()]
        swaps = [k for k, l in enumerate(lines) if l.startswith("global_atomic_swap")]
        if not swaps:
            continue
        seen += 1
        for k in swaps:
            if " sc0" not in lines[k]:
                bad.append((name, "the exchange is not a returning atomic: " + lines[k]))
            adds = [q for q in range(k + 1, len(lines)) if lines[q].startswith("global_atomic_add")]
            if not adds:
                bad.append((name, "no ticket add behind the exchange"))
                continue
            between = lines[k + 1:adds[0]]
            if not any(re.match(r"s_waitcnt.*vmcnt\(0\)", l) for l in between):
                bad.append((name, "no s_waitcnt vmcnt(0) between the exchange and the ticket add"))
            tail = lines[adds[0]:]
            if not any(l.startswith("global_load_dword ") and " sc1" in l for l in tail):
                bad.append((name, "the partial reads behind the ticket are not agent-scope loads (sc1)"))
    return seen, bad


def main():
    total, allbad = 0, []
    for p in sys.argv[1:]:
        seen, bad = check(p)
        total += seen
        allbad += [(p,) + b for b in bad]
    for p, n, msg in allbad:
        print(f"{p}: {n}: {msg}")
    print(f"ordered-sum check: {total} kernels, {len(allbad)} violations")
    return 1 if allbad else 0


if __name__ == "__main__":
    sys.exit(main())
