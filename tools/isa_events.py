#!/usr/bin/env python3
"""Order of memory / matrix / wait events in the gfx950 ISA of a built object: what a kernel's loops really overlap.
usage: isa_events.py file.o [kernel-name-substring] [--width N]
Per kernel one line of run-length-compressed events in program order:
  L global / buffer / flat load   S global store   X scratch access (spill)   r LDS read   s LDS write   M MFMA
  W<n> s_waitcnt vmcnt(n)         | s_barrier
A prefetch that is meant to fly under a compute phase shows as `L.. M.. W..`; `L W0` or `L W<n> ... M` means the loads are waited for before
the MFMAs start (found this way: a `v_cndmask` on a loaded value - `if (!ok) v = 0` - pins the wait behind the load; so does a branch
around the load).  An `X` inside a loop is a register spill."""
import os
import re
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import scan_pk_hazard as S


def events(lines):
    ev = []
    for l in lines:
        if "global_load" in l or "buffer_load" in l or "flat_load" in l:
            ev.append("L")
        elif "global_store" in l or "buffer_store" in l or "flat_store" in l:
            ev.append("S")
        elif "scratch_" in l:
            ev.append("X")
        elif "mfma" in l:
            ev.append("M")
        elif "s_waitcnt" in l and "vmcnt" in l:
            ev.append("W" + re.search(r"vmcnt\((\d+)\)", l).group(1) + " ")
        elif "s_barrier" in l:
            ev.append("|")
        elif "ds_write" in l or "ds_store" in l:
            ev.append("s")
        elif "ds_read" in l or "ds_load" in l:
            ev.append("r")
    out = []
    for e in ev:
        if out and out[-1][0] == e:
            out[-1][1] += 1
        else:
            out.append([e, 1])
    return " ".join(f"{e}{c if c > 1 else ''}" for e, c in out)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    width = int(sys.argv[sys.argv.index("--width") + 1]) if "--width" in sys.argv else 600
    args = [a for a in args if a != str(width) or "--width" not in sys.argv]
    text = S.disassemble(args[0])
    key = args[1] if len(args) > 1 else ""
    for name in re.findall(r"<(\S+)>:", text):
        if key not in name or name.startswith(".") or "__device_stub" in name:
            continue
        i = text.index("<" + name + ">:")
        j = text.find("\n\n", i)
        lines = [l.split("//")[0].rstrip() for l in text[i:j].splitlines()]
        if len(lines) < 30:
            continue
        print(f"{name}  ({len(lines)} instructions)\n   {events(lines)[:width]}")


if __name__ == "__main__":
    main()
