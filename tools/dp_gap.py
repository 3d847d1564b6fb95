"""Data-parallel step on ONE GPU (bench.py --force-dist: RCCL group of one rank) from a rocprofv3 kernel_trace.csv: what sits between
the end of the gradient graph (its last kernel is the per-replica clip, scale_kernel) and the first kernel of the update graph.

usage: python tools/dp_gap.py <kernel_trace.csv> [out.txt]
Per timed step: idle time before the collective's kernel, the collective's duration (world size 1: a device copy), idle time after
it, and the whole exposed time between the two graphs; then the median over the steps."""
import csv, statistics, sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
gaps = []
for i, r in enumerate(rows):
    if not name(r).startswith("scale_kernel"):
        continue
    end_g1 = int(r["End_Timestamp"])
    seq, j = [], i + 1
    while j < len(rows) and not name(rows[j]).startswith(("adam_advance", "adam_kernel", "sumsq")):
        seq.append(rows[j]); j += 1
    if j >= len(rows):
        break
    start_g2 = int(rows[j]["Start_Timestamp"])
    coll = sum(int(k["End_Timestamp"]) - int(k["Start_Timestamp"]) for k in seq)
    before = (int(seq[0]["Start_Timestamp"]) - end_g1) if seq else 0
    after = (start_g2 - int(seq[-1]["End_Timestamp"])) if seq else start_g2 - end_g1
    gaps.append(((start_g2 - end_g1) / 1e3, before / 1e3, coll / 1e3, after / 1e3, [name(k)[:40] for k in seq]))
print("# exposed_us  idle_before_us  collective_kernels_us  idle_after_us  kernels between the graphs", file=out)
for g in gaps:
    print(f"{g[0]:9.1f} {g[1]:9.1f} {g[2]:9.1f} {g[3]:9.1f}  {g[4]}", file=out)
if gaps:
    tail = gaps[len(gaps) // 2:]          # the timed (graph-replay) steps come last
    print(f"# median over the last {len(tail)} steps: exposed {statistics.median(g[0] for g in tail):.1f} us = idle {statistics.median(g[1] for g in tail):.1f} "
          f"+ collective {statistics.median(g[2] for g in tail):.1f} + idle {statistics.median(g[3] for g in tail):.1f}", file=out)
