"""Summarise a rocprofv3 kernel_trace.csv: per (kernel, grid) totals for the LAST full training step."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('adam_kernel')]
step = rows[idx[-3] + 1: idx[-2] + 1] if len(idx) >= 3 else rows
agg = collections.OrderedDict()
for r in step:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')
    g = (int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = (n, g)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1; a[1] += d
tot = sum(a[1] for a in agg.values())
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
print(f"{len(step)} dispatches, busy {tot/1e3:.3f} ms, span {(t1-t0)/1e6:.3f} ms")
bykernel = collections.Counter()
for (n, g), a in agg.items(): bykernel[n] += a[1]
for n, t in bykernel.most_common(12): print(f"  {n[:44]:44s} {t/1e3:7.3f} ms")
print("top (kernel, grid):")
for (n, g), a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"  {n[:36]:36s} grid={str(g):18s} x{a[0]:<3d} {a[1]:8.1f} us  ({a[1]/a[0]:.1f} each)")
