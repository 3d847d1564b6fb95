"""Per-kernel FETCH_SIZE (KiB, doubled per MI355X_MICROARCH.md) of a rocprofv3 --pmc FETCH_SIZE counter_collection.csv, per launch geometry.
usage: python tools/pmc_fetch_kernel.py <counter_collection.csv> <kernel-name prefix> [steps]"""
import csv, sys, collections
tot = collections.Counter(); n = collections.Counter()
pref = sys.argv[2]; steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
for r in csv.DictReader(open(sys.argv[1])):
    if r.get("Counter_Name") != "FETCH_SIZE":
        continue
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if not k.startswith(pref):
        continue
    key = (k, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", ""))
    tot[key] += float(r["Counter_Value"]); n[key] += 1
for key in sorted(tot, key=lambda q: -tot[q]):
    print(f"{key[0][:48]:48s} grid {key[1]:>9s} launches/step {n[key] / steps:5.1f}  read {2 * tot[key] * 1024 / steps / 1e6:9.1f} MB/step")
print(f"total {2 * sum(tot.values()) * 1024 / steps / 1e6:.1f} MB/step")
