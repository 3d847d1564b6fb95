"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --no-graph` into profiles/rNN_pmc_traffic_arch{B,A}.json.

usage: python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <steps in the run> <out.json> [arch] [per-GPU batch]
`steps in the run` = every train step the profiled process executed (bench.py --no-graph: warmup + steps, with
--profile-steps 0).  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request for wide coalesced
reads, so the read side is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Both counters are in KiB."""
import csv, json, sys, collections

def load(path, counter):
    tot = collections.Counter(); n = collections.Counter()
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n

fetch, nf = load(sys.argv[1], "FETCH_SIZE")
write, nw = load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3])
arch = sys.argv[5] if len(sys.argv) > 5 else "B"
batch = int(sys.argv[6]) if len(sys.argv) > 6 else {"B": 16, "A": 32, "T": 8, "S": 16}[arch]
# the launches bench.py times as the conv family (usseg_prof kind 1): every conv kernel + the fused cardinal / stem launches that carry convs
fam = lambda k: k.startswith(("igemm", "conv_halo", "conv_big", "conv_stream", "cardinal_fwd", "cardinal_bwd", "stem_fwd", "dgrad_actbwd", "head_quad_loss"))
out = {"arch": arch, "per_gpu_batch": batch, "hw": 256, "steps_profiled": steps, "kernels": {}}
cb = cl = 0.0
for k in sorted(set(fetch) | set(write)):
    rd = 2.0 * fetch[k] * 1024 / steps      # gfx950: FETCH_SIZE reads exactly half of a wide coalesced stream
    wr = write[k] * 1024 / steps
    out["kernels"][k] = {"launches_per_step": nf[k] / steps, "read_bytes_per_step": rd, "write_bytes_per_step": wr}
    if fam(k):
        cb += rd + wr; cl += nf[k] / steps
out["conv_family_bytes_per_step"] = cb
out["conv_family_launches_per_step"] = cl
out["total_bytes_per_step"] = sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for v in out["kernels"].values())
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps({k: out[k] for k in ("conv_family_bytes_per_step", "conv_family_launches_per_step", "total_bytes_per_step")}))
