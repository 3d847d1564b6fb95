"""Turn one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES) of `bench.py --no-graph` into
profiles/rNN_pmc_mfma_arch{X}.json: MFMA utilisation per kernel and per kernel family.

usage: python tools/pmc_mfma.py <counter_collection.csv> <steps in the run> <out.json> [arch]
SQ_VALU_MFMA_BUSY_CYCLES counts, per SIMD, the cycles its matrix pipe is busy (16 per v_mfma_f32_16x16x32_bf16, 32 per 32x32x16:
MI355X_MICROARCH.md "s_memtime tick vs SQ PMC units"), summed over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is reported as the sum over the
8 XCDs, so a dispatch lasted GRBM_GUI_ACTIVE / 8 cycles.  mfma_busy_frac = MFMA busy cycles / (1024 SIMDs x dispatch cycles) is the
fraction of the dense MFMA issue peak the kernel used while it ran (the counterpart of algorithmic TFLOP/s / 2500 measured from the inside:
it includes padded / masked work and is independent of the clock)."""
import collections
import csv
import json
import sys

rows = collections.defaultdict(lambda: collections.Counter())
n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        n[k] += 1
steps = int(sys.argv[2])
arch = sys.argv[4] if len(sys.argv) > 4 else "B"


def family(k):
    if k.startswith(("igemm", "conv_halo", "conv_big", "conv_stream", "cardinal", "stem_fwd", "dgrad_actbwd", "head_quad_loss")):
        return "conv forward / backward-data"
    if k.startswith("wgrad") and "finish" not in k:
        return "weight gradients"
    if k.startswith(("flash", "window_attn")):
        return "attention"
    return "other (no MFMA: norms, reductions, optimiser, copies)"


out = {"arch": arch, "steps_profiled": steps, "kernels": {}, "families": {}}
fam = collections.defaultdict(lambda: [0.0, 0.0])
tot = [0.0, 0.0]
for k, c in sorted(rows.items()):
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"]
    out["kernels"][k] = {"launches_per_step": n[k] / steps, "dispatch_cycles_per_step": cyc / steps, "mfma_busy_cycles_per_step": busy / steps,
                         "mfma_busy_frac": round(busy / (1024.0 * cyc), 4) if cyc else None,
                         "sq_busy_frac": round(c["SQ_BUSY_CYCLES"] / (8.0 * cyc), 4) if cyc and "SQ_BUSY_CYCLES" in c else None}
    f = fam[family(k)]
    f[0] += busy
    f[1] += cyc
    tot[0] += busy
    tot[1] += cyc
for k, (b, c) in fam.items():
    out["families"][k] = {"mfma_busy_frac": round(b / (1024.0 * c), 4) if c else None, "dispatch_cycles_per_step": c / steps}
out["whole_step_mfma_busy_frac"] = round(tot[0] / (1024.0 * tot[1]), 4) if tot[1] else None
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({"families": out["families"], "whole_step_mfma_busy_frac": out["whole_step_mfma_busy_frac"]}))
