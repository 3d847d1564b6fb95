#!/bin/bash
# usage: tools/ab_env_kernel_stats.sh ENVVAR ARCH PATTERN: rocprofv3 kernel statistics (one stream) of bench.py with ENVVAR=0 and =1;
# prints the average duration of the kernels whose name matches PATTERN - the per-kernel A/B of a run-time switch on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
  export $1=$v
  USSEG_LAZY_WGRAD=0 rocprofv3 --kernel-trace --stats -d gpurun_out/abes_$v -o r --output-format csv -- python3 bench.py --arch $2 --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > gpurun_out/abes_$v.log 2>&1
  echo "== $1=$v"; python3 - "$3" gpurun_out/abes_$v/r_kernel_stats.csv <<'P'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[2])):
    if re.search(sys.argv[1], r["Name"]):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.2f} us")
P
done
