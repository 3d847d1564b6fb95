#!/bin/bash
# usage: tools/ab_kernel_stats.sh BASE.so ARCH PATTERN: rocprofv3 kernel statistics (one stream) of bench.py with the in-tree library and with USSEG_LIB=BASE.so;
# prints the average duration of the kernels whose name matches PATTERN for both - the per-kernel A/B of a source change on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BASE=$1; ARCH=$2; PAT=$3
for tag in base new; do
  if [ $tag = base ]; then export USSEG_LIB=$BASE; else unset USSEG_LIB; fi
  USSEG_LAZY_WGRAD=0 rocprofv3 --kernel-trace --stats -d gpurun_out/abks_$tag -o r --output-format csv -- python3 bench.py --arch $ARCH --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > gpurun_out/abks_$tag.log 2>&1
  echo "== $tag"; python3 - "$PAT" gpurun_out/abks_$tag/r_kernel_stats.csv <<'P'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[2])):
    if re.search(sys.argv[1], r["Name"]):
        print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:8.2f} us")
P
done
