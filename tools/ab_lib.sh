#!/bin/bash
# usage: tools/ab_lib.sh BASE.so ARCH STEPS REPEATS: bench.py with the in-tree library and with USSEG_LIB=BASE.so (a build of an earlier source state),
# interleaved in one call (one box) - the same-box A/B of a source change
set -o pipefail
BASE=$1; ARCH=$2; STEPS=$3; REP=$4
for r in $(seq $REP); do
  for lib in "$BASE" ""; do
    USSEG_LIB=$lib python3 bench.py --arch $ARCH --steps $STEPS --warmup 20 --no-cpu-baseline --profile-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('arch $ARCH lib [${lib:-in-tree}]', d['ms_per_step'])" || exit 1
  done
done
