#!/bin/bash
# usage: tools/ab_sets.sh ARCH STEPS REPEATS "A=1 B=0" "A=0 B=0" ...: bench.py under each environment set, interleaved REPEATS times in one call
# (one box), prints ms per step - the same-box A/B of several switches at once
set -o pipefail
ARCH=$1; STEPS=$2; REP=$3; shift 3
for r in $(seq $REP); do
  for set in "$@"; do
    env $set python3 bench.py --arch $ARCH --steps $STEPS --warmup 20 --no-cpu-baseline --profile-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('arch $ARCH [$set]', d['ms_per_step'])" || exit 1
  done
done
