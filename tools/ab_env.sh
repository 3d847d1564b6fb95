#!/bin/bash
# usage: tools/ab_env.sh ENVVAR [steps]: bench.py with ENVVAR=1,0,1,0 in one call (one box), prints ms per step - the same-box A/B behind the numbers in DESIGN.md 4c
set -o pipefail
for v in 1 0 1 0; do
  env $1=$v python3 bench.py --steps ${2:-200} --warmup 20 --no-cpu-baseline --profile-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', $v, d['ms_per_step'])" || exit 1
done
