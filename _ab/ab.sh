#!/bin/bash
# usage: ab.sh ENVVAR steps  -> alternates ENVVAR=1/0 twice
set -o pipefail
for v in 1 0 1 0; do
  env $1=$v python3 bench.py --steps ${2:-200} --warmup 20 --no-cpu-baseline --profile-steps 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', $v, d['ms_per_step'])" || exit 1
done
