#!/bin/bash
# Regenerates the round's committed evidence on a GPU box: kernel stats, PMC traffic, bench JSON.  usage: bash tools/profile_round.sh <tag>
set -e
TAG=${1:-r01_v13}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/stats_$TAG -o r --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > gpurun_out/stats_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f_$TAG -o r --output-format csv -- python3 bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > gpurun_out/pmc_f_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w_$TAG -o r --output-format csv -- python3 bench.py --no-graph --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > gpurun_out/pmc_w_$TAG.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_f_$TAG/r_counter_collection.csv gpurun_out/pmc_w_$TAG/r_counter_collection.csv 4 gpurun_out/pmc_traffic_$TAG.json
cp gpurun_out/pmc_traffic_$TAG.json profiles/r01_pmc_traffic.json
python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
tail -2 gpurun_out/bench_$TAG.err; ls gpurun_out/stats_$TAG
