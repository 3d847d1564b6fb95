#!/bin/bash
# Regenerates the round's committed evidence on a GPU box FROM ONE BUILD: kernel statistics + step timeline (graph replay under
# the rocprofv3 kernel trace), PMC traffic (two separate --pmc passes, eager), and the bench JSON that reads that traffic file.
# usage: bash tools/profile_round.sh <round tag, e.g. r02> <version tag, e.g. v3> [arch B|A]
set -e
R=${1:-r02}; V=${2:-v1}; ARCH=${3:-B}
TAG=${R}_${V}_arch${ARCH}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out profiles
# per-kernel statistics on ONE stream (USSEG_LAZY_WGRAD=0: no weight-gradient launches running beside the kernel being timed - what bench.py's
# roofline leg measures too), then the step as shipped (lazy weight gradients on the side stream) for the timeline
USSEG_LAZY_WGRAD=0 rocprofv3 --kernel-trace --stats -d gpurun_out/stats_$TAG -o r --output-format csv -- python3 bench.py --arch $ARCH --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > gpurun_out/stats_$TAG.log 2>&1
cp gpurun_out/stats_$TAG/r_kernel_stats.csv profiles/${R}_kernel_stats_${V}_arch${ARCH}.csv
python3 tools/step_timeline.py gpurun_out/stats_$TAG/r_kernel_trace.csv profiles/${R}_step_timeline_1stream_${V}_arch${ARCH}.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/stats2_$TAG -o r --output-format csv -- python3 bench.py --arch $ARCH --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > gpurun_out/stats2_$TAG.log 2>&1
python3 tools/step_timeline.py gpurun_out/stats2_$TAG/r_kernel_trace.csv profiles/${R}_step_timeline_${V}_arch${ARCH}.txt
# eager run (--no-graph): 1 warm-up + 3 timed = 4 steps in the process
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f_$TAG -o r --output-format csv -- python3 bench.py --arch $ARCH --no-graph --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > gpurun_out/pmc_f_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w_$TAG -o r --output-format csv -- python3 bench.py --arch $ARCH --no-graph --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > gpurun_out/pmc_w_$TAG.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_f_$TAG/r_counter_collection.csv gpurun_out/pmc_w_$TAG/r_counter_collection.csv 4 profiles/${R}_pmc_traffic_arch${ARCH}.json $ARCH
# MFMA utilisation (north_star: "rocprof HBM GB/s and MFMA utilisation against peak"): its own --pmc pass, eager, 4 steps
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d gpurun_out/pmc_m_$TAG -o r --output-format csv -- python3 bench.py --arch $ARCH --no-graph --steps 3 --warmup 1 --no-cpu-baseline --profile-steps 0 > gpurun_out/pmc_m_$TAG.log 2>&1
python3 tools/pmc_mfma.py gpurun_out/pmc_m_$TAG/r_counter_collection.csv 4 profiles/${R}_pmc_mfma_arch${ARCH}.json $ARCH
python3 bench.py --arch $ARCH > profiles/${R}_bench_${V}_arch${ARCH}.json 2> gpurun_out/bench_$TAG.err
mkdir -p gpurun_out/profiles && cp profiles/${R}_*_${V}_arch${ARCH}.* profiles/${R}_pmc_traffic_arch${ARCH}.json profiles/${R}_pmc_mfma_arch${ARCH}.json gpurun_out/profiles/   # gpurun merges only gpurun_out/ back: copy these into profiles/ and commit
tail -3 gpurun_out/bench_$TAG.err; tail -6 profiles/${R}_step_timeline_${V}_arch${ARCH}.txt
