set -e
mkdir -p gpurun_out
{
for a in B A T; do python bench.py --arch $a --steps 30 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a stage-lazy', d['ms_per_step'])"; done
for a in B A T; do USSEG_ENC_LAZY=0 python bench.py --arch $a --steps 30 --warmup 5 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$a stage-lazy off', d['ms_per_step'])"; done
python tools/diag_load_race.py 8 256 10 1 1 B
python tools/diag_load_race.py 8 256 10 1 1 A
python tools/diag_load_race.py 8 256 10 1 0 B
} > gpurun_out/lazy4.log 2>&1
