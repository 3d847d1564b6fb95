#!/bin/bash
# Builds diagnostic variants of libusseg_hip.so that differ ONLY in how pointwise.hip is compiled, and runs tools/diag_norm_load.py on each
# (one child process per variant; USSEG_LIB selects the library).  Question: which instruction sequence makes the
# grouped LayerNorm's results depend on what else the GPU is running?
#   pk_plain   packed-fp32 code generation, C++ ladder   (ds_bpermute x3 -> s_waitcnt lgkmcnt(1) -> v_pk_add_f32 on the returned pair)
#   pk_wait0   packed-fp32, every returned value consumed by an empty asm first (=> s_waitcnt lgkmcnt(0) before the v_pk_add_f32)
#   pk_scalar  packed-fp32 everywhere EXCEPT the ladder's adds (single v_add_f32)
#   pk_noocc   packed-fp32, C++ ladder, no occupancy bound on the kernel (the register-allocation suspect)
#   nopk       no packed-fp32 instructions at all (round 2's workaround)
# usage (on the GPU box): bash tools/diag_norm_variants.sh [reps]
set -e
cd "$(dirname "$0")/.."
REPS=${1:-20}
CS=ultrasound_modeling_amd/csrc
OUT=gpurun_out/normvar
mkdir -p $OUT
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Iinclude -I$CS"
NOPK="-Xclang -target-feature -Xclang -packed-fp32-ops"
OBJS=$(ls $CS/*.o | grep -v pointwise.o)
build() {  # name, extra flags
  [ -f $OUT/lib_$1.so ] && return
  $HIPCC $FL $2 -c $CS/pointwise.hip -o $OUT/pw_$1.o 2>/dev/null
  $HIPCC -shared -fPIC --offload-arch=gfx950 $OBJS $OUT/pw_$1.o -o $OUT/lib_$1.so
}
build pk_plain  "-DNORM_LADDER=0"
build pk_wait0  "-DNORM_LADDER=1"
build pk_scalar "-DNORM_LADDER=2"
build pk_noocc  "-DNORM_LADDER=0 -DNORM_NO_OCC"
build nopk      "-DNORM_LADDER=0 $NOPK"
build pk_sums   "-DNORM_LADDER=0 -DNORM_SUMS_SCALAR"      # packed everywhere except the per-group sum / sum-of-squares FMA chains
build noslp     "-DNORM_LADDER=0 -fno-slp-vectorize"      # no SLP vectoriser: the <2 x float> operations the packed instructions come from are never formed
if [ "$2" = "build-only" ]; then exit 0; fi
for v in ${VARIANTS:-pk_plain pk_wait0 pk_scalar pk_noocc nopk pk_sums noslp}; do
  for load in both gemm stream; do
    [ $v != pk_plain ] && [ $load != both ] && continue
    echo "== $v load=$load: $(USSEG_LIB=$PWD/$OUT/lib_$v.so python3 tools/diag_norm_load.py 16 128 128 30 3 $REPS $load 2>&1 | tail -1)"
  done
done
