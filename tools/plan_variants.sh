#!/bin/bash
# Experiments on rounds 2-4's persistent plan kernel (k_plan_persistent, LAV_PLAN_IMPL=lds) as the known victim of the co-residency
# effect: variant builds of gru.hip linked with the other objects of the library, each run beside the tap-pair stem
# and the synthetic matrix + LDS neighbour with the LDS claims OFF.  Usage (GPU box): bash tools/plan_variants.sh [launches] > out.txt
set -u
N=${1:-200}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/plan_variants; mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -I $ROOT/include -I $ROOT/lav_amd/csrc"
OBJS=$(ls $ROOT/lav_amd/build/*.o | grep -v '/gru.o$')
run() {   # name, extra compile flags
  local name=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c $ROOT/lav_amd/csrc/gru.hip -o $OUT/gru_$name.o || { echo "$name: compile failed"; return; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $OUT/gru_$name.o -o $OUT/liblav_$name.so || { echo "$name: link failed"; return; }
  echo "== variant $name ($*)"
  LAV_AMD_LIB=$OUT/liblav_$name.so LAV_LDS_EXCLUSIVE=0 CORES_VICTIMS="k_plan_persistent" CORES_AGGRESSORS="stem,lds_hog" \
      python $ROOT/tools/coresidency.py $N 2>&1 | grep "wrong"
}
run base
run nosleep -DLAV_PLAN_VARIANT=1
run valu_sums -DLAV_PLAN_VARIANT=2
run nosleep_valu_sums -DLAV_PLAN_VARIANT=3
run forcezero -mllvm -amdgpu-waitcnt-forcezero
run lds_sync -DLAV_PLAN_LDS_SYNC=1
run lds_sync_valu_sums -DLAV_PLAN_LDS_SYNC=1 -DLAV_PLAN_VARIANT=2
