#!/bin/bash
# Instrumented (timing-only) builds of libelfhip.so for the profiling scripts; never the product library.
#   benchmarks/_build_prof_lib.sh GO_PROF   -> benchmarks/_prof/libelfhip_GO_PROF.so
# The profiling scripts load it when ELFHIP_PROF_LIB names it (they patch elf_amd._lib.LIB_PATH; the product has no such switch).
set -e
DEF=$1
cd "$(dirname "$0")/.."
OUT=benchmarks/_prof
mkdir -p $OUT/obj_$DEF
DEFS=""; for d in ${DEF//,/ }; do DEFS="$DEFS -D$d"; done
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-inline-asm $DEFS"
pids=()
for s in elf_amd/csrc/*.hip; do
  o=$OUT/obj_$DEF/$(basename ${s%.hip}).o
  /opt/rocm/bin/hipcc $FLAGS -c $s -o $o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT/libelfhip_$DEF.so $OUT/obj_$DEF/*.o
rm -rf $OUT/obj_$DEF
ls -la $OUT/libelfhip_$DEF.so
