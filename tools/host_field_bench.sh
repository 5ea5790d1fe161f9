#!/bin/bash
# builds tools/host_field_bench.hip against the objects of the library (make -C lambdaworks_kzg_amd/csrc first) and runs it
set -e
cd "$(dirname "$0")/.."
B=lambdaworks_kzg_amd/build
make -s -C lambdaworks_kzg_amd/csrc -j8
OUT=${TMPDIR:-/tmp}/host_field_bench
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DLWK_LIMB_BITS=28 -Ilambdaworks_kzg_amd/csrc -c tools/host_field_bench.hip -o $OUT.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 --hip-link $OUT.o $(ls $B/*.o | grep -v "/pairing.o") -o $OUT
exec $OUT
