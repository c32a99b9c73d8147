#!/bin/bash
# tools/build_variant.sh TAG "-DMACRO=.. ..." -- an A/B build of the engine into build_ab/TAG/librpa_hip.so (git-ignored;
# select it with RPA_LIB_PATH).  Same flags as taxator-tk_amd/csrc/Makefile plus the given defines.
set -e
TAG=$1; DEFS=$2
cd "$(dirname "$0")/../taxator-tk_amd/csrc"
OUT=../../build_ab/$TAG
mkdir -p $OUT
for f in seq_kernels align_dna_kernels align_protein_kernels place_kernels lca_kernels rpa_engine; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $DEFS -c $f.hip -o $OUT/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/librpa_hip.so $OUT/*.o
rm -f $OUT/*.o
