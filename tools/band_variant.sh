#!/bin/bash
# Development aid: builds a variant of libasr_amd.so with a different lattice.hip object
# (extra -D flags) into gpurun_scratch/<name>.so — the other objects come from csrc/build.
# usage: tools/band_variant.sh name [-DBAND_STAMPS ...]
set -e
cd "$(dirname "$0")/../pytorch-asr_amd/csrc"
name=$1; shift
make -s all
mkdir -p ../../gpurun_scratch
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function \
    -fno-slp-vectorize -Wno-undefined-internal "$@" -c lattice.hip -o ../../gpurun_scratch/$name.lattice.o
objs=$(ls build/*.o | grep -v 'build/lattice.hip.o')
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_scratch/$name.so $objs ../../gpurun_scratch/$name.lattice.o
echo built gpurun_scratch/$name.so
