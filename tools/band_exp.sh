#!/bin/bash
# Development aid (GPU box): phase stamps of several -D variants of the band kernel
# (gpurun_scratch/<name>.so, built by tools/band_variant.sh).  usage: band_exp.sh B name...
mkdir -p gpurun_out
B=$1; shift
{
for v in "$@"; do
  echo "== $v"
  ASR_AMD_LIB=$PWD/gpurun_scratch/$v.so timeout -k 10 120 python tools/band_stamps.py $B 2>&1 | grep -v amdgpu.ids | head -5 || exit 1
done
} > gpurun_out/band_exp.log 2>&1
rc=$?
cat gpurun_out/band_exp.log
exit $rc
