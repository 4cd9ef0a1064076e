#!/bin/bash
# Development aid (GPU box): A/B timing of libasr_amd.so variants in ONE call (boxes differ by
# several per cent).  usage: band_ab.sh name...   (gpurun_scratch/<name>.so; "cur" = the in-tree library)
mkdir -p gpurun_out
{
for rep in 1 2; do
for v in "$@"; do
  lib=$PWD/gpurun_scratch/$v.so; [ "$v" = cur ] && lib=$PWD/pytorch-asr_amd/csrc/libasr_amd.so
  echo -n "$v: "
  ASR_AMD_LIB=$lib timeout -k 10 120 python tools/bench_lattice.py --cases mono_num --B 768 --iters 200 2>&1 | grep fwbw || exit 1
done
done
} > gpurun_out/band_ab.log 2>&1
rc=$?
cat gpurun_out/band_ab.log
exit $rc
