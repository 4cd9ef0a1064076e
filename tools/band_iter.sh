#!/bin/bash
# Development aid (GPU box): band-kernel tests, micro-benchmark at the bench shape, phase stamps.
mkdir -p gpurun_out
{
timeout -k 10 300 python -m pytest tests/test_lattice_gpu.py -x -q -m gpu -k "band or golden or seeded or sorted or property" 2>&1 | tail -5 &&
timeout -k 10 120 python tools/bench_lattice.py --cases mono_num --B 768 --iters 50 2>&1 | grep fwbw &&
timeout -k 10 120 python tools/bench_lattice.py --cases mono_num --B 512 --iters 50 2>&1 | grep fwbw &&
ASR_AMD_LIB=$PWD/gpurun_scratch/stamps.so timeout -k 10 120 python tools/band_stamps.py 768 2>&1 | head -6
} > gpurun_out/band_iter.log 2>&1
rc=$?
cat gpurun_out/band_iter.log
exit $rc
