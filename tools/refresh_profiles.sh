#!/bin/bash
# Regenerates the measurements behind profiles/r03_* on the GPU box (run through gpurun from
# the repo root: `gpurun --timeout 1200 -- bash tools/refresh_profiles.sh`).  The raw rocprofv3
# output (>64 MiB) is summarised on the box by tools/kstats.py, tools/pmc_summary.py and
# tools/mfma_busy.py; copy gpurun_out/refresh/r03_* into profiles/ afterwards.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 python3 $R/bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/stats.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > $O/mfma.log 2>&1
B=$(python3 -c "import json,sys; print(json.load(open('$O/bench.json'))['config']['batch_per_gpu'])")
cp $O/bench.json $O/r03_bench_b$B.json
python3 $R/tools/kstats.py $O/stats 10 $O/r03_bench_b${B}_kernel_stats.csv 30 > $O/kstats.txt
python3 $R/tools/pmc_summary.py $O/fetch $O/write $O/r03_pmc_step_fetch_write.json --batch $B
python3 $R/tools/mfma_busy.py $O/mfma $O/r03_pmc_mfma_busy.json
rm -rf $O/stats $O/fetch $O/write $O/mfma
cat $O/kstats.txt
tail -c 400 $O/bench.json
