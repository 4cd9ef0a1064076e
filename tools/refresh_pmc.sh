#!/bin/bash
# The two HBM-traffic passes of tools/refresh_profiles.sh alone (gpurun -- bash tools/refresh_pmc.sh).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/refresh
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/write.log 2>&1
python3 $R/tools/pmc_summary.py $O/fetch $O/write $O/r03_pmc_step_fetch_write.json --batch 768
rm -rf $O/fetch $O/write
