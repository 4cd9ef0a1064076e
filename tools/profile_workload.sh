#!/bin/bash
# Per-kernel time of one bench.py workload (gpurun -- bash tools/profile_workload.sh <name> <bench args...>)
# -> gpurun_out/prof/<name>.txt (the top kernels) and <name>_kernel_stats.csv
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof
N=$1; shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw_$N -- python3 $R/bench.py "$@" --steps 6 --warmup 2 --no-cpu-baseline --no-extra > $O/$N.log 2>&1
python3 $R/tools/kstats.py $O/raw_$N 8 $O/${N}_kernel_stats.csv 26 > $O/$N.txt
rm -rf $O/raw_$N
cat $O/$N.txt
