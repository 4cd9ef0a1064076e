#!/bin/bash
# Issue/wait counters of the mono-char lattice kernels (gpurun -- bash tools/refresh_lattice_counters.sh)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/latc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_TRANS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/lattice_counters.py > $O/p$i.log 2>&1 || echo "pass $i failed: $(tail -3 $O/p$i.log)"
done
python3 $R/tools/pmc_counters.py $O/r03_pmc_lattice_issue.json lattice_fwbw $O/p1 $O/p2 $O/p3 $O/p4
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
