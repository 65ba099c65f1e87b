#!/bin/bash
# Kernel trace + PMC passes of the config-5 runner (one GPU's share: 4096 chains x N = 1024, one HMC iteration of
# L = 20 after the warm-up iteration), run on the GPU box from the repo root.  Summaries go to gpurun_out/cfg5_<tag>/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r2}
OUT=gpurun_out/cfg5_$TAG
CMD="python3 tools/bench_config5.py ${2:-4096} 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- $CMD > /dev/null 2>&1
# fabric traffic: FETCH_SIZE and WRITE_SIZE each in a pass of their own (TCC slots), GRBM_GUI_ACTIVE beside them
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/d -- $CMD > /dev/null 2>&1
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
python3 tools/pmc_by_kernel.py $OUT > $OUT/pmc_summary.txt
cat $OUT.log | grep kernel
head -12 $OUT/kernel_stats.csv
head -60 $OUT/pmc_summary.txt
