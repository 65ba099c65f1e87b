#!/bin/bash
# Kernel trace + PMC passes (same counter set as the headline kernel's, tools/pmc_passes.sh) of k_fused16 on the headline
# model in f64 (the reference's default dtype) and on MLP(4-64-64-3) in f32: 4096 chains, N = 150, HMC L = 20, launches of
# five iterations.  Run on the GPU box from the repo root; summaries go to gpurun_out/fused16_<tag>/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r3}
OUT=gpurun_out/fused16_$TAG
mkdir -p $OUT
CMD="python3 tools/bench_fused16.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/run.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/b -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/d -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/e -- $CMD > /dev/null 2>&1
cp $(ls $OUT/trace/*/*kernel_stats.csv | head -1) $OUT/kernel_stats.csv
grep TFLOP $OUT/run.log > $OUT/pmc_summary.txt
python3 tools/pmc_by_kernel.py $OUT | grep -A 40 "k_fused16" >> $OUT/pmc_summary.txt
head -5 $OUT/kernel_stats.csv | cut -c1-160
cat $OUT/pmc_summary.txt | cut -c1-170
