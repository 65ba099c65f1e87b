#!/bin/bash
# PMC passes for the dominant kernel (run on the GPU box from the repo root).  Counters go in their own runs
# (no tracing flags), one pass per counter group, as the rocprofv3 section of MI355X_MICROARCH.md prescribes.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_${1:-r1}
# every dispatch of the dominant kernel is one launch of 25 HMC iterations, the bench default (warm-up 25 = one launch,
# 50 steps = two, the event-timed section ten more)
CMD="python3 bench.py --steps 50 --warmup 25 --iters-per-launch 25 --no-cpu-baseline --no-compare"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $OUT/a -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/b -- $CMD > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/c -- $CMD > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/d -- $CMD > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_ANY SQ_LEVEL_WAVES SQ_WAVES --output-format csv -d $OUT/e -- $CMD > /dev/null 2>&1
python3 tools/pmc_summary.py $OUT "k_mfma32<0" $OUT/pmc_latest.json 25 k_stats_update_run > $OUT/summary.txt
cat $OUT/summary.txt
