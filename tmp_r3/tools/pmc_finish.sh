#!/bin/bash
# PMC passes over the headline batch for one pipeline shape (default: whatever the library picks), kernel-trace only.
#   tools/pmc_finish.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-pmcf}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT SQ_INST_LEVEL_VMEM SQ_IFETCH
run mfma SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run fetch FETCH_SIZE
run write WRITE_SIZE
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- $CMD > $OUT/stats.log 2>&1
python $R/tools/pmc_summary.py $OUT > $OUT/summary.txt 2>&1
tail -40 $OUT/summary.txt
