R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_ic
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d $OUT/ic -o ic -- $CMD > $OUT/ic.log 2>&1
timeout 600 rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/if -o if -- $CMD > $OUT/if.log 2>&1
tail -3 $OUT/ic.log
