#!/bin/bash
# PMC passes for the bench kernel (run on the GPU box through gpurun). Counters are collected in their own runs with
# --kernel-trace only, never together with sys/hip/hsa traces (MI355X_MICROARCH.md, rocprofv3 PMC slots).
R=${GRAFT_REPO_ROOT:-$(pwd)}
# PMC_CONFIG=2|3: the same for bench.py --config 2|3 (HBM traffic passes only), into gpurun_out/pmc_cfg2|3
CFG=${PMC_CONFIG:-1}
OUT=$R/gpurun_out/pmc
[ "$CFG" != "1" ] && OUT=$R/gpurun_out/pmc_cfg$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (one pipeline pass at a time: with concurrent passes a launch is several dispatches of every kernel, and the summary adds
# up the mean per dispatch of the kernels of ONE launch)
export SMRT_DORT_LANES=1
CMD="python $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-secondary"
run() { name=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1; }
run fetch FETCH_SIZE
run write WRITE_SIZE
[ "$CFG" != "1" ] && { find $OUT -name "*counter_collection.csv" | head; exit 0; }
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_UNALIGNED_STALL
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
# matrix-core use (names as listed by `rocprofv3 -L` on gfx950; a name the tool does not know only fails its own pass)
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $OUT/mfma_counter_names.txt
run mfma1 SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES
run mfma2 SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES
run mfma3 SQ_INSTS_MFMA SQ_INSTS_VALU
find $OUT -name "*counter_collection.csv" | head
