#!/bin/bash
# One measurement pass for profiles/ (run on the GPU box through gpurun): bench line, rocprofv3 kernel-trace summary of
# the same command, PMC passes, the other configurations, the randomised parity sweeps.  Output under gpurun_out/final.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
cd $R
python bench.py > $O/bench_line.json 2> $O/bench_line.err
( cd /tmp; rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-secondary > $O/bench_line_under_rocprof.json 2> $O/rocprof.err )
python tools/rocpd_summary.py $(find $O/prof -name "*.db" | head -1) > $O/bench_kernel_stats.txt 2>&1
# the same with ONE pipeline pass at a time (the default runs three concurrently: their kernels overlap and stretch each other)
( cd /tmp; SMRT_DORT_LANES=1 rocprofv3 --kernel-trace --stats -d $O/prof1 -o bench -- python $R/bench.py --no-cpu-baseline --no-secondary > $O/bench_line_under_rocprof_one_pass.json 2> $O/rocprof1.err )
python tools/rocpd_summary.py $(find $O/prof1 -name "*.db" | head -1) > $O/bench_kernel_stats_one_pass.txt 2>&1
( cd /tmp; rocprofv3 --kernel-trace --stats -d $O/prof2 -o bench -- python $R/bench.py --config 2 --no-cpu-baseline --no-secondary > $O/bench_line_cfg2_under_rocprof.json 2> $O/rocprof2.err )
python tools/rocpd_summary.py $(find $O/prof2 -name "*.db" | head -1) > $O/cfg2_kernel_stats.txt 2>&1
bash tools/pmc_passes.sh > $O/pmc_passes.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc ${ROUND_TAG:-r6} > $O/pmc_counters.txt 2>&1
# the other BASELINE shapes as driver-style lines (bench.py --config 2 | 3) with their own HBM-traffic passes
python bench.py --config 2 > $O/bench_line_cfg2.json 2> $O/bench_line_cfg2.err
python bench.py --config 3 > $O/bench_line_cfg3.json 2> $O/bench_line_cfg3.err
PMC_CONFIG=2 bash tools/pmc_passes.sh > $O/pmc_passes_cfg2.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_cfg2 ${ROUND_TAG:-r6} 2 7168 > $O/pmc_counters_cfg2.txt 2>&1
PMC_CONFIG=3 bash tools/pmc_passes.sh > $O/pmc_passes_cfg3.log 2>&1
python tools/pmc_summary.py gpurun_out/pmc_cfg3 ${ROUND_TAG:-r6} 3 512 > $O/pmc_counters_cfg3.txt 2>&1
python bench.py --config 2 --no-secondary > $O/bench_line_cfg2_with_traffic.json 2>> $O/bench_line_cfg2.err
python bench.py --config 3 --no-secondary > $O/bench_line_cfg3_with_traffic.json 2>> $O/bench_line_cfg3.err
cp profiles/hbm_traffic.json $O/hbm_traffic.json 2>/dev/null
{
  echo "# configs[2] shape (DMRT-QCA-SR, 50 layers, 64 streams, 7 AMSR2 frequencies): 64 / 256 / 1024 snowpacks"
  python tools/bench_cfg3.py 64; python tools/bench_cfg3.py 256; python tools/bench_cfg3.py 1024
  echo "# configs[3] shape (IBA active, 30 layers, 128 streams, m_max 2): 512 snowpacks"
  python tools/bench_cfg4.py 512
  echo "# passive 128 / 192 streams, active laws"
  python tools/bench_large_n.py 2>&1 | tail -n 4
  python tools/bench_active.py 2>&1 | tail -n 6
} > $O/other_configs.txt 2>&1
{
  for s in 1 2 3 4 5 6; do python tools/stress_vs_oracle.py $s 2>&1 | tail -n 1; done
  for s in 11 12 13; do python tools/stress_vs_oracle.py $s prune 2>&1 | tail -n 2; done
  for s in 51 52 53 54; do python tools/stress_vs_oracle.py $s coherent 2>&1 | tail -n 2; done
  for s in 71 72 73; do python tools/stress_vs_oracle.py $s wetmicro 2>&1 | tail -n 2; done
  for s in 81 82; do python tools/stress_vs_oracle.py $s family 2>&1 | tail -n 2; done
  for s in 91 92; do python tools/stress_vs_oracle.py $s sce 2>&1 | tail -n 2; done
  echo '# hard passive inputs (tools/stress_reg_extremes.py): register-resident pipeline, then the global-workspace pipeline'
  for s in 1 2 3 4; do python tools/stress_reg_extremes.py $s 2>&1 | tail -n 1; done
  python tools/stress_reg_extremes.py 5 4 big 2>&1 | tail -n 1
} > $O/stress_vs_oracle.txt 2>&1
ls -la $O
