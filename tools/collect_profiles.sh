#!/bin/bash
# Copy the outputs of a measurement pass (tools/measure_round.sh -> gpurun_out/final) into profiles/ under the round's tag.
T=${ROUND_TAG:-r6}
F=gpurun_out/final
P=profiles
cp $F/bench_line.json $P/${T}_bench_line.json
cp $F/bench_line_under_rocprof.json $P/${T}_bench_line_under_rocprof.json
cp $F/bench_kernel_stats.txt $P/${T}_bench_kernel_stats.txt
cp $F/bench_kernel_stats_one_pass.txt $P/${T}_bench_kernel_stats_one_pass.txt
cp $F/cfg2_kernel_stats.txt $P/${T}_cfg2_kernel_stats.txt
cp $F/pmc_counters.txt $P/${T}_pmc_counters.txt
cat $F/pmc_counters_cfg2.txt $F/pmc_counters_cfg3.txt > $P/${T}_pmc_counters_cfg2_cfg3.txt
python - <<PY
import json
out = {k: json.load(open("$F/bench_line_cfg%s_with_traffic.json" % k)) for k in ("2", "3")}
json.dump(out, open("$P/${T}_bench_lines_cfg2_cfg3.json", "w"), indent=1)
PY
cp $F/hbm_traffic.json $P/hbm_traffic.json
cp $F/other_configs.txt $P/${T}_other_configs.txt
cp $F/stress_vs_oracle.txt $P/${T}_stress_vs_oracle.txt
[ -f gpurun_out/parity_audit.txt ] && cp gpurun_out/parity_audit.txt $P/${T}_parity_audit.txt
sed -i 's|/tmp/code/[^ ]*/repo/||; s|/root/repo/||' $P/${T}_bench_kernel_stats.txt $P/${T}_bench_kernel_stats_one_pass.txt $P/${T}_cfg2_kernel_stats.txt
ls -la $P | grep ${T}_ | wc -l
