#!/bin/bash
# Copies what tools/profile_round.sh <tag> left under gpurun_out/<tag>/ (scratch) into profiles/ (tracked) under the
# names DESIGN.md quotes.   tools/collect_profiles.sh <tag> [round-prefix, default r04]
set -eo pipefail
TAG=${1:?tag}
R=${2:-r04}
S=gpurun_out/$TAG
D=profiles
cpif() { [ -f "$1" ] && cp "$1" "$2" && echo "  $2" || true; }
cpif $S/pmc_bench.json $D/pmc_bench.json
cpif $S/pmc_chain_all.json $D/pmc_chain.json
cpif $S/pmc_feeders.json $D/${R}_pmc_feeders.json
cpif $S/bench.json $D/${R}_final_bench.json
cpif $S/bench_under_rocprof.json $D/${R}_final_bench_under_rocprof.json
cpif $S/bench_full_under_rocprof.json $D/${R}_final_full_bench_under_rocprof.json
cpif $S/stats/stats_kernel_stats.csv $D/${R}_final_kernel_stats.csv
cpif $S/stats_full/stats_kernel_stats.csv $D/${R}_final_full_kernel_stats.csv
cpif $S/chain_stats/chain_kernel_stats.csv $D/${R}_chain_config5_kernel_stats.csv
cpif $S/spdsr_stats/spdsr_kernel_stats.csv $D/${R}_spdsr_kernel_stats.csv
cpif $S/chain_config5_fhd.json $D/${R}_chain_config5_fhd.json
cpif $S/chain_config5_vga.json $D/${R}_chain_config5_vga.json
cpif $S/spdsr_fhd.json $D/${R}_spdsr_fhd.json
cpif $S/spdsr_vga.json $D/${R}_spdsr_vga.json
cpif $S/mrf_fhd.json $D/${R}_mrf_fhd.json
cpif $S/mrf_vga.json $D/${R}_mrf_vga.json
cpif $S/feeders.json $D/${R}_feeders_128xfhd.json
cpif $S/shard_vs_bench.json $D/${R}_shard_replay_vs_bench_1gpu.json
cpif $S/sweep_k1_variants.log $D/${R}_sweep_k1_tiles.log
cpif $S/micro_pmc_a.txt $D/${R}_valu_microbench.txt
# (profiles/valu_costs.json needs the micro_pmc step -- cycles per opcode -- and is only copied from a round that ran it)
[ -d $S/micro_pmc ] && cpif $S/valu_costs.json $D/valu_costs.json
cpif $S/sweep_k1_windows.log $D/${R}_sweep_k1_windows.log
