#!/bin/bash
# Collects the profiling evidence of a round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>     ->  gpurun_out/<tag>/...
# 1. PMC passes (one counter group each, no tracing domains mixed in) -> pmc_traffic.json (also written into
#    profiles/ of the box copy so that the bench lines below carry the traffic / VALU figures of THIS build)
# 2. rocprofv3 --kernel-trace --stats over the default bench command
# 3. the plain bench line
set -eo pipefail
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
SHORT="bench.py --steps 3 --warmup 1 --cpu-seconds 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc/fetch" -o fetch -- python3 $SHORT > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc/write" -o write -- python3 $SHORT > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc/sq" -o sq -- python3 $SHORT --no-extra > "$OUT/pmc_sq.log" 2>&1
python3 tools/collect_pmc.py --dir "$OUT/pmc" --out "$OUT/pmc_traffic.json" 2> "$OUT/pmc_summary.log"
cp "$OUT/pmc_traffic.json" profiles/pmc_traffic.json
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o stats -- python3 bench.py --cpu-seconds 0 > "$OUT/bench_under_rocprof.json" 2> "$OUT/bench_under_rocprof.err"
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
find "$OUT/pmc" -name "*.csv" ! -name "*counter_collection.csv" -delete
cat "$OUT/pmc_summary.log" "$OUT/bench.json"
# 4. BASELINE config 5 (full chain) per stage and per kernel, config 3 (tile sweep) and the other windows
python3 tools/bench_chain.py > "$OUT/chain_config5_fhd.json" 2> "$OUT/chain.err"
python3 tools/bench_chain.py --width 640 --height 480 > "$OUT/chain_config5_vga.json" 2>> "$OUT/chain.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/chain_stats" -o chain -- python3 tools/bench_chain.py > /dev/null 2>> "$OUT/chain.err"
{
  python3 tools/sweep_jbf.py --width 1920 --height 1080 --frames 8 --window 19 --with-generic
  python3 tools/sweep_jbf.py --width 640 --height 480 --frames 64 --window 19
  python3 tools/sweep_jbf.py --width 640 --height 480 --frames 64 --window 11 --with-generic
  python3 tools/sweep_jbf.py --width 640 --height 480 --frames 64 --window 7 --spatial-sigma 70 --color-sigma 50 --depth-sigma 20
  python3 tools/sweep_jbf.py --width 640 --height 480 --frames 64 --window 5 --spatial-sigma 70 --color-sigma 50 --depth-sigma 20 --with-generic
} > "$OUT/sweep_k1_variants.log" 2> "$OUT/sweep.err"
echo "profile round $TAG done"
# 5. SPDepthSuperResolution::Process (row f2)
python3 tools/bench_spdsr.py > "$OUT/spdsr_fhd.json" 2> "$OUT/spdsr.err"
python3 tools/bench_spdsr.py --width 640 --height 480 > "$OUT/spdsr_vga.json" 2>> "$OUT/spdsr.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/spdsr_stats" -o spdsr -- python3 tools/bench_spdsr.py > /dev/null 2>> "$OUT/spdsr.err"
echo "spdsr done"
# 6. MarkovRandomField::Process (row f1)
python3 tools/bench_mrf.py > "$OUT/mrf_vga.json" 2> "$OUT/mrf.err"
python3 tools/bench_mrf.py --width 1920 --height 1080 --frames 8 > "$OUT/mrf_fhd.json" 2>> "$OUT/mrf.err"
echo "mrf done"
