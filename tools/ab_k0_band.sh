#!/bin/bash
# A/B of K0's XCD-band tile walk inside the batched chain (K0 runs between other kernels, its input is not cache-resident):
# rocprofv3 kernel stats per (mode, batch).   gpurun -- 'bash tools/ab_k0_band.sh'
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ab_k0_band
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in band noband; do
  if [ $mode = noband ]; then export KDE_K0_BAND_WALK=0; else export KDE_K0_BAND_WALK=1; fi
  for cfg in "8 1920 1080" "16 1280 720" "64 640 480" "4 1920 1080" "2 1920 1080" "1 1920 1080" "1 640 480"; do
    set -- $cfg
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/${mode}_$1x$2 -o s -- python3 $R/tools/bench_chain_batch.py --frames $1 --width $2 --height $3 --iters 10 > $O/${mode}_$1x$2.log 2>&1
    f=$(find $O/${mode}_$1x$2 -name '*kernel_stats.csv' | head -1)
    [ -n "$f" ] || { echo "no stats for $mode $cfg"; exit 1; }
    echo "$mode $cfg: $(grep presmooth "$f" | awk -F, '{print $2, $3, $4}')" | tee -a $O/summary.txt
  done
done
