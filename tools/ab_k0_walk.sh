#!/bin/bash
# K0's HBM-side traffic per tile walk (FETCH_SIZE / WRITE_SIZE of tools/bench_k0.py on 64 x 640x480, one rocprofv3 pass each;
# the bgr3_copy launch of the same run calibrates the counters) + the timing A/B.   gpurun -- 'bash tools/ab_k0_walk.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/ab_k0_walk
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/ab_k0_walk.py > $O/timing.json 2> $O/timing.err || { tail -5 $O/timing.err; exit 1; }
cat $O/timing.json
for mode in 0 1 2; do
  export KDE_K0_BAND_WALK=$mode
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $O/m$mode/$c -o p -- python3 $R/tools/bench_k0.py --wakeup-ms 0 > $O/m$mode.$c.log 2>&1 || { echo "pass $mode $c failed"; exit 1; }
  done
  python3 - $O/m$mode $mode <<'PY' | tee -a $O/traffic.txt
import csv, glob, sys
d, mode = sys.argv[1], sys.argv[2]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{d}/{c}/**/*counter_collection.csv", recursive=True)[0]
    for row in csv.DictReader(open(f)):
        k = "k0" if "presmooth" in row["Kernel_Name"] else "copy" if "bgr3_copy" in row["Kernel_Name"] else None
        if k and row["Counter_Name"] == c:
            tot.setdefault((k, c), []).append(float(row["Counter_Value"]))
med = lambda v: sorted(v)[len(v) // 2]
px = 64 * 640 * 480
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    cal = 3.0 * px / med(tot[("copy", c)])          # bytes per counter unit: the copy moves exactly 3 B/px each way
    out[c] = med(tot[("k0", c)]) * cal / (3.0 * px)
print(f"walk {mode}: K0 reads {out['FETCH_SIZE']:.2f} x algorithmic, writes {out['WRITE_SIZE']:.2f} x, total {(out['FETCH_SIZE'] + out['WRITE_SIZE']) / 2:.2f} x")
PY
done
