#!/bin/bash
# A/B of K0's compile-time shape on the GPU box: pixels per thread (KDE_K0_PX) x register cap in waves per SIMD
# (KDE_K0_RP_WAVES).  Rebuilds jbf_kernels.o per variant, relinks, runs tools/bench_k0.py.   gpurun -- 'bash tools/ab_k0_build.sh'
set -e
cd "$(dirname "$0")/.."
C=kinectdepthmapenhancement_amd/csrc
mkdir -p gpurun_out
for v in "2 7" "2 8" "2 6" "4 5" "4 4" "4 6" "2 7"; do
  set -- $v
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fno-gpu-rdc -DKDE_K0_PX=$1 -DKDE_K0_RP_WAVES=$2 -c $C/jbf_kernels.hip -o $C/jbf_kernels.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc -o kinectdepthmapenhancement_amd/libkde_hip.so $C/kde_api.o $C/jbf_kernels.o $C/jbf_fast.o $C/stream_kernels.o $C/dasp_kernels.o $C/ers_kernels.o $C/spdsr_kernels.o
  a=$(python3 tools/bench_k0.py | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms'],4))")
  b=$(python3 tools/bench_k0.py --width 1920 --height 1080 --frames 1 | python3 -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms']*1e3,2))")
  echo "px $1 waves $2: 64x640x480 $a ms, 1x1920x1080 $b us" | tee -a gpurun_out/ab_k0_build.txt
done
