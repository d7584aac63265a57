#!/bin/bash
# A/B of K10's register bound on the GPU box (waves per SIMD the allocator leaves room for: 0 = none = 80 VGPRs / 6 workgroups per
# CU, 7 = 72 VGPRs, 8 = 64 VGPRs with 6 spills): rebuilds ers_kernels.o per variant, relinks, runs the single-frame chain, the
# batched chains and SPDSR.   gpurun -- 'bash tools/ab_k10_waves.sh'
set -e
cd "$(dirname "$0")/.."
C=kinectdepthmapenhancement_amd/csrc
mkdir -p gpurun_out
: > gpurun_out/ab_k10_waves.txt
for w in 0 8 7 0 8; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -fno-gpu-rdc -DKDE_K10_WAVES=$w -c $C/ers_kernels.hip -o $C/ers_kernels.o 2>/dev/null
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -fno-gpu-rdc -o kinectdepthmapenhancement_amd/libkde_hip.so $C/kde_api.o $C/jbf_kernels.o $C/jbf_fast.o $C/stream_kernels.o $C/dasp_kernels.o $C/ers_kernels.o $C/spdsr_kernels.o
  a=$(python3 tools/bench_chain.py 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['rgbf_process_ms'],4), round(j['chain_ms'],4))")
  b=$(python3 tools/bench_chain_batch.py --frames 8 --width 1920 --height 1080 --iters 10 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['batched_ms_per_frame'],4))")
  c=$(python3 tools/bench_chain_batch.py --frames 64 --width 640 --height 480 --iters 10 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['batched_ms_per_frame'],5))")
  d=$(python3 tools/bench_chain.py --width 640 --height 480 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(j['rgbf_process_ms'],4), round(j['chain_ms'],4))")
  echo "K10 waves bound $w: 1080p single frame rgbf / chain ms $a; 8 x 1080p batched ms/frame $b; 64 x VGA batched ms/frame $c; VGA single rgbf / chain $d" | tee -a gpurun_out/ab_k10_waves.txt
done
