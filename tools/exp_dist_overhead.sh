set -x
mkdir -p gpurun_out/r04l
C="--steps 20 --warmup 5 --cpu-seconds 0 --no-verify --no-extra"
show() { python -c "
import json,sys
j=json.loads([l for l in open('$1') if l.startswith('{')][-1])
b=j['boundary_vs_split']; print('$1', round(j['value']), j['rccl']['backend_used'], 'step', b['step_gpu_ms_all'][:6], b['step_gpu_ms_all'][-3:], 'split/step', round(b['split_leg_ms_per_step'],4), 'k1', round(j['roofline']['avg_launch_ms'],4))
"; }
python bench.py --gpus 1 $C > gpurun_out/r04l/plain.json 2> gpurun_out/r04l/plain.err; show gpurun_out/r04l/plain.json
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29701 python bench.py --gpus 1 $C > gpurun_out/r04l/env_nccl.json 2> gpurun_out/r04l/env_nccl.err; show gpurun_out/r04l/env_nccl.json
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29702 python bench.py --gpus 1 --backend gloo $C > gpurun_out/r04l/env_gloo.json 2> gpurun_out/r04l/env_gloo.err; show gpurun_out/r04l/env_gloo.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29703 bench.py --gpus 1 $C > gpurun_out/r04l/torchrun_nccl.json 2> gpurun_out/r04l/torchrun_nccl.err; show gpurun_out/r04l/torchrun_nccl.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29704 bench.py --gpus 1 --backend gloo $C > gpurun_out/r04l/torchrun_gloo.json 2> gpurun_out/r04l/torchrun_gloo.err; show gpurun_out/r04l/torchrun_gloo.json
OMP_NUM_THREADS=1 python bench.py --gpus 1 $C > gpurun_out/r04l/plain_omp1.json 2> gpurun_out/r04l/plain_omp1.err; show gpurun_out/r04l/plain_omp1.json
