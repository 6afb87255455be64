#!/bin/bash
# bench.py --gpus 4 / 8 rehearsed on the one GPU (ranks share it, gloo transport), both launch forms.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export RK_BENCH_SINGLE_DEVICE=1 RK_BENCH_BACKEND=gloo
for n in 4 8; do
  timeout 600 python3 bench.py --gpus $n --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('self-launched n_gpus', d['n_gpus'], d['value'], d['ms_per_step'], d['scaling'], d['config']['nparts_per_gpu'], d['host']['replicate_via'][:40])"
done
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 5 --warmup 2 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('torchrun n_gpus', d['n_gpus'], d['value'], d['ms_per_step'], d['scaling'], d['config']['nparts_per_gpu'])"
