#!/bin/bash
# two ranks of bench.py on ONE GPU over gloo (what the driver runs over RCCL on 2 GPUs), as a plumbing rehearsal
export QK_CACHE_DIR=${QK_CACHE_DIR:-/tmp/qkc}
mkdir -p gpurun_out
python bench.py --cpu-seconds 0 --steps 1 --warmup 0 > /dev/null 2> gpurun_out/prime.err || { tail -3 gpurun_out/prime.err; exit 1; }
QK_FORCE_DEVICE=0 QK_DIST_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/bench_2rank_rehearsal.json 2> gpurun_out/bench_2rank_rehearsal.err || { tail -5 gpurun_out/bench_2rank_rehearsal.err; exit 2; }
tail -1 gpurun_out/bench_2rank_rehearsal.json | cut -c1-400
