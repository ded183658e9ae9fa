#!/bin/bash
# round 4, GPU call 28: the record on the final build -- per-site cost on uniform small chains, the 4-rank rehearsal of the driver's multi-GPU command over gloo on one GPU
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
timeout -k 10 400 python tools/site_overhead.py > gpurun_out/site_overhead.txt 2>&1 || { tail -5 gpurun_out/site_overhead.txt; exit 1; }
tail -8 gpurun_out/site_overhead.txt | cut -c1-170
python bench.py --cpu-seconds 0 --steps 1 --warmup 0 > /dev/null 2> gpurun_out/prime.err || { tail -3 gpurun_out/prime.err; exit 1; }
QK_FORCE_DEVICE=0 QK_DIST_BACKEND=gloo QK_BENCH_DEVICE_BUILD=0 timeout -k 10 700 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 4 --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/bench_4rank_rehearsal.json 2> gpurun_out/bench_4rank_rehearsal.err || { tail -8 gpurun_out/bench_4rank_rehearsal.err; exit 2; }
tail -1 gpurun_out/bench_4rank_rehearsal.json | cut -c1-600
