#!/bin/bash
# round 4, GPU call 4: first-store accumulation against default / ordered; share times with the one-launch rule; 4-rank rehearsal of the bench (cold step included)
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
O=gpurun_out/exp3.txt
: > $O
run() { echo "== $*" >> $O; timeout -k 10 500 "$@" >> $O 2>&1 || { echo "FAILED rc $?" >> $O; tail -5 $O; exit 1; }; }
run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8 QK_DETERMINISTIC=1
QK_AB_LIB=lab/libqkgram_first.so run python tools/ab_plan.py cfg4 3 QK_PLAN_TILE=8 QK_DETERMINISTIC=1 QK_DETERMINISTIC=1
run python tools/ab_plan.py cfg3 5 QK_PLAN_TILE=8 QK_DETERMINISTIC=1
QK_AB_LIB=lab/libqkgram_first.so run python tools/ab_plan.py cfg3 5 QK_PLAN_TILE=8 QK_DETERMINISTIC=1 QK_DETERMINISTIC=1
run python tools/share_times.py cfg4 3 1,2,4,8
grep -E "^==|kernel |world|rank " $O | cut -c1-250
QK_FORCE_DEVICE=0 QK_DIST_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 4 --steps 3 --warmup 1 --cpu-seconds 0 > gpurun_out/bench_4rank_rehearsal.json 2> gpurun_out/bench_4rank_rehearsal.err || { tail -5 gpurun_out/bench_4rank_rehearsal.err; exit 2; }
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/bench_4rank_rehearsal.json") if l.startswith("{")][-1])
c=d["config"]
print("4-rank rehearsal (one GPU, gloo): ms/step %.1f cold %.1f" % (d["ms_per_step"], d["cold_step_ms"]))
for k in ("rank_kernel_ms","rank_padded_tflop","rank_tail_frac","rank_cold_ms","rank_plan_ms","rank_job_setup_ms","rank_derive_ms","rank_cold_sweep_ms","rank_allgather_ms","k_identical_on_all_ranks"):
    print("  ",k,c[k])
PY
