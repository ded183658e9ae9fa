#!/bin/bash
# round 4: the record -- parity suite, smoke, the driver's bench command, the other configs
mkdir -p gpurun_out
export QK_CACHE_DIR=/tmp/qkc
timeout -k 10 800 python -m pytest tests -m gpu -x -q > gpurun_out/suite.log 2>&1; rc=$?; echo "rc $rc" >> gpurun_out/suite.log
tail -4 gpurun_out/suite.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1 || { tail -5 gpurun_out/smoke.txt; exit 1; }
grep smoke gpurun_out/smoke.txt
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_cfg4_final.json 2> gpurun_out/bench_cfg4_final.err || { tail -20 gpurun_out/bench_cfg4_final.err; exit 1; }
timeout -k 10 200 python bench.py --config cfg3 --steps 20 --warmup 5 > gpurun_out/bench_cfg3_final.json 2> gpurun_out/bench_cfg3_final.err || { tail -20 gpurun_out/bench_cfg3_final.err; exit 1; }
timeout -k 10 300 python bench.py --config cfg5 --steps 10 --warmup 3 > gpurun_out/bench_cfg5_final.json 2> gpurun_out/bench_cfg5_final.err || { tail -20 gpurun_out/bench_cfg5_final.err; exit 1; }
timeout -k 10 200 python bench.py --config cfg2 --steps 20 --warmup 5 > gpurun_out/bench_cfg2_final.json 2> gpurun_out/bench_cfg2_final.err || { tail -20 gpurun_out/bench_cfg2_final.err; exit 1; }
QK_DETERMINISTIC=1 timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-seconds 0 > gpurun_out/bench_cfg4_deterministic.json 2> gpurun_out/bench_cfg4_deterministic.err || { tail -20 gpurun_out/bench_cfg4_deterministic.err; exit 1; }
python - <<'PY'
import json
for c in ("cfg4_final","cfg3_final","cfg5_final","cfg2_final","cfg4_deterministic"):
    d=json.load(open(f"gpurun_out/bench_{c}.json"))
    r=d["roofline"]
    print(c, "ms/step %.2f value %.4g cold %.1f (x%.3f) plan %.1f derive %.1f cold_abi %s frac %.4f pipe %s reuse_GB %.1f traffic %s" % (d["ms_per_step"], d["value"], d["cold_step_ms"], d["cold_over_steady"], d["plan_ms"], d["derive_ms"], d.get("cold_c_abi_ms"), r["frac"], r["matrix_pipe_frac"], r["tile_reuse_gbytes"], r["traffic"]), [ (round(l["kernel_ms"],2), l["bound"], l.get("traffic_tb_per_s")) for l in r["launches"]])
    if "cpu_baseline" in d: print("   cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["parity_max_abs_err_vs_gpu"], "mps_build", {k:v for k,v in d["config"]["mps_build"].items() if "device" in k or "host_pool_s" in k})
PY
