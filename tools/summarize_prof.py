#!/usr/bin/env python3
"""Condense a profiles/run_rocprof.sh output directory into the files committed under profiles/<tag>/.

usage: python tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<tag> [kernel-substring] [workload text]

Writes kernel_stats.csv (rocprofv3 --stats table), pmc_<pass>_sweep_rows.csv (the sweep kernels' counter rows of each --pmc pass),
phase_ranges.csv (the roctx ranges of the marker-trace run, when present) and pmc_summary.json: per sweep kernel the bytes per
launch (FETCH_SIZE x the gfx950 factor calibrated by tools/fetch_calib.sh, default 2 = MI355X_MICROARCH.md), executed MFMA flops
and MFMA utilisation -- and `source_sha`, the digest of the sweep kernels' sources (bench.py quotes the traffic only while it matches).
"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PEAK_F64_MFMA = 256 * 4 * 32 * 2.4e9  # flop/s, dense fp64 MFMA, MI355X
FETCH_FACTOR = float(os.environ.get("QK_FETCH_FACTOR", "2.0"))  # bytes really fetched per byte FETCH_SIZE reports (profiles/r03/fetch_calibration.txt)


def sweep_rows(path, needle):
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))
    return [r for r in rows if needle in r["Kernel_Name"]], (rows[0].keys() if rows else [])


def main():
    src, dst = sys.argv[1], sys.argv[2]
    needle = sys.argv[3] if len(sys.argv) > 3 else "qk_sweep"
    os.makedirs(dst, exist_ok=True)
    shutil.copyfile(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    avg_ms = {}
    with open(os.path.join(dst, "kernel_stats.csv"), newline="") as f:
        for r in csv.DictReader(f):
            if needle in r["Name"]:
                avg_ms[r["Name"]] = float(r["AverageNs"]) / 1e6
    kernels = {}
    for tag in ("fetch", "write", "sq", "lds", "tcc"):
        p = os.path.join(src, f"pmc_{tag}", "pmc_counter_collection.csv")
        if not os.path.exists(p):
            continue
        rows, keys = sweep_rows(p, needle)
        if not rows:
            continue
        with open(os.path.join(dst, f"pmc_{tag}_sweep_rows.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(keys))
            w.writeheader()
            w.writerows(rows)
        for name in sorted({r["Kernel_Name"] for r in rows}):
            mine = [r for r in rows if r["Kernel_Name"] == name]
            launches = len({r["Dispatch_Id"] for r in mine})
            acc = {}
            for r in mine:
                acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            kernels.setdefault(name, {"raw": {}})["raw"][f"pmc_{tag}"] = {k: v / launches for k, v in acc.items()}
    for name, ent in kernels.items():
        raw = ent["raw"]
        if "pmc_fetch" in raw and "pmc_write" in raw:
            rd = raw["pmc_fetch"]["FETCH_SIZE"] * 1024 * FETCH_FACTOR  # KiB -> B
            wr = raw["pmc_write"]["WRITE_SIZE"] * 1024
            ent.update(hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, traffic_bytes_per_launch=rd + wr, fetch_factor=FETCH_FACTOR)
        ms = next((v for k, v in avg_ms.items() if k in name or name in k), None)
        if ms:
            ent["kernel_avg_ms_trace"] = ms
        if "pmc_sq" in raw and "SQ_INSTS_VALU_MFMA_MOPS_F64" in raw["pmc_sq"]:
            flops = raw["pmc_sq"]["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512  # counter unit: 512 flop
            ent["mfma_flops_executed"] = flops
            if ms:
                ent["mfma_util"] = flops / (ms / 1e3) / PEAK_F64_MFMA
            sq = raw["pmc_sq"]
            if sq.get("SQ_BUSY_CU_CYCLES"):
                ent["mfma_busy_share_of_cu_busy"] = sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / sq["SQ_BUSY_CU_CYCLES"] / 4.0  # four SIMDs per CU
        if "pmc_lds" in raw and raw["pmc_lds"].get("SQ_BUSY_CYCLES"):
            ld = raw["pmc_lds"]
            # SQ_LDS_IDX_ACTIVE = cycles the LDS array works for indexed ops, SQ_LDS_BANK_CONFLICT = the extra cycles of conflicts (MI355X_MICROARCH.md, LDS);
            # both summed over the CUs like SQ_BUSY_CYCLES; SQ_WAIT_INST_LDS in quad-cycles of waves, against SQ_WAVE_CYCLES
            # (SQ_BUSY_CYCLES comes back summed over the 32 shader engines, the LDS counters over the 256 CUs: 8 CUs per engine)
            ent["lds_active_share_of_busy"] = ld.get("SQ_LDS_IDX_ACTIVE", 0.0) / ld["SQ_BUSY_CYCLES"] / 8.0
            ent["lds_bank_conflict_share_of_lds_active"] = ld.get("SQ_LDS_BANK_CONFLICT", 0.0) / ld["SQ_LDS_IDX_ACTIVE"] if ld.get("SQ_LDS_IDX_ACTIVE") else None
            ent["wave_share_waiting_to_issue_lds"] = ld.get("SQ_WAIT_INST_LDS", 0.0) / ld["SQ_WAVE_CYCLES"] if ld.get("SQ_WAVE_CYCLES") else None
            ent["lds_instructions_per_launch"] = ld.get("SQ_INSTS_LDS")
        if "pmc_tcc" in raw and (raw["pmc_tcc"].get("TCC_HIT_sum", 0.0) + raw["pmc_tcc"].get("TCC_MISS_sum", 0.0)) > 0:
            tc = raw["pmc_tcc"]
            ent["l2_hit_rate"] = tc["TCC_HIT_sum"] / (tc["TCC_HIT_sum"] + tc["TCC_MISS_sum"])
    # roctx ranges of the marker-trace run: one line per range name
    mk = glob.glob(os.path.join(src, "marker", "**", "*marker_api_trace.csv"), recursive=True)
    if mk:
        agg = {}
        with open(mk[0], newline="") as f:
            for r in csv.DictReader(f):
                name = r.get("Function") or r.get("Name") or ""
                try:
                    dur = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
                except (KeyError, ValueError):
                    continue
                a = agg.setdefault(name, [0, 0.0])
                a[0] += 1
                a[1] += dur
        with open(os.path.join(dst, "phase_ranges.csv"), "w") as f:
            f.write("range,calls,total_ms,mean_ms\n")
            for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{name},{n},{tot:.3f},{tot / n:.3f}\n")
    import bench

    out = {"kernels": kernels, "source_sha": bench.sweep_source_sha(), "sources": list(bench.SWEEP_SOURCES),
           "note": "FETCH_SIZE / WRITE_SIZE are in KiB, separate --pmc passes (profiles/run_rocprof.sh); FETCH_SIZE x fetch_factor = bytes (tools/fetch_calib.sh)"}
    if len(sys.argv) > 4:
        out["workload"] = sys.argv[4]
    with open(os.path.join(dst, "pmc_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "raw"} for k, v in kernels.items()}, indent=1))


if __name__ == "__main__":
    main()
