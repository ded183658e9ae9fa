#!/usr/bin/env python3
"""Condense a profiles/run_rocprof.sh output directory into the files committed under profiles/<tag>/.

usage: python tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<tag> [kernel-substring]

Writes kernel_stats.csv (rocprofv3 --stats table), pmc_<pass>_sweep_rows.csv (the sweep kernel's counter
rows of each --pmc pass) and pmc_summary.json (bytes per launch with the gfx950 FETCH_SIZE x2 correction of
MI355X_MICROARCH.md, executed MFMA flops and MFMA utilisation).
"""
import csv
import json
import os
import shutil
import sys

PEAK_F64_MFMA = 256 * 4 * 32 * 2.4e9  # flop/s, dense fp64 MFMA, MI355X


def sweep_rows(path, needle):
    with open(path, newline="") as f:
        rows = list(csv.DictReader(f))
    return [r for r in rows if needle in r["Kernel_Name"]], (rows[0].keys() if rows else [])


def main():
    src, dst = sys.argv[1], sys.argv[2]
    needle = sys.argv[3] if len(sys.argv) > 3 else "qk_sweep"
    os.makedirs(dst, exist_ok=True)
    shutil.copyfile(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    raw, kernel_name = {}, None
    for tag in ("fetch", "write", "sq"):
        p = os.path.join(src, f"pmc_{tag}", "pmc_counter_collection.csv")
        if not os.path.exists(p):
            continue
        rows, keys = sweep_rows(p, needle)
        if not rows:
            continue
        kernel_name = rows[0]["Kernel_Name"]
        with open(os.path.join(dst, f"pmc_{tag}_sweep_rows.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(keys))
            w.writeheader()
            w.writerows(rows)
        launches = len({r["Dispatch_Id"] for r in rows})
        acc = {}
        for r in rows:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        raw[f"pmc_{tag}"] = {k: v / launches for k, v in acc.items()}
    derived = {}
    if "pmc_fetch" in raw and "pmc_write" in raw:
        rd = raw["pmc_fetch"]["FETCH_SIZE"] * 1024 * 2  # KiB -> B, x2: gfx950 reports half of wide coalesced reads
        wr = raw["pmc_write"]["WRITE_SIZE"] * 1024
        derived.update(hbm_read_bytes_per_launch=rd, hbm_write_bytes_per_launch=wr, traffic_bytes_per_launch=rd + wr,
                       note="FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the "
                            "bytes of wide coalesced 16 B/lane reads); separate --pmc passes (profiles/run_rocprof.sh)")
    if "pmc_sq" in raw and "SQ_INSTS_VALU_MFMA_MOPS_F64" in raw["pmc_sq"]:
        sq = raw["pmc_sq"]
        flops = sq["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512  # counter unit: 512 flop
        derived["mfma_flops_executed"] = flops
        with open(os.path.join(dst, "kernel_stats.csv"), newline="") as f:
            for r in csv.DictReader(f):
                if needle in r["Name"]:
                    derived["kernel_avg_ms_trace"] = float(r["AverageNs"]) / 1e6
                    derived["mfma_util"] = flops / (float(r["AverageNs"]) / 1e9) / PEAK_F64_MFMA
                    break
    out = {"kernel": kernel_name, "raw": raw, "derived": derived}
    if len(sys.argv) > 4:
        out["workload"] = sys.argv[4]
    with open(os.path.join(dst, "pmc_summary.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out["derived"], indent=1))


if __name__ == "__main__":
    main()
