#!/bin/bash
# Known-bytes calibration of FETCH_SIZE for the access patterns of the sweep kernels (run on the GPU box):
#   bash tools/fetch_calib.sh   -> gpurun_out/fetch_calib/summary.txt  (copy to profiles/r03/fetch_calibration.txt)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/fetch_calib
mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 "$R/tools/fetch_calib.hip" -o /tmp/fetch_calib || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc" -o pmc -- /tmp/fetch_calib > "$OUT/run.log" 2>&1 || exit 2
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/pmc/**/pmc_counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "calib_" in r["Kernel_Name"]]
known = 4 * 2**30
lds = [64, 128, 256, 1024]
seen = {}
lines = ["FETCH_SIZE calibration on gfx950: 4 GiB read exactly once per launch (16 x the Infinity Cache); FETCH_SIZE is in KiB", "kernel / pattern                               FETCH_SIZE bytes   known / FETCH_SIZE"]
for r in rows:
    name = r["Kernel_Name"].split("(")[0]
    k = seen.get(name, 0); seen[name] = k + 1
    tag = name if "wide" in name else f"{name} ld={lds[k % 4]} complex per row"
    fb = float(r["Counter_Value"]) * 1024
    lines.append(f"{tag:46s} {fb:16.0f}   {known / fb:6.3f}")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
