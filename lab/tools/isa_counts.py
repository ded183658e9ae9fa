#!/usr/bin/env python3
"""Static instruction counts of the site-fused sweep kernels in an ISA listing (hipcc --offload-arch=gfx950 -O3 --offload-device-only -S
-Rpass-analysis=kernel-resource-usage of csrc/qkgram.hip; the remarks go to a second file).   usage: isa_counts.py file.s remarks.txt"""
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    rem = open(sys.argv[2]).read() if len(sys.argv) > 2 else ""
    print("%-66s %5s %6s %5s %6s %6s %6s %6s %7s   scratch B/lane, spilled SGPRs, VGPRs, spilled VGPRs" % ("kernel", "mfma", "gload", "flat", "scr_ld", "scr_st", "lanemv", "ds_add", "barrier"))
    for m in re.finditer(r"^(_Z\d+qk_sweep_fused\w*):", txt, re.M):
        sym = m.group(1)
        start = m.end()
        end = txt.index("s_endpgm", start)
        body = [l.strip() for l in txt[start:end].split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = lambda p: sum(1 for i in body if i.startswith(p))  # noqa: E731
        r = rem[rem.index(sym):][:1500] if sym in rem else ""
        g = lambda k: (re.search(k + r": (\d+)", r) or [None, "?"])[1]  # noqa: E731
        print("%-66s %5d %6d %5d %6d %6d %6d %6d %7d   %s %s %s %s" % (sym[3:60], c("v_mfma"), c("global_load"), c("flat_load"), c("scratch_load"), c("scratch_store"),
              c("v_readlane") + c("v_writelane"), c("ds_add"), c("s_barrier"), g(r"ScratchSize \[bytes/lane\]"), g("SGPRs Spill"), g("VGPRs"), g("VGPRs Spill")))


if __name__ == "__main__":
    main()
