#!/usr/bin/env python3
"""Per-basic-block instruction counts of the matrix loops of a kernel (blocks with >= 8 matrix instructions) in an ISA listing
(hipcc --offload-arch=gfx950 -O3 --offload-device-only -S).   usage: isa_blocks.py file.s [kernel-name substring ...]"""
import re
import sys


def analyze(path, key):
    txt = open(path).read()
    m = re.search(r"^(_Z\w*" + key + r"\w*):.*\n", txt, re.M)
    if not m:
        return None
    start = m.end()
    end = txt.index("s_endpgm", start)
    blocks, cur, name = [], [], "entry"
    for line in txt[start:end].split("\n"):
        if re.match(r"^\.LBB\d+_\d+:", line):
            blocks.append((name, cur))
            name, cur = line.split(":")[0], []
        else:
            ins = line.strip()
            if ins and not ins.startswith(";") and not ins.startswith("."):
                cur.append(ins)
    blocks.append((name, cur))
    out = []
    for n, b in blocks:
        c = lambda p: sum(1 for i in b if i.startswith(p))  # noqa: E731
        if c("v_mfma") >= 8:
            out.append((n, len(b), c("v_mfma"), c("v_") - c("v_mfma"), c("v_add_f64"), c("v_mov") + c("v_accvgpr"), c("ds_add"), c("ds_read") + c("ds_load"), c("global_load"), c("scratch"), c("s_nop"), c("s_")))
    return out


if __name__ == "__main__":
    keys = sys.argv[2:] or ["qk_sweep_fused_dual_kernelILi12ELi8192ELi3ELb0E", "qk_sweep_fused_kernelILi8ELi1ELi4608ELi4ELb0E"]
    for k in keys:
        print(k)
        for r in analyze(sys.argv[1], k) or []:
            print("  %-10s n=%4d mfma=%3d valu=%3d (add_f64 %3d, mov %3d) ds_add=%3d ds_read=%2d gload=%2d scratch=%2d s_nop=%d salu=%d" % r)
