import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qml_cutensornet_amd import engine

engine.use_lab_library()  # the experimental kernels live in libqklab.so only
ctx = engine.Context(0)
names = ["bare MFMA block", "+ barrier per step", "+ LDS stash", "+ global fetch (1 step ahead)", "+ global fetch (2 steps ahead)", "  same, 1-ahead, from HBM stream", "  same, 2-ahead, from HBM stream"]
for nw in (4, 8):
    for fi, nm in enumerate(names):
        row = []
        for w in (1, 2):
            tf = ctx.debug_mma_bench(8 * (nw == 8) + fi, w, 4000)
            row.append(f"{tf:6.2f} TF ({tf / 78.6432:.3f})")
        print(f"{nw} waves  {nm:32s} wgs/cu 1: {row[0]}   wgs/cu 2: {row[1]}")
    for epi in (6, 3, 2, 1):  # epilogue every `epi` pairs of steps = K of 192, 96, 64, 32
        row = []
        for w in (1, 2):
            tf = ctx.debug_mma_bench(16 * epi + 8 * (nw == 8) + 7, w, 4800)
            row.append(f"{tf:6.2f} TF ({tf / 78.6432:.3f})")
        print(f"{nw} waves    + epilogue every {2 * epi:2d} steps (HBM stream, 2-ahead)  wgs/cu 1: {row[0]}   wgs/cu 2: {row[1]}")
