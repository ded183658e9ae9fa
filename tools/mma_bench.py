import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qml_cutensornet_amd import engine
ctx = engine.Context(0)
names = {0: "4 waves, pipelined frags", 1: "4 waves, plain", 2: "8 waves, pipelined frags", 3: "8 waves, plain"}
for which in (0, 1, 2, 3):
    for w in (1, 2):
        tf = ctx.debug_mma_bench(which, w, 4000)
        print(f"{names[which]:28s} wgs/cu {w}: {tf:6.2f} TFLOP/s  ({tf / 78.6432:.3f} of 78.6)")
