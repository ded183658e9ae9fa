#!/usr/bin/env python3
"""Section shares of the instrumented sweep kernel (QK_VARIANT=9) on the cfg4 workload (or a uniform-chi set).
usage: [QK_VARIANT=19|9] python lab/tools/profile_sections.py [chi]   (19 = instrumented shipped kernel, 9 = 4-wave flat)"""
import os, pickle, sys, glob
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("QK_VARIANT", "19")
import qml_cutensornet_amd as Q
from qml_cutensornet_amd import engine

engine.use_lab_library()  # the experimental kernels live in libqklab.so only

names = ["fetch issue", "MFMA block", "epilogue stores", "stash (+vmcnt wait)", "barrier", "phase prologue", "phase-end barrier", "wave lifetime"]
ctx = engine.Context(0)
if len(sys.argv) > 1:
    chi = int(sys.argv[1]); n = 60
    rng = np.random.default_rng(0)
    m0 = Q.random_mps(n, [min(2 ** min(k, n - k), chi) for k in range(n + 1)], rng)
    states = [m0] * 91
    label = f"uniform chi={chi}"
else:
    cdir = sorted(glob.glob(os.path.join(os.environ.get("QK_CACHE_DIR", "/tmp/qkc"), "*")))[-1]
    tensors = []
    for f in sorted(glob.glob(os.path.join(cdir, "chunk_*.pkl")))[:8]:
        tensors += pickle.load(open(f, "rb"))[0]
    states = [Q.MPS(t) for t in tensors]
    label = f"cfg4 real states ({len(states)})"
xs = ctx.upload(states)
plan = engine.Plan(xs.dims)
ctx.gram_values_host(xs, None, plan)
st = ctx.stats(); pr = ctx.debug_profile()
life = pr[7]
print(f"{label}: {st['pairs']} pairs, kernel {st['kernel_ms']:.1f} ms (instrumented), padded TF/s {st['padded_flops']/st['kernel_ms']/1e9:.1f}")
acc = 0
for nm, v in zip(names[:7], pr[:7]):
    print(f"  {nm:22s} {100.0 * v / life:6.2f} %")
    acc += v
print(f"  {'other (pair setup,..)':22s} {100.0 * (life - acc) / life:6.2f} %")
