#!/usr/bin/env python3
"""Section shares of the instrumented site-fused sweep (a -DQKF_PROF experiment build: QK_LIB=...) on cfg4's real
states (the bench cache) or a uniform-bond set.  usage: QK_LIB=gpurun_exp/lib_prof.so python lab/tools/fused_sections.py [chi]"""
import glob, os, pickle, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("QK_FUSED", "2")
import qml_cutensornet_amd as Q
from qml_cutensornet_amd import engine
engine.use_lab_library()  # $QK_LIB (an instrumented experiment build) or libqklab.so
names = ["pair set-up", "X LDS<->global", "phase 1", "wait P1 + zero", "phase 2", "wait P2", "touch / strip zero", "wave lifetime"]
ctx = engine.Context(0)
if len(sys.argv) > 1:
    chi = int(sys.argv[1]); n = 60
    rng = np.random.default_rng(0)
    m0 = Q.random_mps(n, [min(2 ** min(k, n - k), chi) for k in range(n + 1)], rng)
    states = [m0] * 181
    label = f"uniform chi={chi}"
else:
    cdir = sorted(glob.glob(os.path.join(os.environ.get("QK_CACHE_DIR", "/tmp/qkc"), "*")))[-1]
    tensors = []
    for f in sorted(glob.glob(os.path.join(cdir, "chunk_*.pkl")))[: int(os.environ.get("CHUNKS", "20"))]:
        tensors += pickle.load(open(f, "rb"))[0]
    states = [Q.MPS(t) for t in tensors]
    label = f"cfg4 real states ({len(states)})"
xs = ctx.upload(states)
plan = engine.Plan(xs.dims)
ctx.gram_values_host(xs, None, plan)
pr0 = ctx.debug_profile()
ctx.gram_values_host(xs, None, plan)
st = ctx.stats(); pr = [b - a for a, b in zip(pr0, ctx.debug_profile())]
life = pr[7]
print(f"{label}: {st['pairs']} pairs, kernel {st['kernel_ms']:.1f} ms (instrumented), padded TF/s {st['padded_flops']/st['kernel_ms']/1e9:.1f}")
acc = 0
for nm, v in zip(names[:7], pr[:7]):
    print(f"  {nm:22s} {100.0 * v / life:6.2f} %")
    acc += v
print(f"  {'other':22s} {100.0 * (life - acc) / life:6.2f} %")
