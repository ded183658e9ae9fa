import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import qml_cutensornet_amd as Q
from qml_cutensornet_amd.data import synthetic_features
from qml_cutensornet_amd.dist import SingleComm
from qml_cutensornet_amd.gpu_backend.kernel_state_ansatz import build_kernel_matrix
n, reps, d, N = 40, 4, 2, 200
X = synthetic_features(N, n, 5)
ans = Q.KernelStateAnsatz(n, reps, 1.0, Q.entanglement_graph(n, d))
out = {}
for which in ("auto", "host"):
    os.environ["QK_BUILDER"] = which
    t0 = time.perf_counter()
    K = build_kernel_matrix(SingleComm(), ans, X=X, info_file=f"/tmp/e2e_{which}", truncation_error=1e-16)
    out[which] = (K, time.perf_counter() - t0)
    Kt = build_kernel_matrix(SingleComm(), ans, X=X, Y=X[:30] * 0.97 + 0.01, truncation_error=1e-16)
    out[which + "_test"] = Kt
print("train Gram auto %.2f s, host %.2f s; max |dK| = %.2e; test Gram max |dK| = %.2e" % (out["auto"][1], out["host"][1], np.abs(out["auto"][0] - out["host"][0]).max(), np.abs(out["auto_test"] - out["host_test"]).max()))
import json
print({k: v for k, v in json.load(open("/tmp/e2e_auto.json")).items() if k in ("r0_circ_sim", "kernel_mat_time", "avg_fidelity", "ave max chi", "avg_product")})
print({k: v for k, v in json.load(open("/tmp/e2e_host.json")).items() if k in ("r0_circ_sim", "kernel_mat_time", "avg_fidelity", "ave max chi", "avg_product")})
