"""Benchmark-harness driver with the command line of the reference's ``main_no_test.py``
(/root/reference/main_no_test.py:79-175; same nine positional arguments as ``main.py``, ref
main.py:76-93) on top of the MI355X drop-in ``build_kernel_matrix``.

    python -m qml_cutensornet_amd.driver <backend> <num_features> <layers> <gamma> <distance>
                                         <n_illicit> <n_licit> <data_seed> <data_file>

Outputs, named exactly as the reference names them (ref main.py:161-175) so that its
``runs/*/to_csv.py`` post-processing applies unchanged:
    kernels/train_Nf{n}_r{r}_g{gamma}_p0.0_nn{d}_mslinear_Ntr{n_illicit}_s{seed}_{file stem}.npy
    <same stem>.json   (profiling keys of gpu_backend/kernel_state_ansatz.py:160-444)

If ``datasets/<data_file>`` exists it is read, sampled and scaled as the reference does
(ref main.py:126-143: QuantileTransformer -> StandardScaler -> MinMaxScaler((0, 2)), first
``num_features`` columns).  Without the (non-redistributable) Elliptic CSV the same number of
training points, 0.8 * (n_illicit + n_licit) (ref main.py:62 test_size=0.2), is drawn from the
synthetic generator of ``data.py``.  Only the GPU backend exists here; "CPU" is refused.
"""
from __future__ import annotations

import os
import pathlib
import sys
import time

import numpy as np

USAGE = (
    "\nCall script as 'python -m qml_cutensornet_amd.driver <backend> <num_features> <layers> <gamma> <distance> "
    "<n_illicit> <n_licit> <data_seed> <data_file>'.\nThe value of <backend> must be GPU."
)
TRUNCATION_ERROR = 1e-16  # hard-coded in the reference (main.py:73)


def run_name(kind, num_features, reps, gamma, nn, n_illicit, seed, data_file):
    """File stem of the reference (main.py:161-162)."""
    return f"{kind}_Nf{num_features}_r{reps}_g{gamma}_p0.0_nn{nn}_mslinear_Ntr{n_illicit}_s{seed}_{data_file.split('.')[0]}"


def load_training_features(data_file, n_illicit, n_licit, seed, num_features):
    path = os.path.join("datasets", data_file)
    if os.path.exists(path):
        import pandas as pd
        from sklearn.model_selection import train_test_split
        from sklearn.preprocessing import MinMaxScaler, QuantileTransformer, StandardScaler

        df = pd.read_csv(path)
        picked = pd.concat([
            df[df["Class"] == 0].sample(n_illicit, random_state=seed * 20 + 2),
            df[df["Class"] == 1].sample(n_licit, random_state=seed * 46 + 9),
        ], axis=0)
        train_df, _ = train_test_split(picked, stratify=picked["Class"], test_size=0.2, random_state=seed * 26 + 19)
        train_df = train_df.drop(columns=["Class"])
        x = QuantileTransformer(output_distribution="normal").fit_transform(np.array(train_df))
        x = StandardScaler().fit_transform(x)
        x = MinMaxScaler((0, 2)).fit_transform(x)
        return x[:, :num_features], "dataset"
    from .data import synthetic_features

    n_train = int(round(0.8 * (n_illicit + n_licit)))
    return synthetic_features(n_train, num_features, seed), "synthetic"


def make_comm():
    """torch.distributed group if the launcher set one up (torchrun), else a single process."""
    from .dist import SingleComm, TorchComm

    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        return TorchComm()
    return SingleComm()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) < 9:
        raise ValueError(USAGE)
    backend = str(argv[0])
    num_features, reps = int(argv[1]), int(argv[2])
    gamma = float(argv[3])
    nn = int(argv[4])
    n_illicit, n_licit, seed = int(argv[5]), int(argv[6]), int(argv[7])
    data_file = str(argv[8])
    if backend != "GPU":
        raise ValueError(USAGE)

    from .ansatz import entanglement_graph
    from .gpu_backend.kernel_state_ansatz import KernelStateAnsatz, build_kernel_matrix

    comm = make_comm()
    root = comm.Get_rank() == 0
    emap = entanglement_graph(num_features, nn)
    if root:
        print(f"\nUsing the following parameters:\n\n\tn_procs: {comm.Get_size()}\n\tbackend: {backend}\n\n\tnum_features: {num_features}"
              f"\n\treps: {reps}\n\tgamma: {gamma}\n\tinteraction distance: {nn}\n\n\tn_illicit: {n_illicit}\n\tn_licit: {n_licit}"
              f"\n\n\tdata_seed: {seed}\n\tdata_file: {data_file}\n")
        sys.stdout.flush()
    x_train, source = load_training_features(data_file, n_illicit, n_licit, seed, num_features)
    if root:
        print(f"\ttraining points: {len(x_train)} ({source} features)")
        pathlib.Path("kernels").mkdir(exist_ok=True)
        pathlib.Path("data").mkdir(exist_ok=True)
    ansatz = KernelStateAnsatz(num_qubits=num_features, reps=reps, gamma=gamma, entanglement_map=emap, hadamard_init=True)
    train_info = run_name("train", num_features, reps, gamma, nn, n_illicit, seed, data_file)
    t0 = time.perf_counter()
    kernel_train = build_kernel_matrix(comm, ansatz, X=x_train, info_file=train_info, truncation_error=TRUNCATION_ERROR)
    if root:
        print(f"Built kernel matrix on training set. Time: {round(time.perf_counter() - t0, 2)} seconds\n")
        np.save(f"kernels/{train_info}.npy", kernel_train)
    return kernel_train


if __name__ == "__main__":
    main()
