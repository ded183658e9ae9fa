"""Synthetic stand-in for the reference's feature pipeline (/root/reference/main.py:130-143):
QuantileTransformer(normal) -> StandardScaler -> MinMaxScaler((0, 2)).  Without the Elliptic data
set the first step is replaced by drawing standard-normal columns directly."""
from __future__ import annotations

import numpy as np


def synthetic_features(n_points: int, n_features: int, seed: int = 5) -> np.ndarray:
    gen = np.random.default_rng(seed)
    x = gen.standard_normal(size=(n_points, n_features))
    x -= x.mean(axis=0, keepdims=True)
    x /= x.std(axis=0, keepdims=True)
    lo = x.min(axis=0, keepdims=True)
    span = x.max(axis=0, keepdims=True) - lo
    return (x - lo) * (2.0 / span)
