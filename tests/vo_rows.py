"""Synthetic per-frame motion rows [R (9), t (3), step length] shared by tests/test_runner_cpu.py and its gloo workers."""
import numpy as np


def vo_rows(n, seed=4):
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n):
        ax = rng.normal(0, 0.05, 3)
        th = np.linalg.norm(ax)
        k = ax / th
        Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
        R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
        t = rng.normal(size=3)
        t /= np.linalg.norm(t)
        step = 0.0 if i % 4 == 0 else float(rng.uniform(0.01, 0.5))      # every fourth frame stands still (< 1 mm: no update)
        rows.append(np.r_[R.ravel(), t, step].astype(np.float32).tolist())
    return rows
