"""Synthetic Boolean inputs: the planted-factor recipe of PyBMF's SyntheticMatrixGenerator.

Mirrors ``PyBMF/generators/SyntheticMatrixGenerator.py:28-70`` + ``BaseGenerator.py:158-215`` (same class name,
constructor, ``generate(seed)``, ``add_noise(noise, seed)``, attributes ``X, U, V``) for matrices that fit the host,
bit-for-bit (pinned by tests/golden/g6_generator.json), and adds ``PlantedBooleanOnDevice`` for the benchmark
size, where the 2e9 cells are produced chunk by chunk on the GPU and only ever exist as bits.
"""
from __future__ import annotations

import time

import numpy as np


def _planted(rows: int, k: int, density: float, rng) -> np.ndarray:
    out = np.zeros((rows, k), dtype=np.uint8)
    block = int(np.ceil(rows / 100))
    tail = rows - k * block
    for c in range(k):
        out[c * block:(c + 1) * block, c] = 1
        out[k * block:rows, c] = rng.binomial(size=tail, n=1, p=density)
    return out


class SyntheticMatrixGenerator:
    """Host generator, same draws in the same order as the reference (one RandomState shared by all steps)."""

    def __init__(self, m, n, k, density=(0.2, 0.2)):
        self.m, self.n, self.k = int(m), int(n), int(k)
        self.density = list(density) if not np.isscalar(density) else [density, density]
        self.X = self.U = self.V = None

    def _seed(self, seed):
        if seed is None and not hasattr(self, "seed"):
            seed = int(time.time())
        if seed is not None:
            self.seed = seed
            self.rng = np.random.RandomState(seed)

    def generate(self, seed=None):
        self._seed(seed)
        U = _planted(self.m, self.k, self.density[0], self.rng)
        V = _planted(self.n, self.k, self.density[1], self.rng)
        self.U_order = self.rng.rand(self.m).argsort()
        self.U = U[self.U_order]
        self.V_order = self.rng.rand(self.n).argsort()
        self.V = V[self.V_order]
        self.X = (self.U.astype(np.int32) @ self.V.T.astype(np.int32) > 0).astype(np.uint8)
        return self

    def add_noise(self, noise=(0.0, 0.0), seed=None):
        self._seed(seed)
        self.noise = list(noise)
        p_pos, p_neg = noise
        drop = self.rng.binomial(size=self.X.shape, n=1, p=p_pos).astype(bool)
        X = self.X.astype(bool) & ~drop
        add = self.rng.binomial(size=self.X.shape, n=1, p=p_neg).astype(bool)
        self.X = (X | add).astype(np.uint8)
        return self


class PlantedBooleanOnDevice:
    """Row-sliceable lazy Boolean matrix for BitMatrix: X[a:b] is produced on the GPU.

    Planted factors follow the reference recipe (host RandomState, cheap: (m+n)*k draws); the Boolean product and
    the two noise flips are evaluated per row chunk with torch on the device (counter-based generator seeded per
    chunk, so any chunking / sharding of the rows yields the same matrix)."""

    def __init__(self, m, n, k, density=(0.2, 0.2), seed=1000, noise=(0.05, 0.01), noise_seed=2000, device="cuda:0"):
        import torch
        self.shape = (int(m), int(n))
        rng = np.random.RandomState(seed)
        U = _planted(m, k, density[0], rng)
        V = _planted(n, k, density[1], rng)
        U = U[rng.rand(m).argsort()]
        V = V[rng.rand(n).argsort()]
        self.device = torch.device(device)
        self.Ub = torch.from_numpy(U).to(self.device).to(torch.float16)
        self.Vt = torch.from_numpy(np.ascontiguousarray(V.T)).to(self.device).to(torch.float16)
        self.noise, self.noise_seed = noise, int(noise_seed)
        self.block = 1024  # rows per RNG block: the noise of a row depends only on (noise_seed, row // block)

    def __getitem__(self, sl):
        import torch
        a, b, step = sl.indices(self.shape[0])
        assert step == 1
        out = torch.empty((b - a, self.shape[1]), dtype=torch.uint8, device=self.device)
        r = a
        while r < b:
            blk = r // self.block
            lo, hi = blk * self.block, min((blk + 1) * self.block, self.shape[0])
            x = (self.Ub[lo:hi] @ self.Vt) > 0.5
            g = torch.Generator(device=self.device)
            g.manual_seed(self.noise_seed * 1000003 + blk)
            u = torch.rand((hi - lo, self.shape[1]), generator=g, device=self.device)
            x = x & ~(u < self.noise[0])
            u = torch.rand((hi - lo, self.shape[1]), generator=g, device=self.device)
            x = x | (u < self.noise[1])
            e = min(hi, b)
            out[r - a:e - a] = x[r - lo:e - lo].to(torch.uint8)
            r = e
        return out
