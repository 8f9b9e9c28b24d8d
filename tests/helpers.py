"""Shared helpers for the parity tests (fixture loading; oracle replay)."""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch

from oracle import fb_oracle as fo

GOLDEN = Path(__file__).resolve().parent / "golden"


def load_meta(name: str) -> dict:
    return json.loads((GOLDEN / f"{name}.json").read_text())


def cfg_from_meta(meta: dict) -> fo.OracleConfig:
    return fo.OracleConfig(**meta["cfg"])


def regenerate_inputs(meta: dict):
    """Recreate (nets, storage, lengths, rng) exactly as tests/golden/make_golden.py::trace_fixture did."""
    cfg = cfg_from_meta(meta)
    rng = np.random.default_rng(meta["seed"])
    nets = {n: fo.synthetic_params(rng, fo.NET_SHAPES[n](cfg)) for n in ("actor", "forward_net", "backward_net")}
    lengths = None
    n_eps, T = meta["n_eps"], meta["T"]
    if meta["variable_len"]:
        lengths = rng.integers(max(2, T // 2), T + 1, size=n_eps).astype(np.int32)
        lengths[0] = T
    storage, lengths = fo.synthetic_storage(rng, n_eps, T, cfg.obs_dim, cfg.action_dim,
                                            cfg.goal_dim if cfg.use_goal else None, lengths)
    return cfg, nets, storage, lengths, rng


def checksums(state: dict) -> dict:
    return {k: [float(np.sum(v, dtype=np.float64)), float(np.sqrt(np.sum(np.asarray(v, np.float64) ** 2)))]
            for k, v in state.items() if not k.startswith("adam_")}


def rel_err(a, b) -> float:
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


LOSS_KEYS = ("fb_loss", "fb_offdiag", "fb_diag", "orth_loss", "orth_loss_offdiag", "actor_loss", "q")
