"""Shared helpers for the parity tests: load golden fixtures and turn them into oracle Specs."""
import os

import numpy as np

from oracle import mlp_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name))


def groups(npz):
    """Group 'a/b/c/field' keys of an npz into {'a/b/c': {field: array}}."""
    out = {}
    for k in npz.files:
        g, f = k.rsplit("/", 1)
        out.setdefault(g, {})[f] = npz[k]
    return out


def subgroups(npz, prefix):
    """{'name': {field: array}} for the keys 'prefix/name/field' of an npz."""
    out = {}
    for k in npz.files:
        if k.startswith(prefix + "/"):
            name, field = k[len(prefix) + 1:].split("/", 1)
            out.setdefault(name, {})[field] = npz[k]
    return out


def spec_from(rec, dtype=np.float64, temperature=None):
    t = temperature
    if t is None and "temperature" in rec and not np.isnan(rec["temperature"]):
        t = float(rec["temperature"])
    return orc.Spec(rec["dims"].tolist(), rec["acts"].tolist(), int(rec["lik"]),
                    mu=rec["prior_mu"].astype(dtype), sigma=rec["prior_sigma"].astype(dtype), temperature=t)
