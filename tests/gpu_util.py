"""Helpers shared by the -m gpu parity tests."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def log(name, **kv):
    """Append a JSON line to gpurun_out/parity.jsonl (merged back by gpurun)."""
    os.makedirs(OUT, exist_ok=True)
    rec = {"test": name}
    for k, v in kv.items():
        if isinstance(v, (np.floating, np.integer)):
            v = v.item()
        elif isinstance(v, np.ndarray):
            v = v.tolist()
        rec[k] = v
    with open(os.path.join(OUT, "parity.jsonl"), "a") as f:
        f.write(json.dumps(rec) + "\n")


def corres_sets(corres_j, wrow_positive):
    """set of target indices per source row, restricted to entries with wij > 0."""
    return [set(int(j) for j, ok in zip(row, okrow) if ok) for row, okrow in zip(corres_j, wrow_positive)]
