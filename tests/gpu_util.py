"""Helpers shared by the -m gpu parity tests."""
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


# one file per test run (gpurun merges gpurun_out/ back by file name: a shared parity.jsonl written by a later partial
# run would replace the full log of an earlier one)
RUN_LOG = os.path.join(OUT, "parity", time.strftime("%Y%m%d-%H%M%S") + f"-{os.getpid()}.jsonl")


def log(name, **kv):
    """Append a JSON line to this run's gpurun_out/parity/<start time>-<pid>.jsonl (merged back by gpurun)."""
    os.makedirs(os.path.dirname(RUN_LOG), exist_ok=True)
    rec = {"test": name}
    for k, v in kv.items():
        if isinstance(v, (np.floating, np.integer)):
            v = v.item()
        elif isinstance(v, np.ndarray):
            v = v.tolist()
        rec[k] = v
    with open(RUN_LOG, "a") as f:
        f.write(json.dumps(rec) + "\n")


def corres_sets(corres_j, wrow_positive):
    """set of target indices per source row, restricted to entries with wij > 0."""
    return [set(int(j) for j, ok in zip(row, okrow) if ok) for row, okrow in zip(corres_j, wrow_positive)]
