"""Raw counter sums of the kernels whose name contains PATTERN in a rocprofv3 --pmc results db (per launch averages).
    python tools/pmc_kernel.py <p_results.db> <pattern>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_stats import pmc, short
db, pat = sys.argv[1], sys.argv[2]
agg, n, us = {}, 0, 0.0
for (did, kn, st, en), v in pmc(db).items():
    if pat not in kn:
        continue
    n += 1; us += (en - st) / 1e3
    for c, x in v.items():
        agg[c] = agg.get(c, 0.0) + x
print(f"{pat}: {n} launches, {us / max(n, 1):.1f} us each")
wc = agg.get("SQ_WAVE_CYCLES", 0.0)
for c in sorted(agg):
    print(f"  {c:28s} {agg[c] / max(n, 1):14.4g}" + (f"  ({agg[c] / wc * 100:5.1f} % of SQ_WAVE_CYCLES)" if wc and c.startswith("SQ_") else ""))
