"""Per-kernel SQ counter summary of a rocprofv3 --pmc pass over tools/scnet_only.py (last forward)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_stats import pmc, short
db = sys.argv[1]
items = sorted(pmc(db).items(), key=lambda kv: kv[0][2])
idx = [i for i, (k, v) in enumerate(items) if 'resize_in' in k[1]]
items = items[idx[-1]:]
agg = {}
for (did, kn, st, en), v in items:
    k = short(kn)[:50]
    a = agg.setdefault(k, {"n": 0, "us": 0.0})
    a["n"] += 1; a["us"] += (en - st) / 1e3
    for c, x in v.items():
        a[c] = a.get(c, 0.0) + x
names = sorted({c for a in agg.values() for c in a if c not in ("n", "us")})
print("counters:", names)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["us"])[:12]:
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    line = f"{k:52s} n={a['n']:3d} {a['us']:9.1f} us"
    if wc:
        for c in names:
            if c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES"):
                line += f" {c[3:]}={a.get(c, 0.0) / wc * 100:5.1f}%"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in a and "GRBM_GUI_ACTIVE" in a and a["GRBM_GUI_ACTIVE"]:
        # MFMA busy cycles are summed over the 1024 SIMDs (64 per v_mfma_f32_32x32x2_f32: checked against the MFMA count of
        # conv3); GRBM_GUI_ACTIVE is summed over the 8 XCDs, so wall cycles = GUI / 8
        wall = a['GRBM_GUI_ACTIVE'] / 8
        line += f" | MFMA busy = {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (wall * 1024) * 100:5.1f}% of SIMD-cycles, clk={wall / a['us'] / 1e3:.2f} GHz"
    print(line)
