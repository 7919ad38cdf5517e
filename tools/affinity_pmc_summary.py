"""Summarise the rocprofv3 --pmc passes of tools/affinity_pmc.py (one directory per counter set) per affinity kernel.
    python tools/affinity_pmc_summary.py ROOT TAG   (reads ROOT/TAG_<set>/p_results.db for every set that exists)
Counter sets (tools/gpu_evidence.sh affinity): fetch (FETCH_SIZE), write (WRITE_SIZE), sqa (instruction mix), sqb (wait / issue breakdown), sqc (LDS)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_stats import pmc, short

root, tag = sys.argv[1], sys.argv[2]


def agg_pass(name):
    db = os.path.join(root, f"{tag}_{name}", "p_results.db")
    if not os.path.exists(db):
        return {}
    out = {}
    for (did, kn, st, en), v in sorted(pmc(db).items(), key=lambda kv: kv[0][2]):
        if "affinity" not in kn and "aff3" not in kn:
            continue
        k = short(kn)[:90]
        a = out.setdefault(k, {"n": 0, "us": 0.0})
        a["n"] += 1
        a["us"] += (en - st) / 1e3
        for c, x in v.items():
            a[c] = a.get(c, 0.0) + x
    return out


P = {n: agg_pass(n) for n in ("fetch", "write", "sqa", "sqb", "sqc")}
kernels = sorted({k for p in P.values() for k in p})
print("per-launch averages; FETCH_SIZE / WRITE_SIZE are KiB counters (FETCH x2 = the gfx950 correction of MI355X_MICROARCH.md, HBM section)")
for k in kernels:
    print("==", k)
    f, w = P["fetch"].get(k), P["write"].get(k)
    if f:
        print(f"   time {f['us'] / f['n']:9.1f} us/launch ({f['n']} launches)   FETCH_SIZE {f.get('FETCH_SIZE', 0) * 1024 / f['n'] / 1e6:9.2f} MB (x2 = {2 * f.get('FETCH_SIZE', 0) * 1024 / f['n'] / 1e6:9.2f} MB)")
    if w:
        print(f"   time {w['us'] / w['n']:9.1f} us/launch   WRITE_SIZE {w.get('WRITE_SIZE', 0) * 1024 / w['n'] / 1e6:9.2f} MB")
    for s in ("sqa", "sqb", "sqc"):
        a = P[s].get(k)
        if not a:
            continue
        n = a["n"]
        wc = a.get("SQ_WAVE_CYCLES", 0.0)
        line = f"   [{s}] {a['us'] / n:9.1f} us"
        for c in sorted(a):
            if c in ("n", "us"):
                continue
            line += f"  {c.replace('SQ_', '')}={a[c] / n:.4g}"
            if wc and c.startswith("SQ_") and c not in ("SQ_WAVE_CYCLES", "SQ_WAVES") and ("WAIT" in c or "ACTIVE" in c):
                line += f"({a[c] / wc * 100:.1f}%)"
        if "SQ_WAVES" in a and "GRBM_GUI_ACTIVE" in a and a["GRBM_GUI_ACTIVE"]:
            wall = a["GRBM_GUI_ACTIVE"] / 8          # summed over the 8 XCDs
            # SQ_WAVE_CYCLES counts quad-cycles per wave: mean resident waves per SIMD = 4 * wave_cycles / (wall * 1024 SIMDs)
            line += f"  | mean waves/SIMD = {4 * wc / (wall * 1024):.2f}, clk = {wall / (a['us'] / n) / n / 1e3:.2f} GHz"
        print(line)
