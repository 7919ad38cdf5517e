"""Summarise the rocprofv3 --pmc passes of tools/scnet_only.py (one forward at 64 images) into profiles/."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_stats import pmc, short

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"


def last_forward(db):
    items = sorted(pmc(db).items(), key=lambda kv: kv[0][2])
    idx = [i for i, (k, v) in enumerate(items) if 'resize_in' in k[1]]
    return items[idx[-1]:]


rd = last_forward(os.path.join(root, "pmc_FETCH_SIZE", "p_results.db"))
wr = last_forward(os.path.join(root, "pmc_WRITE_SIZE", "p_results.db"))
agg = {}
for seg, key in ((rd, "FETCH_SIZE"), (wr, "WRITE_SIZE")):
    for (did, kn, st, en), v in seg:
        k = short(kn)[:44]
        a = agg.setdefault(k, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0, "us": 0.0})
        a[key] += v.get(key, 0.0)
        if key == "FETCH_SIZE":
            a["n"] += 1; a["us"] += (en - st) / 1e3
print("One SCNet forward, 64 images (32 scan pairs), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes.")
print("Units: counters are KiB.  WRITE_SIZE calibrates 1.00x on resize_out (1.416 GB written).  FETCH_SIZE on gfx950")
print("reports 1/2 of a wide coalesced read (MI355X_MICROARCH.md, HBM section; bn_partial reads 8.5 GB and reports 4.25):")
print("the corrected column doubles it.")
print(f"{'kernel':46s} {'calls':>5s} {'time_us':>9s} {'fetch_GB':>9s} {'fetch_x2_GB':>11s} {'write_GB':>9s}")
tf = tw = 0
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
    f, w = a["FETCH_SIZE"] * 1024 / 1e9, a["WRITE_SIZE"] * 1024 / 1e9
    print(f"{k:46s} {a['n']:5d} {a['us']:9.1f} {f:9.2f} {2*f:11.2f} {w:9.2f}")
    if any(t in k for t in ('conv_igemm', 'conv1_direct', 'conv1_mfma', 'conv_s2_tile', 'conv_s2_strip', 'deconv_tile')):
        tf += f; tw += w
print(f"conv stack (conv_igemm + conv_s2_tile + conv_s2_strip + deconv_tile + conv1): fetch {tf:.2f} GB raw / {2*tf:.2f} GB corrected, write {tw:.2f} GB "
      f"-> HBM traffic per forward {2*tf+tw:.2f} GB (corrected)")
