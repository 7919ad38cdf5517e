"""Summarise a rocprofv3 rocpd database: per-kernel totals and (optionally) per-layer conv timings."""
import sqlite3
import sys


def load(db):
    c = sqlite3.connect(db)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if 'kernel_dispatch' in x][0]
    ks = [x for x in t if 'kernel_symbol' in x][0]
    return list(c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))


def short(n):
    n = n.replace('_ZN12_GLOBAL__N_1', '').replace('.kd', '')
    return n[:70]


def summary(rows, top=25):
    agg = {}
    for r in rows:
        a = agg.setdefault(r[0], [0, 0.0, 1e30, 0.0])
        d = (r[2] - r[1]) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    out = [f"total kernel time {tot/1e3:.3f} ms over {len(rows)} dispatches"]
    out.append(f"{'kernel':70s} {'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'pct':>6s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        out.append(f"{short(k):70s} {a[0]:6d} {a[1]/1e3:10.3f} {a[1]/a[0]:10.1f} {a[2]:9.1f} {a[3]:10.1f} {100*a[1]/tot:6.1f}")
    return "\n".join(out)


if __name__ == "__main__":
    rows = load(sys.argv[1])
    print(summary(rows))
