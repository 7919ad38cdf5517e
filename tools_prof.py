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


def conv_layers(rows, n=64, which=3):
    """Per-layer conv timing of the `which`-th forward in the trace (launch order is fixed, csrc/scnet.hip)."""
    seq, on, count = [], False, 0
    for r in rows:
        if 'resize_in' in r[0]:
            count += 1
            on = (count == which)
        if on:
            seq.append(r)
        if 'resize_out' in r[0] and on:
            break
    L = []
    add = lambda name, M, K, C: L.append((name, M, K, C))
    add('conv1', n * 224 * 224, 144, 192)
    for q in range(6): add('conv2', n * 112 * 112, 16 * 32, 64)
    for q in range(6): add('conv3', n * 56 * 56, 16 * 64, 128)
    add('conv4', n * 28 * 28, 16 * 768, 256); add('conv5', n * 14 * 14, 16 * 256, 512); add('conv6', n * 7 * 7, 16 * 512, 512)
    add('conv7', n * 9, 9 * 512, 512); add('conv8', n * 9, 9 * 512, 512); add('conv9', n * 1, 9 * 512, 1024)
    add('deconv9', n * 9, 9 * 1024, 512); add('deconv8', n * 9, 9 * 1024, 512)
    for (hp, wp, t) in ((4, 4, 4), (4, 3, 2), (3, 4, 2), (3, 3, 1)): add('deconv7', n * hp * wp, t * 1024, 512)
    for ph in range(4): add('deconv6', n * 49, 4 * 1024, 512)
    for ph in range(4): add('deconv5', n * 196, 4 * 1024, 256)
    for ph in range(4): add('deconv4', n * 784, 4 * 512, 128)
    for m in range(5):
        for ph in range(4): add('deconv3', n * 56 * 56, 4 * (256 if m < 3 else 128), 64)
    for m in range(5):
        for ph in range(4): add('deconv2', n * 112 * 112, 4 * (128 if m < 3 else 64), 32 if m < 3 else 64)
    for m, co in enumerate((3, 3, 1, 15, 32)): add('head', n * 224 * 224, 64, co)
    convs = [r for r in seq if 'conv_igemm' in r[0]]
    out = [f"{len(convs)} conv launches in forward #{which} (expected {len(L)})"]
    agg = {}
    for r, (name, M, K, C) in zip(convs, L):
        a = agg.setdefault(name, [0.0, 0.0, 0])
        a[0] += (r[2] - r[1]) / 1e3; a[1] += 2.0 * M * K * C; a[2] += 1
    tot = sum(a[0] for a in agg.values())
    for k, a in agg.items():
        out.append(f"{k:8s} launches={a[2]:2d} time={a[0]:9.1f}us {100*a[0]/tot:5.1f}%  useful TFLOP/s={a[1]/a[0]/1e6:6.1f}")
    out.append(f"conv total {tot:.1f} us, useful {sum(a[1] for a in agg.values())/tot/1e6:.1f} TFLOP/s; forward wall {(seq[-1][2]-seq[0][1])/1e3:.1f} us")
    return "\n".join(out)


if __name__ == "__main__" and len(sys.argv) > 2:
    print(conv_layers(load(sys.argv[1]), int(sys.argv[2])))
