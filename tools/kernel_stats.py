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
    """Per-group conv timing of the `which`-th forward in the trace (launch order is fixed, csrc/scnet.hip)."""
    seq, on, count = [], False, 0
    for r in rows:
        if 'resize_in' in r[0]:
            count += 1
            on = (count == which)
        if on:
            seq.append(r)
        if 'resize_out' in r[0] and on:
            break
    P = lambda M, K, C: 2.0 * M * K * C
    G = [('conv2 x6', 6 * P(n * 112 * 112, 512, 64)), ('conv3 x6', 6 * P(n * 56 * 56, 1024, 128)),
         ('conv4', P(n * 784, 12288, 256)), ('conv5', P(n * 196, 4096, 512)), ('conv6', P(n * 49, 8192, 512)),
         ('conv7', P(n * 9, 4608, 512)), ('conv8', P(n * 9, 4608, 512)), ('conv9', P(n, 4608, 1024)),
         ('deconv9', P(n * 9, 9216, 512)), ('deconv8', P(n * 9, 9216, 512)),
         ('deconv7', sum(P(n * a * b, t * 1024, 512) for a, b, t in ((4, 4, 4), (4, 3, 2), (3, 4, 2), (3, 3, 1)))),
         ('deconv6', 4 * P(n * 49, 4096, 512)), ('deconv5', 4 * P(n * 196, 4096, 256)), ('deconv4', 4 * P(n * 784, 2048, 128)),
         ('deconv3 x5', 4 * (3 * P(n * 3136, 1024, 64) + 2 * P(n * 3136, 512, 64))),
         ('deconv2 rgb/n/d', 12 * P(n * 12544, 512, 32)), ('deconv2 s/f', 8 * P(n * 12544, 256, 64)),
         ('heads', sum(P(n * 50176, 64, c) for c in (3, 3, 1, 15, 32)))]
    isconv = lambda nm: 'conv_igemm' in nm or 'deconv_tile' in nm or 'conv_s2_tile' in nm or 'conv_s2_strip' in nm      # the implicit-GEMM launches, in network order
    convs = [r for r in seq if isconv(r[0])]
    c1 = [r for r in seq if 'conv1_direct' in r[0] or 'conv1_mfma' in r[0]]
    red = sum((r[2] - r[1]) / 1e3 for r in seq if 'splitk_reduce' in r[0])
    out = [f"{len(convs)} conv launches in forward #{which} (expected {len(G)})"]
    tot = sum((r[2] - r[1]) / 1e3 for r in convs)
    for r, (name, fl) in zip(convs, G):
        d = (r[2] - r[1]) / 1e3
        out.append(f"{name:16s} grid=({r[3]//256:5d},{r[4]:3d},{r[5]:2d}) time={d:9.1f}us {100*d/tot:5.1f}%  useful TFLOP/s={fl/d/1e6:6.1f}")
    if c1:
        d1 = (c1[0][2] - c1[0][1]) / 1e3
        f1 = 2.0 * n * 224 * 224 * 192 * 36 * (10 / 12)      # 4 blocks with 36 and 2 blocks with 18 real MACs per output
        out.append(f"{'conv1':16s} time={d1:9.1f}us  useful TFLOP/s={f1/d1/1e6:6.1f} ({'MFMA' if 'mfma' in c1[0][0] else 'VALU'}; writes {n*224*224*192*4/1e9:.2f} GB -> {n*224*224*192*4/d1/1e3:.0f} GB/s)")
    out.append(f"conv total {tot:.1f} us (+ split-K reduce {red:.1f} us), useful {sum(f for _, f in G)/tot/1e6:.1f} TFLOP/s; "
               f"forward wall {(seq[-1][2]-seq[0][1])/1e3:.1f} us")
    oth = {}
    for r in seq:
        if not isconv(r[0]):
            oth[short(r[0])[:24]] = oth.get(short(r[0])[:24], 0) + (r[2] - r[1]) / 1e3
    out.append("other kernels (us): " + ", ".join(f"{k}={v:.0f}" for k, v in oth.items()))
    return "\n".join(out)


if __name__ == "__main__" and len(sys.argv) > 2:
    print(conv_layers(load(sys.argv[1]), int(sys.argv[2])))


def pmc(db):
    """{(dispatch_id, kernel, start, end): {counter: summed value}} from a rocprofv3 --pmc run."""
    c = sqlite3.connect(db)
    t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    T = lambda k: [x for x in t if k in x][0]
    pe, ip, kd, ks = T('rocpd_pmc_event'), T('rocpd_info_pmc'), T('kernel_dispatch'), T('kernel_symbol')
    out = {}
    for name, st, en, did, cname, val in c.execute(
            f"select s.kernel_name, d.start, d.end, d.id, p.name, e.value from {pe} e join {ip} p on e.pmc_id=p.id "
            f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id"):
        d = out.setdefault((did, name, st, en), {})
        d[cname] = d.get(cname, 0.0) + val
    return out
