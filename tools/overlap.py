"""Stream-overlap analysis of a rocprofv3 (rocpd) trace of bench.py: how much of the timed window has 0 / 1 / 2
SCNet kernels in flight, and per-stream busy time split into SCNet and matcher/geometry kernels."""
import sqlite3, collections, sys
db = sqlite3.connect(sys.argv[1])
win = float(sys.argv[2]) if len(sys.argv) > 2 else 250.0
t = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if 'kernel_dispatch' in x][0]; ks = [x for x in t if 'kernel_symbol' in x][0]
rows = list(db.execute(f"select s.kernel_name,d.start,d.end,d.queue_id,d.stream_id from {kd} d join {ks} s on d.kernel_id=s.id where d.stream_id != 0 order by d.start"))
t_end = max(r[2] for r in rows); T0 = t_end - win * 1e6
sel = [r for r in rows if r[1] >= T0]
def union(iv):
    iv = sorted(iv)
    if not iv: return 0
    tot = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: tot += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    return tot + ce - cs
isnet = lambda n: ('conv' in n or 'heads' in n or 'resize' in n or 'bn_' in n or 'splitk' in n)
print("window ms", win, "any kernel busy", union([(r[1], r[2]) for r in sel]) / 1e6, "SCNet busy", union([(r[1], r[2]) for r in sel if isnet(r[0])]) / 1e6)
ev = []
for r in sel:
    if isnet(r[0]): ev += [(r[1], 1), (r[2], -1)]
ev.sort(); c = 0; last = ev[0][0]; acc = collections.Counter()
for tt, dv in ev:
    acc[c] += tt - last; last = tt; c += dv
print("ms with k SCNet kernels in flight:", {k: round(v / 1e6, 2) for k, v in sorted(acc.items())})
for st in sorted(set(r[4] for r in sel)):
    ss = [r for r in sel if r[4] == st]
    print("stream", st, "busy", round(union([(r[1], r[2]) for r in ss]) / 1e6, 2), "SCNet", round(union([(r[1], r[2]) for r in ss if isnet(r[0])]) / 1e6, 2),
          "matcher+geometry", round(union([(r[1], r[2]) for r in ss if not isnet(r[0])]) / 1e6, 2), "sum kernel time SCNet", round(sum(r[2]-r[1] for r in ss if isnet(r[0]))/1e6,2))
# timeline of phases for the first stream: print segments
if len(sys.argv) > 3:
    for st in sorted(set(r[4] for r in sel)):
        ss = [r for r in sel if r[4] == st]; seg = []; cur = None
        for r in ss:
            k = 'N' if isnet(r[0]) else 'M'
            if cur and cur[0] == k and r[1] - cur[2] < 0.5e6: cur[2] = r[2]
            else:
                cur = [k, r[1], r[2]]; seg.append(cur)
        print("stream", st, " ".join(f"{k}[{(a-T0)/1e6:.1f}-{(b-T0)/1e6:.1f}]" for k, a, b in seg))
