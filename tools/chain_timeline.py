"""Timeline of the SCNet stream inside one forward of a bench trace (rocprofv3 rocpd db): per dispatch start offset, duration and the
gap to the previous dispatch of the same stream -- where do the small layers (conv6 .. deconv7 + reduce + finalize) spend their time?
    python tools/chain_timeline.py <db> [forward index from the end, default 5]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
t = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if 'kernel_dispatch' in x][0]; ks = [x for x in t if 'kernel_symbol' in x][0]
rows = list(db.execute(f"select s.kernel_name,d.start,d.end,d.stream_id,d.grid_size_x,d.grid_size_y from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
# the SCNet stream = the stream with the most conv_s2_strip dispatches
cnt = {}
for r in rows:
    if 'conv_s2_strip' in r[0]: cnt[r[3]] = cnt.get(r[3], 0) + 1
net = max(cnt, key=cnt.get)
ns = [r for r in rows if r[3] == net]
# forwards are delimited by the first conv_s2_tile (conv2) dispatch
starts = [i for i, r in enumerate(ns) if 'conv_s2_tile_kernelILi2ELi2' in r[0] and (i == 0 or 'conv_s2_tile_kernelILi2ELi2' not in ns[i-1][0])]
i0 = starts[-back]; i1 = starts[-back + 1]
seg = ns[i0:i1]
T0 = seg[0][1]; prev_end = None
tot_small = gap_small = 0.0
for r in seg:
    nm = r[0].replace('_ZN12_GLOBAL__N_1', '')[:48]
    gap = 0.0 if prev_end is None else (r[1] - prev_end) / 1e3
    dur = (r[2] - r[1]) / 1e3
    small = dur < 400
    if small: tot_small += dur; gap_small += max(gap, 0)
    print(f"{(r[1]-T0)/1e3:9.1f} us  dur {dur:8.1f}  gap {gap:7.1f}  grid ({r[4]//256 if r[4] else 0},{r[5]})  {nm}")
    prev_end = r[2]
print(f"forward wall {(seg[-1][2]-T0)/1e3:.1f} us; dispatches < 400 us: {tot_small:.1f} us busy + {gap_small:.1f} us of gaps in front of them")
