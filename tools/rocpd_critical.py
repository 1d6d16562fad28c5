"""dev tool (round 4): the step's DEPENDENT CHAIN from a rocprofv3 rocpd database -- per stream, for one steady step (between two
k_adamw launches): kernels, busy time, and the idle gaps between consecutive kernels of that stream; then the main stream's
kernels grouped by name with their summed duration and the gap that follows each (launch / dependency latency the chain pays).
  python tools/rocpd_critical.py <db> [steps back from the end]"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info('kernels')")]
key = 'stream_id' if 'stream_id' in cols else 'queue_id'
if key == 'stream_id' and 'queue_id' in cols and len(db.execute('select distinct stream_id from kernels').fetchall()) <= 1:
    key = 'queue_id'                      # (kernel-trace only: every stream_id is 0; the hardware queue tells the streams apart)
rows = db.execute(f'select start, end, name, {key} from kernels order by start').fetchall()
marks = [i for i, r in enumerate(rows) if r[2].startswith('k_adamw')]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = rows[marks[-back - 1] + 1: marks[-back] + 1]
t0, t1 = rows[0][0], max(r[1] for r in rows)
print(f'step span {(t1 - t0) / 1e6:.2f} ms, {len(rows)} kernels')
by = defaultdict(list)
for r in rows:
    by[r[3]].append(r)
for s, rs in sorted(by.items()):
    busy = sum(e - b for b, e, _, _ in rs)
    gaps = [max(0, rs[i + 1][0] - rs[i][1]) for i in range(len(rs) - 1)]
    small = [g for g in gaps if g < 50e3]
    print(f'stream {s}: {len(rs)} kernels, busy {busy / 1e6:.2f} ms, first..last {(rs[0][0] - t0) / 1e6:.2f}..{(rs[-1][1] - t0) / 1e6:.2f} ms, '
          f'gaps < 50 us: {len(small)} totalling {sum(small) / 1e6:.2f} ms (median {sorted(small)[len(small) // 2] / 1e3 if small else 0:.1f} us), '
          f'longer gaps {sum(g for g in gaps if g >= 50e3) / 1e6:.2f} ms')
main = max(by.items(), key=lambda kv: len(kv[1]))[0]
rs = by[main]
agg = defaultdict(lambda: [0, 0.0, 0.0])
for i, (b, e, n, _) in enumerate(rs):
    g = max(0, rs[i + 1][0] - e) if i + 1 < len(rs) else 0
    a = agg[n[:60]]
    a[0] += 1
    a[1] += e - b
    a[2] += min(g, 50e3)
print(f'\nmain stream {main}: kernel groups by busy + following gap')
print(f'{"kernel":60s} {"calls":>6s} {"busy_ms":>9s} {"gap_ms":>8s} {"avg_us":>8s}')
for n, (c, bsy, g) in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2]))[:45]:
    print(f'{n:60s} {c:6d} {bsy / 1e6:9.3f} {g / 1e6:8.3f} {bsy / c / 1e3:8.1f}')

# the longest idle gaps of the main stream with the kernels on either side and what the OTHER streams ran meanwhile (round 5)
print('\nlongest main-stream gaps: start_ms gap_us | before -> after | busiest other-stream kernel inside the gap')
gl = sorted(((rs[i + 1][0] - rs[i][1], i) for i in range(len(rs) - 1)), reverse=True)[:28]
others = [r for s, v in by.items() if s != main for r in v]
for g, i in sorted(gl, key=lambda t: t[1]):
    b0, b1 = rs[i][1], rs[i + 1][0]
    inside = defaultdict(float)
    for b, e, n, s in others:
        ov = min(e, b1) - max(b, b0)
        if ov > 0:
            inside[(s, n[:38])] += ov
    top = max(inside.items(), key=lambda kv: kv[1]) if inside else (('-', '-'), 0.0)
    print(f'{(b0 - t0) / 1e6:7.2f} {g / 1e3:8.1f} | {rs[i][2][:34]:34s} -> {rs[i + 1][2][:34]:34s} | s{top[0][0]} {top[0][1]} {top[1] / 1e3:.0f} us')
