"""dev tool (round 5): what runs between two marker kernels of one steady train step (rocprofv3 rocpd database, single-stream run):
    python tools/rocpd_window.py <db> <first-kernel-prefix> <last-kernel-prefix> [steps back]
kernels of the window grouped by name (calls, busy ms), the window's span and its idle time, and the 15 longest gaps."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
a, b = sys.argv[2], sys.argv[3]
back = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rows = db.execute('select start, end, name from kernels order by start').fetchall()
marks = [i for i, r in enumerate(rows) if r[2].startswith('k_adamw')]
rows = rows[marks[-back - 1] + 1: marks[-back] + 1]
ia = max(i for i, r in enumerate(rows) if r[2].startswith(a) or a in r[2][:60])      # LAST occurrence of the first marker
ib = (max if b.endswith('$') else min)(i for i, r in enumerate(rows) if i > ia and (r[2].startswith(b.rstrip('$')) or b.rstrip('$') in r[2][:60]))   # 'name$': LAST occurrence
win = rows[ia + 1: ib + 1]
t0, t1 = rows[ia][1], win[-1][1]
busy = sum(e - s for s, e, _ in win)
print(f'window after the last {a}* up to the first {b}*: {len(win)} kernels, span {(t1 - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {(t1 - t0 - busy) / 1e6:.3f} ms')
g = defaultdict(lambda: [0, 0.0])
for s, e, n in win:
    g[n[:70]][0] += 1
    g[n[:70]][1] += e - s
for n, (c, t) in sorted(g.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f'{n:70s} {c:5d} {t / 1e6:8.3f} ms {t / c / 1e3:8.1f} us')
gaps = sorted(((win[i + 1][0] - win[i][1], i) for i in range(len(win) - 1)), reverse=True)[:15]
print('longest gaps (us): before -> after')
for gp, i in sorted(gaps, key=lambda t: t[1]):
    print(f'  {(win[i][1] - t0) / 1e6:7.3f} ms {gp / 1e3:8.1f} | {win[i][2][:40]:40s} -> {win[i + 1][2][:40]}')
