"""dev tool: one steady-state train step (between the last two k_adamw launches) from a rocprofv3 rocpd database:
GPU busy time (union of kernel intervals), summed kernel time, idle gaps, and the per-(kernel, grid) table."""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
rows = db.execute('select start, end, name, grid_x, grid_y, grid_z from kernels order by start').fetchall()
marks = [i for i, r in enumerate(rows) if r[2].startswith('k_adamw')]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if len(marks) >= back:
    rows = rows[marks[-back] + 1: (marks[-back + 1] + 1) if back > 1 else None]
tot = sum(e - s for s, e, *_ in rows)
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e, *_ in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s; gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = rows[-1][1] - rows[0][0]
print(f'one step: kernels={len(rows)} span_ms={span / 1e6:.2f} sum_ms={tot / 1e6:.2f} busy_ms={busy / 1e6:.2f} idle_ms={(span - busy) / 1e6:.2f}')
gaps.sort(reverse=True)
print('largest gaps (us):', [round(g / 1e3) for g in gaps[:16]])
for lo, hi in ((0, 5e3), (5e3, 2e4), (2e4, 1e5), (1e5, 1e12)):
    sel = [g for g in gaps if lo <= g < hi]
    print(f'gaps in [{lo / 1e3:.0f},{hi / 1e3:.0f}) us: n={len(sel)} total_ms={sum(sel) / 1e6:.2f}')
agg = defaultdict(lambda: [0, 0])
for s, e, name, gx, gy, gz in rows:
    a = agg[(name[:44], gx, gy, gz)]; a[0] += 1; a[1] += e - s
for (name, gx, gy, gz), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f'{name:44s} grid=({gx},{gy},{gz}) calls={n:4d} total_us={t / 1e3:8.1f} avg_us={t / n / 1e3:8.1f}')
