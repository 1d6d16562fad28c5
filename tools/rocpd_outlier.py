"""dev tool (round 5): which step of a rocprofv3 kernel trace is slow, and why -- per step (between two k_adamw* launches): span, kernel
busy time (union over the queues), the longest idle gap of the whole device with the kernels either side of it, and the kernels whose
duration exceeds 3x their median over the trace.
  python tools/rocpd_outlier.py <db>"""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = db.execute('select start, end, name from kernels order by start').fetchall()
marks = [i for i, r in enumerate(rows) if r[2].startswith('k_adamw')]
med = defaultdict(list)
for b, e, n in rows:
    med[n].append(e - b)
med = {n: sorted(v)[len(v) // 2] for n, v in med.items()}
print(f'{len(rows)} kernels, {len(marks)} optimiser launches')
for k in range(len(marks) - 1):
    rs = rows[marks[k] + 1: marks[k + 1] + 1]
    t0, t1 = rs[0][0], max(r[1] for r in rs)
    busy, cur_b, cur_e = 0, rs[0][0], rs[0][1]
    gap, gap_at = 0, None
    for b, e, n in rs[1:]:
        if b > cur_e:
            busy += cur_e - cur_b
            if b - cur_e > gap:
                gap, gap_at = b - cur_e, (cur_e - t0, n)
            cur_b, cur_e = b, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_b
    slow = [(n[:50], (e - b) / 1e3, med[n] / 1e3) for b, e, n in rs if e - b > 3 * med[n] + 100e3]
    print(f'step {k:3d}: span {(t1 - t0) / 1e6:7.2f} ms  device busy {busy / 1e6:7.2f} ms  longest gap {gap / 1e6:6.2f} ms'
          + (f' at {gap_at[0] / 1e6:.2f} ms before {gap_at[1][:40]}' if gap_at else '')
          + (f'  slow kernels (us vs median): {slow[:4]}' if slow else ''))
