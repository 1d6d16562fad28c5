"""dev tool: per-stream timeline of one steady train step (between the last two k_adamw launches) from a rocprofv3
rocpd database: per 1-ms bin, busy fraction of each stream and the kernel that took most of the bin."""
import sqlite3, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info('kernels')")]
print('columns:', cols)
key = 'stream_id' if 'stream_id' in cols else ('queue_id' if 'queue_id' in cols else None)
if key == 'stream_id' and 'queue_id' in cols and len(db.execute('select distinct stream_id from kernels').fetchall()) <= 1:
    key = 'queue_id'
rows = db.execute(f'select start, end, name, {key or 0} from kernels order by start').fetchall()
marks = [i for i, r in enumerate(rows) if r[2].startswith('k_adamw')]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3     # which step, counted from the end (k_adamw marks)
if len(marks) >= back:
    rows = rows[marks[-back] + 1: marks[-back + 1] + 1]
t0, t1 = rows[0][0], max(r[1] for r in rows)
streams = sorted({r[3] for r in rows})
print(f'step span {(t1 - t0) / 1e6:.2f} ms, streams {streams}')
for s in streams:
    tot = sum(e - b for b, e, _, q in rows if q == s)
    print(f'  stream {s}: {sum(1 for r in rows if r[3] == s)} kernels, {tot / 1e6:.2f} ms busy')
BIN = 1e6
nb = int((t1 - t0) / BIN) + 1
busy = defaultdict(lambda: [0.0] * nb)
top = [defaultdict(float) for _ in range(nb)]
for b, e, name, q in rows:
    i = int((b - t0) / BIN)
    while b < e:
        lim = min(e, t0 + (i + 1) * BIN)
        busy[q][i] += lim - b
        top[i][name[:28]] += lim - b
        b = lim; i += 1
for i in range(nb):
    fr = ' '.join(f'{busy[s][i] / BIN:4.2f}' for s in streams)
    k = max(top[i].items(), key=lambda kv: kv[1])[0] if top[i] else ''
    print(f'{i:3d} ms | {fr} | {k}')
