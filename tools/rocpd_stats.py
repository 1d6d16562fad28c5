"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a --stats style text table."""
import sqlite3
import sys


def main(db_path, out_path=None, skip_first_frac=0.0):
    db = sqlite3.connect(db_path)
    rows = db.execute('select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels '
                      'group by name order by sum(duration) desc').fetchall()
    tot = sum(r[2] for r in rows)
    lines = [f'# kernel-trace summary of {db_path}', f'# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches',
             f'{"kernel":70s} {"calls":>7s} {"total_ms":>10s} {"avg_us":>10s} {"min_us":>9s} {"max_us":>9s} {"pct":>6s}']
    for name, n, s, a, mn, mx in rows:
        lines.append(f'{name[:70]:70s} {n:7d} {s / 1e6:10.3f} {a / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * s / tot:6.2f}')
    txt = '\n'.join(lines)
    if out_path:
        open(out_path, 'w').write(txt + '\n')
    print(txt)


if __name__ == '__main__':
    main(*sys.argv[1:3])
