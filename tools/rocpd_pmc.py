"""Per-kernel sums of the PMC counters in a rocprofv3 rocpd database -> small text table."""
import sqlite3
import sys


def main(db_path, out_path=None, top=400):
    db = sqlite3.connect(db_path)
    rows = db.execute('select kernel_name, counter_name, sum(value), count(*), sum(duration) from counters_collection '
                      'group by kernel_name, counter_name').fetchall()
    per = {}
    for kn, cn, v, n, d in rows:
        e = per.setdefault(kn, {'_calls': n, '_dur_ms': d / 1e6})
        e[cn] = v
    names = sorted({cn for _, cn, _, _, _ in rows})
    order = sorted(per, key=lambda k: -per[k]['_dur_ms'])[:top]
    lines = [f'# PMC sums per kernel from {db_path}', 'kernel | calls | dur_ms | ' + ' | '.join(names)]
    for kn in order:
        e = per[kn]
        lines.append(f"{kn[:64]} | {e['_calls']} | {e['_dur_ms']:.3f} | " + ' | '.join(f'{e.get(c, 0):.4g}' for c in names))
    txt = '\n'.join(lines)
    if out_path:
        open(out_path, 'w').write(txt + '\n')
    print(txt)


if __name__ == '__main__':
    main(*sys.argv[1:3])
