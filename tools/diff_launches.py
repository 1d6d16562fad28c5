#!/usr/bin/env python
"""dev tool: per-launch-class A/B of two `ES_BENCH_DUMP=<file> python bench.py ...` dumps (every engine launch of one
single-stream step with its entry point, shape and stand-alone duration).  Launches are grouped by (entry point, K, C_in,
C_out, n_out); prints launches, total us in A and B, the difference, worst first.

  ES_BENCH_DUMP=a.jsonl python bench.py --no-cpu-baseline --no-other-configs --steps 4 --warmup 2
  ES_BENCH_DUMP=b.jsonl ES_DMA_MIN_CIN=0 python bench.py --no-cpu-baseline --no-other-configs --steps 4 --warmup 2
  python tools/diff_launches.py a.jsonl b.jsonl [--by kind]      # kind: fwd/dgrad vs wgrad x K x channels only"""
import argparse
import json
from collections import defaultdict


def load(path, by):
    g = defaultdict(lambda: [0, 0.0])
    for line in open(path):
        r = json.loads(line)
        kind = 'wgrad' if 'wgrad' in r['fn'] else 'fwd/dgrad'
        key = (kind, r['K'], r['cin'], r['cout']) if by == 'kind' else (r['fn'].replace('es_spconv_', ''), r['K'], r['cin'], r['cout'], r['n_out'])
        g[key][0] += 1
        g[key][1] += r['us']
    return g


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('a')
    ap.add_argument('b', nargs='?')
    ap.add_argument('--by', default='launch', choices=['launch', 'kind'])
    ap.add_argument('--top', type=int, default=40)
    args = ap.parse_args()
    A = load(args.a, args.by)
    B = load(args.b, args.by) if args.b else {}
    keys = sorted(set(A) | set(B), key=lambda k: -(A.get(k, [0, 0.0])[1] + B.get(k, [0, 0.0])[1]))
    ta, tb = sum(v[1] for v in A.values()), sum(v[1] for v in B.values())
    print(f'A: {sum(v[0] for v in A.values())} launches {ta / 1e3:.3f} ms' + (f'   B: {sum(v[0] for v in B.values())} launches {tb / 1e3:.3f} ms   B - A {(tb - ta) / 1e3:+.3f} ms' if B else ''))
    for k in keys[:args.top]:
        a, b = A.get(k, [0, 0.0]), B.get(k, [0, 0.0])
        line = f'{str(k):64s} {a[0]:3d} x {a[1]:8.1f} us'
        if B:
            line += f'   {b[0]:3d} x {b[1]:8.1f} us   {b[1] - a[1]:+8.1f}'
        print(line)


if __name__ == '__main__':
    main()
