"""profiles/r2_pmc_fetch.txt + r2_pmc_write.txt (tools/rocpd_pmc.py tables of two separate rocprofv3 --pmc passes of
`bench.py --no-cpu-baseline --steps 4 --warmup 2`) -> profiles/r2_pmc_traffic.json: HBM-side bytes of the convolution-engine
family per step and per launch.  FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE is doubled per the gfx950 note of
MI355X_MICROARCH.md (128-B requests tallied at 64 B); WRITE_SIZE as reported."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(path):
    rows = {}
    lines = open(path).read().splitlines()
    names = [c.strip() for c in lines[1].split('|')]
    for l in lines[2:]:
        c = [x.strip() for x in l.split('|')]
        rows[c[0]] = dict(zip(names[1:], [float(v) for v in c[1:]]))
    return rows


def main(steps=7):
    f, w = table(os.path.join(ROOT, 'profiles', 'r2_pmc_fetch.txt')), table(os.path.join(ROOT, 'profiles', 'r2_pmc_write.txt'))
    fam = [k for k in f if 'k_spconv' in k or 'k_rowgemm' in k]      # every kernel behind the es_spconv_* entry points
    fetch = sum(f[k]['FETCH_SIZE'] for k in fam) * 1024 * 2
    write = sum(w[k]['WRITE_SIZE'] for k in fam if k in w) * 1024
    launches = int(sum(f[k]['calls'] for k in fam))
    out = dict(source='rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum / --pmc WRITE_SIZE TCC_MISS_sum (two separate passes) of '
                      '`python bench.py --no-cpu-baseline --steps 4 --warmup 2` (7 steps incl. the extra single-stream step); tables: '
                      'profiles/r2_pmc_fetch.txt, r2_pmc_write.txt; FETCH_SIZE (KB) doubled per the gfx950 note of MI355X_MICROARCH.md, '
                      'WRITE_SIZE (KB) as reported',
               family=sorted(fam), steps=steps, launches=launches, fetch_bytes=fetch, write_bytes=write,
               bytes_per_step=int((fetch + write) / steps), bytes_per_launch=int((fetch + write) / max(launches, 1)))
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'r2_pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k not in ('source', 'family')}))


if __name__ == '__main__':
    main()
