"""Per-kernel PMC tables (tools/rocpd_pmc.py output of two separate rocprofv3 --pmc passes: FETCH_SIZE, WRITE_SIZE) ->
profiles/<tag>_pmc_traffic[_<config>].json: HBM-side bytes of a kernel family per step and per launch, and the achieved
HBM GB/s of the scatter path from the counters (north_star).  FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH_SIZE is
doubled per the gfx950 note of MI355X_MICROARCH.md (128-B requests tallied at 64 B); WRITE_SIZE as reported.

  python tools/pmc_traffic.py --tag r3 --config mv3ddet --steps 7 --fetch profiles/r3_pmc_fetch.txt --write profiles/r3_pmc_write.txt
"""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILY = {
    # the convolution engine: every kernel behind the es_spconv_* entry points (+ their split reductions)
    # round 6: + the halo kernel (k_spconv_halo) and its plan, the image-grid kernels (k_img_conv3, k_img_wgrad9 + reduce, k_rows_wgrad1), the few-row
    # linear kernels and the expansion stream kernel -- every kernel bench.py's ENGINE entry points can launch
    'mv3ddet': ('k_spconv', 'k_rowgemm', 'k_wgrad_reduce', 'k_sum_splits', 'k_dconv', 'k_halo_plan', 'k_img_conv3', 'k_img_wgrad', 'k_rows_wgrad1',
                'k_lin_small', 'k_lin_wgrad_small', 'k_expand_bf16'),
    'occupancy': ('k_spconv', 'k_rowgemm', 'k_wgrad_reduce', 'k_sum_splits', 'k_dconv', 'k_img_conv3', 'k_img_wgrad', 'k_rows_wgrad1', 'k_lin_small',
                  'k_lin_wgrad_small', 'k_expand_bf16'),     # round 5: + the dense-volume engine
    'grounding': ('k_attn_',),
}
SCATTER = ('k_voxel_keys', 'k_insert_min', 'k_unique', 'k_morton', 'k_rs_', 'k_apply_sorted', 'k_stride_keys', 'k_kernel_map',
           'k_inverse_map', 'k_union', 'k_point_sample_fwd', 'k_ps_link', 'k_ps_gather', 'k_depth_to_points')


def table(path):
    rows = {}
    lines = open(path).read().splitlines()
    names = [c.strip() for c in lines[1].split('|')]
    for l in lines[2:]:
        c = [x.strip() for x in l.split('|')]
        rows[c[0]] = dict(zip(names[1:], [float(v) for v in c[1:]]))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tag', default='r3')
    ap.add_argument('--config', default='mv3ddet', choices=sorted(FAMILY))
    ap.add_argument('--steps', type=int, default=0, help='train steps the profiled command executed (timed + warm-up + extra); 0: the number of optimiser launches (k_adamw*) in the table')
    ap.add_argument('--fetch', required=True)
    ap.add_argument('--write', required=True)
    ap.add_argument('--command', default='python bench.py --no-cpu-baseline --no-other-configs --steps 4 --warmup 2')
    a = ap.parse_args()
    f, w = table(a.fetch), table(a.write)
    if a.steps <= 0:
        a.steps = max(1, int(sum(v['calls'] for k, v in f.items() if k.startswith('k_adamw'))))
    pick = lambda fam: [k for k in f if any(p in k for p in fam)]
    fam = pick(FAMILY[a.config])
    fetch = sum(f[k]['FETCH_SIZE'] for k in fam) * 1024 * 2
    write = sum(w[k]['WRITE_SIZE'] for k in fam if k in w) * 1024
    launches = int(sum(f[k]['calls'] for k in fam))
    out = dict(source=f'rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum / --pmc WRITE_SIZE TCC_MISS_sum (two separate passes) of `{a.command}` '
                      f'({a.steps} steps); tables: {os.path.relpath(a.fetch, ROOT)}, {os.path.relpath(a.write, ROOT)}; FETCH_SIZE (KB) doubled per '
                      'the gfx950 note of MI355X_MICROARCH.md, WRITE_SIZE (KB) as reported',
               config=a.config, family=sorted(fam), steps=a.steps, launches=launches, fetch_bytes=fetch, write_bytes=write,
               bytes_per_step=int((fetch + write) / a.steps), bytes_per_launch=int((fetch + write) / max(launches, 1)))
    if a.config == 'occupancy':                  # the dense-volume kernels alone: the neck (bench.py's roofline family)
        dn = pick(('k_dconv',))
        db = sum(f[k]['FETCH_SIZE'] for k in dn) * 1024 * 2 + sum(w[k]['WRITE_SIZE'] for k in dn if k in w) * 1024
        dl = int(sum(f[k]['calls'] for k in dn))
        out['dense'] = dict(kernels=sorted(dn), launches=dl, bytes_per_step=int(db / a.steps), bytes_per_launch=int(db / max(dl, 1)))
    if a.config == 'mv3ddet':
        sc = pick(SCATTER)
        sb = sum(f[k]['FETCH_SIZE'] for k in sc) * 1024 * 2 + sum(w[k]['WRITE_SIZE'] for k in sc if k in w) * 1024
        ms = sum(f[k]['dur_ms'] for k in sc)
        out['scatter_path'] = dict(kernels=sorted(sc), hbm_bytes_per_step=int(sb / a.steps), kernel_ms_per_step=round(ms / a.steps, 3),
                                   achieved_GBps=round(sb / (ms * 1e-3) / 1e9, 1) if ms else 0.0, peak_GBps=8000.0,
                                   frac=round(sb / (ms * 1e-3) / 1e9 / 8000.0, 4) if ms else 0.0,
                                   note='HBM bytes from the counters (FETCH_SIZE x 2 + WRITE_SIZE) of the A1-A4 / A6 / A8 kernels divided by their '
                                        'kernel time in the FETCH pass (counter collection serialises the kernels)')
    name = f'{a.tag}_pmc_traffic.json' if a.config == 'mv3ddet' else f'{a.tag}_pmc_traffic_{a.config}.json'
    json.dump(out, open(os.path.join(ROOT, 'profiles', name), 'w'), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k not in ('source', 'family')}))


if __name__ == '__main__':
    main()
