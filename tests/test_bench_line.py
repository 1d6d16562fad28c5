"""The line bench.py prints last must survive the driver's capture (a 21.8 KB line left BENCH_r05.parsed = null).
Canned inputs: the full result objects of round 5 (profiles/r5_bench_*.json = what bench.py used to print)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CANNED = ['r5_bench_default.json', 'r5_bench_default_box2.json', 'r5_bench_2ranks_gloo_one_gpu.json']


def _load(name):
    p = os.path.join(ROOT, 'profiles', name)
    if not os.path.exists(p):
        pytest.skip(f'{name} not in profiles/')
    txt = open(p).read().strip().splitlines()[-1]
    return json.loads(txt)


@pytest.mark.parametrize('name', CANNED)
def test_compact_line_is_short_and_complete(name):
    full = _load(name)
    if full['n_gpus'] == 1:
        assert len(json.dumps(full)) > 8000                   # the canned object is the long form
    line = bench.compact_line(full, detail='bench_detail.json')
    txt = json.dumps(line)
    assert len(txt) < bench.LINE_LIMIT, len(txt)
    back = json.loads(txt)
    assert back == line
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
              'dtype', 'data', 'config', 'roofline', 'ranks', 'dist_backend'):
        assert k in back, k
    assert back['value'] == full['value'] and back['ms_per_step'] == full['ms_per_step']
    assert 'workload' in back['config'] and 'model' not in back['config']
    r = back['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert k in r, k
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-3
    if full['n_gpus'] == 1:
        cb = back['cpu_baseline']
        for k in ('value', 'unit', 'cores', 'kind', 'sample'):
            assert k in cb, k
        assert back['parity']['ok'] is True and back['parity']['max_rel_err'] < back['parity']['tol']
        for kind, leg in back['other_configs'].items():
            assert set(leg) <= {'value', 'ms_per_step', 'roofline_frac', 'parity_ok', 'vs_synthetic', 'error'}, (kind, leg)
            assert leg['value'] == full['other_configs'][kind]['value']


def test_compact_line_sheds_optional_keys_rather_than_grow():
    full = _load(CANNED[0])
    full = dict(full, scatter_path=dict(full.get('scatter_path') or {}, achieved_GBps=1.0),
                other_configs={f'leg{i}': dict(value=1.0, ms_per_step=2.0, roofline=dict(frac=0.5), parity=dict(ok=True)) for i in range(200)})
    line = bench.compact_line(full, detail='x' * 100)
    assert len(json.dumps(line)) < bench.LINE_LIMIT
    for k in ('metric', 'value', 'roofline', 'cpu_baseline', 'parity'):
        assert k in line


def test_emit_prints_the_short_line_last(tmp_path, capsys, monkeypatch):
    full = _load(CANNED[0])
    monkeypatch.setattr(bench, 'ROOT', str(tmp_path))
    os.makedirs(tmp_path / 'gpurun_out')
    line = bench.emit(full)
    out = capsys.readouterr().out.strip().splitlines()
    assert json.loads(out[-1]) == line and len(out[-1]) < bench.LINE_LIMIT
    for p in (tmp_path / 'bench_detail.json', tmp_path / 'gpurun_out' / 'bench_detail.json'):
        assert json.load(open(p))['value'] == full['value']            # the long form is on disk, complete
    assert line['detail'] == 'bench_detail.json'
