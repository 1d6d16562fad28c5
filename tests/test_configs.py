"""Boundary (north_star: "configs/detection and configs/grounding run unchanged"): every `mv-*` configuration the
reference ships under configs/detection, configs/grounding and configs/occupancy is read UNCHANGED by
embodiedscan_amd.config.load_config and builds its detector through the registry (the `cont-*` configurations name detectors
SURVEY section 2 puts out of scope).  /root/reference does not exist on the GPU box, so the model sections are mirrored in
configs/*.py: this test (build container) pins the mirrors to the reference files, tests/test_gpu_configs.py (GPU box) trains
one step with each mirror."""
import glob
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference/configs'
MIRROR = {'detection/mv-det3d_8xb4_embodiedscan-3d-284class-9dof.py': 'mv_3ddet.py',
          'grounding/mv-grounding_8xb12_embodiedscan-vg-9dof.py': 'mv_grounding.py',
          'grounding/mv-grounding_8xb12_embodiedscan-vg-9dof-full.py': 'mv_grounding.py',
          'grounding/mv-grounding_8xb12_embodiedscan-vg-9dof_complex-all.py': 'mv_grounding.py',
          'grounding/mv-grounding_8xb12_embodiedscan-vg-9dof_fcaf-coder.py': 'mv_grounding_fcaf.py',
          'occupancy/mv-occ_8xb1_embodiedscan-occ-80class.py': 'mv_occ.py'}


def _norm(d):
    if isinstance(d, dict):
        return {k: _norm(v) for k, v in d.items()}
    if isinstance(d, (list, tuple)):
        return [_norm(v) for v in d]
    return d


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference checkout exists in the build container only')
def test_every_reference_mv_config_is_mirrored_and_builds():
    from embodiedscan_amd.config import build_detector, build_optim_wrapper, load_config
    shipped = sorted(os.path.relpath(p, REF) for sub in ('detection', 'grounding', 'occupancy')
                     for p in glob.glob(os.path.join(REF, sub, 'mv-*.py')))
    assert shipped == sorted(MIRROR), (shipped, sorted(MIRROR))           # a new reference config must get a mirror
    for rel in shipped:
        ref = load_config(os.path.join(REF, rel))
        loc = load_config(os.path.join(ROOT, 'configs', MIRROR[rel]))
        a, b = _norm(ref['model']), _norm(loc['model'])
        a['backbone'].pop('init_cfg', None)                               # torchvision://resnet50: no checkpoints offline
        assert a == b, f'{rel}: model section differs from configs/{MIRROR[rel]}: ' + \
            str([k for k in set(a) | set(b) if a.get(k) != b.get(k)])
        det = build_detector(ref, device='cpu')                           # the UNCHANGED reference file through the registry
        assert type(det).__name__ == ref['model']['type']
        opt = build_optim_wrapper(ref)
        assert opt.lr == ref['optim_wrapper']['optimizer']['lr']
        del det
