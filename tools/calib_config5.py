#!/usr/bin/env python
"""One-off calibration behind tests/test_gpu_config5.py: at BASELINE config-5 scale, how far are (a) the HIP exact-f32 path and
(b) the CPU f32 oracle from an f64 evaluation of the dense neck + head + loss on the SAME neck input, per weight-gradient
tensor?  (The neck input volume is captured from the f32 oracle run; only the neck, head and losses are re-run in f64.)
Prints one line per tensor.   python tools/calib_config5.py > profiles/r3_config5_f64_calibration.txt"""
import os
import sys
import time
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def main():
    from embodiedscan_amd import engine as E, pipeline
    from embodiedscan_amd.config import build_detector, load_config
    from embodiedscan_amd.synth import make_occ_gt, make_scan
    from oracle import model as OM, occ as OO
    dev = torch.device('cuda:0')
    cfg = load_config(os.path.join(ROOT, 'configs', 'mv_occ.py'))
    m = cfg['model']
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    scan = make_scan(5100, n_views=10, augment=False, render_device='cuda:0')
    occ = make_occ_gt(scan, seed=51)
    dscan = pipeline.upload_scan(scan, dev)
    sd = {k: v.cpu() for k, v in det.state_dict().items()}
    watch = [k for k in det.arena.grad_dict() if k.startswith(('neck_3d.', 'bbox_head.'))]
    E.PRECISION[0] = 'f32'
    batch = pipeline.make_occ_batch([dscan], [occ])
    points_host = [p.cpu() for p in batch['inputs']['points']]
    data = det.data_preprocessor(batch, True)
    det._bind()
    det.arena.grad.zero_()
    det.forward(data['inputs'], data['data_samples'], mode='loss')
    E.TAPE.backward()
    torch.cuda.synchronize()
    hip = {k: v.cpu() for k, v in det.arena.grad_dict().items() if k in watch}
    del det
    torch.cuda.empty_cache()
    imgs = OM.preprocess_img(torch.from_numpy(scan['img']), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375])[None]
    captured = {}
    orig = OO.imvoxel_neck

    def capture(x, sd_, **kw):
        captured['x'] = x.detach().clone()
        return orig(x, sd_, **kw)
    OO.imvoxel_neck = capture
    osd = {k: (v.clone().requires_grad_(True) if k in watch else v) for k, v in sd.items()}
    t0 = time.perf_counter()
    ol = OO.detector_loss(osd, points_host, imgs, [scan['meta']], [torch.from_numpy(occ['gt_occupancy'])],
                          [torch.from_numpy(occ['gt_occupancy_masks'])], m['n_voxels'], m['point_cloud_range'],
                          cfg['prior_generator']['ranges'][0], tuple(m['neck_3d']['n_blocks']))
    sum(ol.values()).backward()
    OO.imvoxel_neck = orig
    print(f'# f32 oracle forward+backward {time.perf_counter() - t0:.1f} s', flush=True)
    # f64: neck + head + loss on the captured neck input
    sd64 = {k: (v.double().clone().requires_grad_(True) if k in watch else v.double()) for k, v in sd.items()
            if k.startswith(('neck_3d.', 'bbox_head.'))}
    t0 = time.perf_counter()
    x3 = OO.imvoxel_neck(captured['x'].double(), sd64, n_blocks=tuple(m['neck_3d']['n_blocks']), training=True)
    preds = [torch.nn.functional.conv3d(l, sd64[f'bbox_head.occ.{i}.weight']) for i, l in enumerate(x3)]
    l64 = OO.head_loss(preds, [torch.from_numpy(occ['gt_occupancy'])], [torch.from_numpy(occ['gt_occupancy_masks'])])
    sum(l64.values()).backward()
    print(f'# f64 neck+head forward+backward {time.perf_counter() - t0:.1f} s', flush=True)
    print('# tensor | HIP exact-f32 vs f64 | CPU f32 oracle vs f64 | HIP vs CPU f32')
    for k in watch:
        if sd64[k].grad is None:
            continue
        print(f'{k} | {rel(hip[k], sd64[k].grad):.3e} | {rel(osd[k].grad, sd64[k].grad):.3e} | {rel(hip[k], osd[k].grad):.3e}', flush=True)


if __name__ == '__main__':
    main()
