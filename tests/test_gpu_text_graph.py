"""The grounder's frozen text encoder (transformers RobertaModel, eval, no grad) as a captured graph on a side stream (text.TextGraph,
round 6): same bits as the eager call for fresh inputs of the captured shape; the detector's start_text / finish_text pair gives the
features of encode_text on the calling stream, whichever of ES_TEXT_ASYNC / ES_TEXT_GRAPH is switched off."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_graph_replay_equals_eager_call():
    from embodiedscan_amd.text import TextGraph, build_text_encoder
    dev = torch.device('cuda:0')
    enc = build_text_encoder(dict(num_hidden_layers=2), seed=3).to(dev)
    side = torch.cuda.Stream()
    g = torch.Generator().manual_seed(1)
    B, T = 5, 17
    mask = torch.ones((B, T), dtype=torch.long)
    for b in range(B):
        mask[b, T - b:] = 0
    mask = mask.to(dev)
    tg = TextGraph(enc, B, T, dev, side)
    for _ in range(3):
        ids = torch.where(mask.bool(), torch.randint(3, 50000, (B, T), generator=g).to(dev), torch.ones((B, T), dtype=torch.long, device=dev))
        with torch.cuda.stream(side):
            out = tg.run(ids, mask).clone()
        side.synchronize()
        with torch.no_grad():
            ref = enc(input_ids=ids, attention_mask=mask).last_hidden_state
        torch.cuda.synchronize()
        assert torch.equal(out, ref), float((out - ref).abs().max())


def test_detector_text_paths_agree():
    import embodiedscan_amd.models.detectors.sparse_featfusion_grounder as G
    from embodiedscan_amd import engine as E
    from embodiedscan_amd.config import build_detector, load_config
    from embodiedscan_amd.synth import make_grounding_sample, make_scan
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = torch.device('cuda:0')
    cfg = load_config(os.path.join(root, 'configs', 'mv_grounding.py'))
    det = build_detector(cfg, device=dev, seed=0).to(dev)
    det._bind()

    class S:                                     # the slice of a data sample encode_text touches
        pass
    samples = []
    for i in range(3):
        sc = make_scan(50 + i, n_views=2, augment=True, render_device=str(dev))
        a = make_grounding_sample(sc, seed=i)
        s = S()
        s.text, s.tokens_positive = a['text'], a['tokens_positive']
        s.gt_instances_3d = S()
        samples.append(s)
    outs = {}
    prev = (G.TEXT_ASYNC[0], G.TEXT_GRAPH[0], E.TAPE.enabled)
    E.TAPE.enabled = False
    try:
        for name, a_, g_ in (('graph', True, True), ('async eager', True, False), ('inline', False, False)):
            G.TEXT_ASYNC[0], G.TEXT_GRAPH[0] = a_, g_
            det._text_stream = None
            for rep in range(2):                 # (second pass: the captured graph is replayed)
                text, mask, tlen, T = det.encode_text(samples)
            torch.cuda.synchronize()
            outs[name] = (text.d.clone(), mask.clone(), tlen.clone(), T)
    finally:
        G.TEXT_ASYNC[0], G.TEXT_GRAPH[0], E.TAPE.enabled = prev
        det._text_stream = None
    for name in ('async eager', 'inline'):
        assert torch.equal(outs['graph'][0], outs[name][0]) and torch.equal(outs['graph'][1], outs[name][1]) and torch.equal(outs['graph'][2], outs[name][2]), name
    assert getattr(det, '_text_graphs', {}) and all(det._text_graphs.values()), 'the graph path was not taken'
