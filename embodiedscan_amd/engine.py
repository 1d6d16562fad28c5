"""Operator layer of the MI355X path: thin Python drivers around the C-ABI kernels plus a
minimal reverse-mode tape (HIP launches in, HIP launches out -- no torch.autograd, no CPU path).

torch is used for device memory (caching allocator) and the current HIP stream only.
Every forward function registers one closure on the tape; `TAPE.backward()` replays them in
reverse.  Gradients w.r.t. parameters accumulate straight into the flat gradient arena.
"""
import os

import torch
from . import hip
from .hip import P, call, iarr


def _stream():
    return hip.stream()


_NEW_VARS = None            # list while graphed() captures a segment


class Var:
    """A row matrix (N, C) f32 on the device with an optional gradient (and lazily made bf16 shadows of both, the
    gather sources of the bf16 convolution kernels)."""
    __slots__ = ('d', 'g', 'rg', 'dh', 'gh', 'gate', 'gated', 'fresh')

    def __init__(self, d, rg=True):
        self.d, self.g, self.rg = d, None, rg
        self.dh = self.gh = None
        if _NEW_VARS is not None:
            _NEW_VARS.append(self)     # a segment is being captured: its Vars are reset before every replay (graphed())
        self.fresh = False      # True between conv() and the norm() that consumes its output (nobody else sees this Var)
        self.gate = None        # folded-BN scale of the fused conv+BN+ReLU that produced this Var (see conv_affine)
        self.gated = False      # True: .g already is the gradient w.r.t. the producer's (pre-BN) conv output

    def shadow(self):
        if self.dh is None:
            self.dh = self.d if self.d.dtype == torch.bfloat16 else _cast_rows(self.d)     # bf16 rows are their own shadow
        return self.dh

    def grad_shadow(self):
        if self.gh is None:
            self.gh = _cast_rows(self.g)
        return self.gh

    @property
    def shape(self):
        return self.d.shape


class Param:
    """View into the parameter / gradient arenas (+ lazily refreshed bf16 copies for the bf16 MFMA path)."""
    __slots__ = ('d', 'g', 'bf_n', 'bf_t', 'bf_step', 'owner')

    def __init__(self, d, g=None):
        self.d, self.g = d, g
        self.bf_n = self.bf_t = None
        self.bf_step = -1
        self.owner = _OWNER[0]
        if d.dim() == 3 and d.is_cuda:
            CONV_PARAMS.setdefault(d.device, []).append(self)

    def bf16(self):
        """(natural [K][Cin][Cout], transposed [K][Cout][Cin]) bf16 copies, re-made when the weights changed.
        All conv kernels registered for this device are refreshed together by ONE table-driven launch."""
        if self.bf_step != WEIGHT_VERSION[0]:
            dev = self.d.device
            tab = _CAST_TABLE.get(dev)
            if tab is None or self.bf_n is None or self not in tab['set']:
                tab = _build_cast_table(dev)
            call('es_cast_weights_table', P(tab['dev']), tab['n'], tab['tiles'], _stream())
            for p in tab['params']:
                p.bf_step = WEIGHT_VERSION[0]
        return self.bf_n, self.bf_t


def refresh_weight_copies():
    """bf16 mode: bring the bf16 copies of every registered conv kernel up to date now (one launch on the current
    stream) instead of lazily at the first convolution that needs them."""
    if PRECISION[0] != 'bf16':
        return
    dev = torch.device('cuda', torch.cuda.current_device())
    ps = CONV_PARAMS.get(dev)
    if ps:
        ps[-1].bf16()


# Registry of the conv kernels (3-D Params) living on each GPU, the input of the one-launch bf16 weight cast.  Entries are
# tagged with the owner that was binding when they were made (a detector passes id(self) to begin_bind()); re-binding or
# releasing an owner drops its old entries, so a process that builds several detectors neither leaks their arenas nor
# keeps re-casting stale weights, and the tables of different devices never mix pointers.
_GC_FROZEN = [False]
CONV_PARAMS = {}          # torch.device -> [Param]
_CAST_TABLE = {}          # torch.device -> dict(dev=table tensor, n, tiles, params, set)
_OWNER = [None]


def begin_bind(owner):
    """drop every conv Param `owner` registered earlier and tag the ones created from now on with it"""
    release(owner)
    _OWNER[0] = owner


def end_bind():
    _OWNER[0] = None


def release(owner):
    """forget what detector `owner` registered here; objects frozen out of the collector's sight by settle_gc (gc.freeze at the
    detector's 6th step: everything alive then, the dropped detector's graph included) are handed back to it"""
    if _GC_FROZEN[0]:
        import gc
        gc.unfreeze()
        _GC_FROZEN[0] = False
    for dev in list(CONV_PARAMS):
        keep = [p for p in CONV_PARAMS[dev] if p.owner != owner]
        if len(keep) != len(CONV_PARAMS[dev]):
            CONV_PARAMS[dev] = keep
            _CAST_TABLE.pop(dev, None)
        if not keep:
            CONV_PARAMS.pop(dev, None)


def _build_cast_table(dev):
    ps = [p for p in CONV_PARAMS.get(dev, ()) if p.d.dim() == 3]
    rows, tiles = [], 0
    for p in ps:
        K, a, b = p.d.shape
        if p.bf_n is None:
            p.bf_n = torch.empty((K, a, b), dtype=torch.bfloat16, device=dev)
            p.bf_t = torch.empty((K, b, a), dtype=torch.bfloat16, device=dev)
        rows.append([p.d.data_ptr(), p.bf_n.data_ptr(), p.bf_t.data_ptr(), K, a, b, tiles])
        tiles += K * ((a + 63) // 64) * ((b + 63) // 64)
    tab = _CAST_TABLE[dev] = dict(dev=torch.tensor(rows, dtype=torch.int64).to(dev), n=len(rows), tiles=tiles, params=ps,
                                  set=set(ps))
    return tab


PRECISION = ['f32']       # 'f32': exact-f32 MFMA everywhere; 'bf16': bf16 MFMA (f32 accumulate) for conv fwd / dgrad
SHADOW = [os.environ.get('ES_SHADOW', '1') == '1']   # gather from bf16 shadow copies of activations / gradients: half the
                          # gather bytes, no conversion instructions in the staging loop; on by default since round 2 (the GPU
                          # suite passes identically with it; ES_SHADOW=0 restores f32 gathers)
WGRAD_BF16 = [True]       # in bf16 mode also run the weight-gradient GEMMs on the bf16 matrix cores
ACT16 = [os.environ.get('ES_ACT16', '1') != '0']   # round 3: the image backbone stores its ACTIVATIONS in bf16 (bf16 mode only):
                          # the fused conv + frozen-BN (+ residual) + ReLU launches read and write bf16 rows, the data
                          # gradients stay f32; halves the bytes of the HBM-bound 1x1 convolutions, no shadow copies
WEIGHT_VERSION = [0]      # bumped by the optimiser: invalidates the bf16 weight copies


class Tape:
    def __init__(self):
        self.fns = []
        self.enabled = True
        self.epoch = 0                  # bumped whenever the tape is emptied (graphed(): one replay of a segment per epoch)

    def add(self, fn):
        if self.enabled:
            self.fns.append(fn)

    def backward(self):
        for fn in reversed(self.fns):
            fn()
        join_wgrad_streams()
        self.fns = []
        self.epoch += 1

    def clear(self):
        self.fns = []
        self.epoch += 1


TAPE = Tape()

# ------------------------------------------------------------------ captured launch sequences (hipGraph)
# The image backbone's forward is ~65 launches with static shapes, static addresses and no host round trip, but queuing them
# costs the host ~4 ms per step -- during which the point branch (whose kernels the image branch is meant to run under) has not
# even been issued (profiles/r3s_timeline.txt: main stream idle for the first 4 ms of every step).  graphed() captures such
# a sequence once into a hipGraph (torch.cuda.CUDAGraph: activations live in the graph's private pool) and replays it with ONE
# launch per step; the backward closures the sequence put on the tape are kept and re-registered at every replay (they run
# eagerly, on the same buffers), the gradient fields of the sequence's Vars are reset first.
GRAPHS = [os.environ.get('ES_GRAPHS', '1') != '0']
GRAPH_STATS = dict(captured=0, replayed=0, eager=0, failed=0)


class _Segment:
    __slots__ = ('graph', 'outs', 'fns', 'vars', 'calls', 'failed', 'epoch')

    def __init__(self):
        self.graph = self.outs = None
        self.fns, self.vars, self.calls, self.failed, self.epoch = [], [], 0, False, -1


def reset_graphs(cache):
    """forget captured sequences (kernel-selection options or folded constants changed)"""
    cache.clear()


def graphed(cache, key, fn):
    """outs = fn(), through a hipGraph from the third call with the same key on (1st call: eager -- lazily built maps and
    shadows; 2nd: captured, then replayed).  Eager whenever per-launch instrumentation is on, on the default stream (capture
    needs a non-default stream: the side stream of the two-stream schedule), or after a failed capture."""
    global _NEW_VARS
    instrumented = hip.PROFILE is not None or DEBUG_CONV is not None or DEBUG_GRADS is not None or DEBUG_OPS is not None
    if not GRAPHS[0] or instrumented or _NEW_VARS is not None or \
            torch.cuda.current_stream() == torch.cuda.default_stream():
        GRAPH_STATS['eager'] += 1
        return fn()
    seg = cache.get(key)
    if seg is None:
        if len(cache) > 8:
            cache.clear()
        seg = cache[key] = _Segment()
    seg.calls += 1
    # a replay hands out the SAME Vars and closures as the one before: two of them on one tape (the sequence run twice before a
    # backward pass) would double-count gradients, so the second run in a tape epoch is eager
    if seg.failed or seg.calls == 1 or (TAPE.enabled and seg.epoch == TAPE.epoch):
        GRAPH_STATS['eager'] += 1
        return fn()
    seg.epoch = TAPE.epoch
    if seg.graph is None:
        n0 = len(TAPE.fns)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        _NEW_VARS = []
        try:
            g.capture_begin(capture_error_mode='thread_local')
            try:
                outs = fn()
            finally:
                g.capture_end()
        except Exception as e:                                  # leave the eager path in charge
            _NEW_VARS = None
            seg.failed = True
            GRAPH_STATS['failed'] += 1
            del TAPE.fns[n0:]
            import warnings
            warnings.warn(f'hipGraph capture of {key} failed ({e!r}): running this sequence eagerly')
            torch.cuda.synchronize()
            return fn()
        seg.vars, _NEW_VARS = _NEW_VARS, None
        seg.graph, seg.outs = g, outs
        seg.fns = TAPE.fns[n0:]
        del TAPE.fns[n0:]
        GRAPH_STATS['captured'] += 1
    for v in seg.vars:
        v.g = v.gh = None
        v.gated = False
    seg.graph.replay()
    GRAPH_STATS['replayed'] += 1
    if TAPE.enabled:
        TAPE.fns.extend(seg.fns)
    return seg.outs


# ------------------------------------------------------------------ side stream
# Independent branches of the step (image branch vs. point branch, samples of the loss) are issued on two HIP streams so
# that under-filled launches overlap.  Discipline that keeps the caching allocator safe without record_stream():
# (1) the side stream always starts ordered after everything queued on the main stream (fork event), (2) the main stream
# joins before it touches anything the side stream produced, (3) tensors crossing streams stay referenced until the
# join, so a block is only recycled by its own stream after the other stream's readers were ordered before it.
TWO_STREAMS = [os.environ.get('ES_TWO_STREAMS', '1') != '0']
_SIDE = {}


class side_stream:
    """with side_stream(): ... -- run the enclosed launches on the side stream.  fork=True (default): ordered after
    everything queued on the current stream so far; fork=False: just continue on the side stream."""

    def __init__(self, fork=True):
        self.fork = fork

    def __enter__(self):
        if not _SIDE:
            _SIDE.update(s=torch.cuda.Stream(), fork=torch.cuda.Event(), join=torch.cuda.Event())
        if self.fork:
            _SIDE['fork'].record(hip.stream_obj())
            _SIDE['s'].wait_event(_SIDE['fork'])
        self.ctx = torch.cuda.stream(_SIDE['s'])
        self.ctx.__enter__()
        hip.refresh_stream()
        return self

    def __exit__(self, *exc):
        _SIDE['join'].record(_SIDE['s'])
        r = self.ctx.__exit__(*exc)
        hip.refresh_stream()
        return r


def join_side():
    """the current (main) stream waits for everything issued on the side stream so far"""
    if _SIDE:
        hip.stream_obj().wait_event(_SIDE['join'])


# Weight gradients hang off the backward chain (nothing downstream reads them before the optimiser), so they are
# launched on a second stream paired with the compute stream: the chain of data-gradient launches does not wait for them
# and under-filled launches of both kinds overlap.
WGRAD_ASYNC = [os.environ.get('ES_WGRAD_ASYNC', '1') != '0']
_WGRAD_STREAMS = {}     # compute-stream handle -> dict(s=torch Stream, h=handle, fork=Event, join=Event, used=bool)
_KEEP = []              # temporaries read by queued weight-gradient launches; released by the final join


def _wgrad_stream(*keep):
    """handle of the weight-gradient stream of the current compute stream, made to wait for everything queued on the
    compute stream so far; `keep`: tensors the launch reads that nobody else references."""
    if not WGRAD_ASYNC[0]:
        return _stream()
    h = _stream()
    ws = _WGRAD_STREAMS.get(h)
    if ws is None:
        st = torch.cuda.Stream()
        ws = _WGRAD_STREAMS[h] = dict(s=st, h=st.cuda_stream, fork=torch.cuda.Event(), join=torch.cuda.Event(), used=False)
    ws['fork'].record(hip.stream_obj())
    ws['s'].wait_event(ws['fork'])
    ws['used'] = True
    _KEEP.extend(keep)
    return ws['h']


_WGRAD_WS = {}          # stream handle -> persistent workspace of the deterministic weight-gradient row split
# The first weight-gradient launch a weight receives in a step OVERWRITES its slot of the gradient arena (a plain store: the
# read-modify-write of a 1 GB gradient made the wide occupancy layers wait on 128 dependent loads per thread); later launches of
# the same step (shared modules, the 8 taps of a transposed convolution have their own slots) accumulate.  Keyed by the slot's
# device pointer, so aliases of one tensor share the stamp.  new_grad_epoch() is called where the arena is zeroed.
GRAD_EPOCH = [1]
_GRAD_STAMP = {}
WGRAD_OVERWRITE = [os.environ.get('ES_WGRAD_OVERWRITE', '1') != '0']


def new_grad_epoch():
    GRAD_EPOCH[0] += 1
    if len(_GRAD_STAMP) > 1 << 16:
        _GRAD_STAMP.clear()


def _first_write(ptr):
    """0 (overwrite) for the first gradient written to `ptr` in this epoch, 1 (accumulate) afterwards"""
    if not WGRAD_OVERWRITE[0]:
        return 1
    acc = 1 if _GRAD_STAMP.get(ptr) == GRAD_EPOCH[0] else 0
    _GRAD_STAMP[ptr] = GRAD_EPOCH[0]
    return acc


def _wgrad(name, sw, dW, *args):
    """launch a weight-gradient entry point (`args` = everything between the function name and dW) on stream `sw` with the
    workspace its row split asks for.  One buffer per stream, grown on demand and reused by every launch of that stream:
    the launches of a stream (partial tiles -> fixed-order reduction into dW) are ordered, so the reuse is race free."""
    if name == 'es_spconv_wgrad_bf16_src':
        X, xh, ldx, dY, yh, ldy, nbr, n_out, n_in, K, cin, cout = args
        need = hip.raw('es_spconv_wgrad_workspace_floats')(1, X, xh, ldx, dY, yh, ldy, n_out, n_in, K, cin, cout)
    else:
        X, ldx, dY, ldy, nbr, n_out, n_in, K, cin, cout = args
        need = hip.raw('es_spconv_wgrad_workspace_floats')(int(name == 'es_spconv_wgrad_bf16'), X, 0, ldx, dY, 0, ldy, n_out, n_in,
                                                           K, cin, cout)
    ws = None
    if need:
        ws = _WGRAD_WS.get(sw)
        if ws is None or ws.numel() < need:
            if ws is not None:
                _KEEP.append(ws)                 # launches already queued on `sw` may still use the old buffer
            ws = _WGRAD_WS[sw] = torch.empty(max(int(need), 1 << 22), dtype=torch.float32, device=torch.device('cuda', torch.cuda.current_device()))
    call(name, *args, dW, _first_write(dW), P(ws), ws.numel() if ws is not None else 0, sw)


IMG_WGRAD = [os.environ.get('ES_IMG_WGRAD', '1') != '0']     # round 6: 3x3 image weight gradients on csrc/imgwgrad.hip (A/B switch)
ROWS_WGRAD = [os.environ.get('ES_ROWS_WGRAD', '1') != '0']   # round 6: 1x1 weight gradients on contiguous rows (streaming kernel of csrc/imgwgrad.hip)
IMG_CONV = [os.environ.get('ES_IMG_CONV', '1') != '0']       # round 6: 3x3 image forward / gated data gradient on csrc/imgconv.hip


def _img_wgrad_floats(img, K, cin, cout, x, gy):
    """workspace floats of the image weight-gradient kernel for this launch, 0 when it does not take it: `img` = (n_img, H, W, stride)
    of a 3x3 / pad 1 convolution on image rows ((H, W) = its INPUT grid); bf16 activation rows, f32 gradient rows, C -> C channels
    (32 / 64), stride 1 or 2"""
    if not (IMG_WGRAD[0] and img is not None and K == 9 and cin == cout and x.dh is not None and gy.dtype == torch.float32):
        return 0
    return int(hip.raw('es_img_wgrad9_workspace_floats')(img[0], img[1], img[2], cin, img[3]))


def _img_wgrad(sw, dW, xh, ldx, gy, ldy, img, C, need):
    ws = _WGRAD_WS.get(sw)
    if ws is None or ws.numel() < need:
        if ws is not None:
            _KEEP.append(ws)                     # launches already queued on `sw` may still use the old buffer
        ws = _WGRAD_WS[sw] = torch.empty(max(int(need), 1 << 22), dtype=torch.float32, device=gy.device)
    call('es_img_wgrad9_bf16', P(xh), ldx, P(gy), ldy, img[0], img[1], img[2], C, img[3], dW, _first_write(dW), P(ws), ws.numel(), sw)


def wgrad_stream_obj():
    """the torch Stream that carries the weight-gradient launches of the current compute stream (created on first use) -- idle during
    the forward pass, so small independent forward-time work (the grounder's frozen text encoder) can ride on it instead of opening
    one more stream: a process has four hardware queues, and a fifth, sixth ... stream ends up sharing one with a stream it was meant to
    overlap (bench.py copy_stream: ~ 10 - 18 % per step when that happens)"""
    h = _stream()
    ws = _WGRAD_STREAMS.get(h)
    if ws is None:
        st = torch.cuda.Stream()
        ws = _WGRAD_STREAMS[h] = dict(s=st, h=st.cuda_stream, fork=torch.cuda.Event(), join=torch.cuda.Event(), used=False)
    return ws['s']


def join_wgrad_streams(final=True):
    """make the current stream wait for every queued weight-gradient launch (before the gradients are reduced or
    consumed by the optimiser)"""
    cur = hip.stream_obj()
    for ws in _WGRAD_STREAMS.values():
        if ws['used']:
            ws['join'].record(ws['s'])
            cur.wait_event(ws['join'])
            ws['used'] = False
    if final:
        _KEEP.clear()


_TICKET_WS = {}         # stream handle -> zero-initialised f32 workspace of the last-block reductions queued on that stream


def ticket_ws(floats, like, tag=''):
    """(tensor, floats) workspace for a kernel that reduces per-workgroup partials in its last workgroup (csrc/common.h
    es_last_block): its head holds a ticket counter that must be 0 before every launch and is left 0 by every launch, so ONE
    zero-initialised buffer per stream serves all such launches of the stream (they are ordered); grown on demand.  `tag`: a
    kernel family that leaves more than the ticket behind (es_topk_mask_ws keeps its selection state there) gets a buffer of its own."""
    h = (_stream(), tag)
    ws = _TICKET_WS.get(h)
    if ws is None or ws.numel() < floats or ws.device != like.device:
        if ws is not None:
            _KEEP.append(ws)                     # launches already queued may still use the old buffer
        ws = _TICKET_WS[h] = torch.zeros(max(int(floats), 1 << 16), dtype=torch.float32, device=like.device)
    return ws, ws.numel()


# Python's cyclic collector and the train loop (round 5, profiles/r5d / r5f_bench_grounding_diag.json: `gc_collections`): a generation-2
# collection walks every live container of the process -- model, configs, the frozen text encoder's module tree, torch's own tables
# -- and costs 110-170 ms of host time whenever the allocation counters trip it (every ~20-25 steps: the 3.5x step-time outliers of
# rounds 3-4).  A step leaves no cyclic garbage (sparse.CoordSet caches are weak), so after a few steps everything alive is long-lived:
# it is moved to the permanent generation once (gc.freeze) and later collections only look at what was allocated since (measured on
# this image: a full collection of 10^6 tracked objects 180 ms -> 0.0 ms after the freeze).  ES_GC_FREEZE=0 turns this off; a process
# that drops a detector and builds another should gc.unfreeze() + gc.collect() in between (bench.py release_frozen).
GC_SETTLE_STEP = 6


_HOST_SETTLED = [False]


def settle_host_threads(force=False):
    """Once per process (first train step): cap torch's intra-op (OpenMP) pool.  torch sizes it by the CPUs it SEES (128 threads on
    the GPU boxes); the cgroup GRANTS 16 cores there, and after any CPU-side tensor operation of a step the idle workers spin before
    they sleep -- together they exhaust the quota and the kernel throttles the whole process, the thread that queues the GPU's work
    included: every ~9th occupancy step was 12 - 27 ms longer with the device idle (cpu.stat nr_throttled 28 -> 59 over two 60-step runs,
    none with one thread: profiles/r5w_*).  A training step has no CPU tensor work worth a pool: a quarter of the granted cores per local
    rank (ES_HOST_THREADS=n overrides, 0 leaves torch alone)."""
    if _HOST_SETTLED[0] and not force:
        return
    _HOST_SETTLED[0] = True
    want = os.environ.get('ES_HOST_THREADS', 'auto')
    if want == '0':
        return
    if want == 'auto':
        from .datasets.loader import effective_cpus
        local = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1') or 1))
        n = max(1, effective_cpus() // 4 // local)
    else:
        n = max(1, int(want))
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)


def settle_gc(det):
    """called at the top of every train_step: freezes the collector's generations when `det` reaches its GC_SETTLE_STEP-th step
    (and, at the process's first step, caps the host thread pool: settle_host_threads)"""
    settle_host_threads()
    n = getattr(det, '_steps_done', 0) + 1
    det._steps_done = n
    if n == GC_SETTLE_STEP and os.environ.get('ES_GC_FREEZE', '1') != '0':
        import gc
        gc.collect()
        gc.freeze()
        _GC_FROZEN[0] = True                     # (release() unfreezes)


def reset_tickets():
    """re-zero the ticket heads of every cached election workspace (on the current stream).  A launch that faulted inside an
    in-launch reduction leaves its ticket non-zero; the step that saw the fault has already raised HipError, a caller that
    catches it and continues must call this (after synchronising) before the next step."""
    for ws in _TICKET_WS.values():
        ws[:min(ws.numel(), 2048)].zero_()


def drop_caches():
    """forget every per-stream workspace / kept temporary (tests that switch between devices or libraries in one process)"""
    _TICKET_WS.clear()
    _WGRAD_WS.clear()
    _DC_WS.clear()
    _KEEP.clear()


MARKS = None            # bench.py sets a list: stage boundaries as (name, event) recorded on the current stream


HOST_MARKS = None       # tools/host_profile.py sets a list: (name, host perf_counter) at the same stage boundaries


def mark(name):
    """stage boundary for the per-stage timing of bench.py (SURVEY 8d "Reporting"); free when MARKS is None"""
    if HOST_MARKS is not None:
        import time
        HOST_MARKS.append((name, time.perf_counter()))
    if MARKS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(hip.stream_obj())
        MARKS.append((name, ev))


DEBUG_GRADS = None      # tools/debug_grads.py sets a dict: id(Var) -> snapshot of its gradient when consumed
DEBUG_CONV = None       # tests/test_gpu_insitu.py sets a list: one record per convolution backward of the step (operands as the
                        # launches saw them + the data gradient this launch produced), for the in-situ check of every
                        # weight- / data-gradient launch against its arithmetic specification
DEBUG_OPS = None        # ... and a list for the other backward launches with parameters or matrix-core arithmetic: attention
                        # (kind 'attn'), LayerNorm ('ln'), ContrastiveEmbed ('contrastive', appended by GroundingHead)


def _cast_rows(t):
    n, C = t.shape
    h = torch.empty((n, C), dtype=torch.bfloat16, device=t.device)
    call('es_cast_rows_bf16', P(t), t.stride(0), n, C, P(h), _stream())
    return h


GEN_FUSED = [os.environ.get('ES_GEN_FUSED', '1') != '0']          # generative transposed conv: 8 taps in one launch per direction (round 4:
                                                                  # run on hardware, forward bit-identical, step -0.6 ms; profiles/r4a_*)
NORM_SHADOW = [os.environ.get('ES_NORM_SHADOW', '1') != '0']     # norm apply passes write the bf16 shadows of their outputs
WGRAD_SHADOW = [os.environ.get('ES_WGRAD_SHADOW', '1') != '0']   # weight-gradient launches gather from the bf16 shadows too
DET_SPLIT = [os.environ.get('ES_DET_SPLIT', '1') != '0']   # deterministic tap split (workspace + fixed-order reduction)


def _split_ws(n_out, K, cin, cout, like):
    """workspace for the deterministic tap split of an under-filled bf16 conv launch (size from the library's own rule,
    es_spconv_split_workspace_floats): (tensor or None, floats).  Its head holds the tile tickets of the in-kernel reduction (zero
    before and after every launch), so it is the stream's persistent zero-initialised ticket workspace (ticket_ws), not a fresh
    allocation: the split launches of a stream are ordered, each has read its partial tiles before the next one starts."""
    if not DET_SPLIT[0] or K <= 1 or n_out <= 0:
        return None, 0
    nf = int(hip.raw('es_spconv_split_workspace_floats')(n_out, K, cin, cout))
    if nf == 0:
        return None, 0
    return ticket_ws(nf, like, tag='split')


def _fwd_bf16(X, x_is_bf16, ldx, Wp, nbr, n_out, n_in, K, cin, cout, bias_p, Y, ldy, acc, like):
    """es_spconv_fwd_bf16 with the deterministic-split workspace when the launch would split its tap list"""
    ws, nf = _split_ws(n_out, K, cin, cout, like)
    if ws is not None:
        call('es_spconv_fwd_bf16_ws', X, x_is_bf16, ldx, Wp, nbr, n_out, n_in, K, cin, cout, bias_p, Y, ldy, acc, P(ws), nf, _stream())
    else:
        call('es_spconv_fwd_bf16', X, x_is_bf16, ldx, Wp, nbr, n_out, n_in, K, cin, cout, bias_p, Y, ldy, acc, _stream())


def _use_shadow(n_rows, C, K, cin, cout):
    """bf16 shadow pays when the rows are gathered several times (K > 1) and the fast kernel takes the shape"""
    return K > 1 and hip.raw('es_spconv_bf16_is_fast')(n_rows, C, K, cin, cout) == 1


# ------------------------------------------------------------------ halo-tile K = 27 convolutions (round 6, csrc/halo.hip)
HALO = [os.environ.get('ES_HALO', '1') != '0']
if os.environ.get('ES_HALO_MIN_WGS') is not None:
    hip.raw('es_halo_set_option')(30, int(os.environ['ES_HALO_MIN_WGS']))


def _halo_ok(nbr, n_out, n_in, ldx, K, cin, cout):
    return (HALO[0] and K == 27 and nbr is not None
            and hip.raw('es_spconv_halo_supported')(n_out, n_in, ldx, K, cin, cout) == 1)


def halo_plan(nbr):
    """(loc, hrows, hcnt) of a kernel map: per 256-row tile the sorted distinct source rows and the 16-bit positions of every
    (row, tap) neighbour in that list (es_halo_plan).  Built on the current stream at first use and kept ON the map tensor, so it
    lives and dies with the map's cache entry (sparse.CoordSet.kernel_map / inverse_map)."""
    plan = getattr(nbr, '_halo', None)
    if plan is None:
        n_out, K = nbr.shape
        rows = int(hip.raw('es_halo_plan_rows')(n_out))
        loc = torch.empty((rows, K), dtype=torch.int16, device=nbr.device)
        hrows = torch.empty((rows // 256, 256 * K), dtype=torch.int32, device=nbr.device)
        hcnt = torch.empty((rows // 256,), dtype=torch.int32, device=nbr.device)
        call('es_halo_plan', P(nbr), n_out, K, P(loc), P(hrows), P(hcnt), _stream())
        plan = nbr._halo = (loc, hrows, hcnt)
        if hip.PROFILE is not None:
            hip.PAIRS[loc.data_ptr()] = (nbr >= 0).sum()
    return plan


def _halo_launch(Xh, ldx, Wp, nbr, n_out, n_in, K, cin, cout, bias_p, Y, ldy, acc):
    """nbr: the launch's gather map.  The INVERSE map of a stride-1 convolution on one coordinate set is its forward map with the
    taps mirrored (sparse.CoordSet.inverse_map marks it `_mirror_of`): such a data gradient runs on the forward map's plan."""
    fwd = getattr(nbr, '_mirror_of', None)
    loc, hrows, hcnt = halo_plan(fwd if fwd is not None else nbr)
    call('es_spconv_halo_bf16', Xh, ldx, Wp, P(loc), P(hrows), P(hcnt), n_out, n_in, K, cin, cout, bias_p, Y, ldy, acc,
         int(fwd is not None), _stream())


def empty(shape, like, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=like.device)


def zeros(shape, like, dtype=torch.float32):
    return torch.zeros(shape, dtype=dtype, device=like.device)


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, 'row matrices must be contiguous along channels'
    return t.stride(0)


def _grad_target(v, shape_like):
    """(tensor, accumulate flag) for writing the gradient of v."""
    v.gh = None                                       # whatever shadow of the gradient exists is about to go stale
    if v.g is None:
        v.g = torch.empty(shape_like.shape, dtype=torch.float32, device=shape_like.device)    # gradients are f32 rows
        return v.g, 0
    return v.g, 1



# ------------------------------------------------------------------ dense-volume convolutions (round 5, csrc/dconv.hip)
# nn.Conv3d on a dense (B, X, Y, Z) grid by address arithmetic: no neighbour map.  `dense` = (B, X, Y, Z, ksize, stride, pad) of the
# INPUT grid; a convolution that passes it runs on the dense engine wherever the library takes the shape (bf16 mode, reduction
# channels % 64, output channels % 256; the data gradient for stride 1) and falls back to the map kernels elsewhere -- `maps` is then
# a callable returning (nbr, inv), so the maps of a layer the dense engine covers completely are never built.
# Z = 0 names a FLAT grid: nn.Conv2d on (B, X, Y) images (the FPN's 3x3 output convolutions); a bias is pre-filled, the launch accumulates.
DENSE = [os.environ.get('ES_DENSE', '1') != '0']
_DC_WS = {}             # stream handle -> persistent workspace of the partial tiles of split dense launches


def _dense_geom(dense):
    return iarr(dense) if dense is not None else None


def dense_ok(dense, mode, cin, cout):
    if not (DENSE[0] and dense is not None and PRECISION[0] == 'bf16'):
        return False
    if mode == 2 and dense[4] == 1 and (cin // 256) * (cout // 256) < 48:
        # weight gradient of a 1x1x1 convolution: one workgroup per 256 x 256 channel tile walking every row -- the neck's 768 -> 1536
        # down-sample would launch 18 of them (134 us against the map kernel's row-split 49 us, profiles/r5p_occ_launches.jsonl)
        return False
    # the library's own 32-bit index limits (rows x leading dimension, taps x channels^2: es_dconv_* return -4 beyond them) are part of
    # the answer, so that such a shape falls back to the map kernels instead of raising; callers check contiguity
    B, X, Y, Z, ks = dense[:5]
    rows = B * X * Y * max(Z, 1) * (8 if mode >= 3 else 1)
    taps = ks ** (2 if Z == 0 else 3)
    if rows * max(cin, cout) >= (1 << 31) or taps * cin * cout >= (1 << 31):
        return False
    return hip.raw('es_dconv_supported')(_dense_geom(dense), mode, cin, cout) == 1


def _dense_ws(s, need, like):
    """the partial-tile workspace of split dense launches on stream `s` (None when the launch needs none)"""
    if not need:
        return None
    ws = _DC_WS.get(s)
    if ws is None or ws.numel() < need or ws.device != like.device:
        if ws is not None:
            _KEEP.append(ws)                     # launches already queued on this stream may still use the old buffer
        ws = _DC_WS[s] = torch.empty(max(need, 1 << 22), dtype=torch.float32, device=like.device)
    return ws


def _dense_launch(Xh, ldx, Wp, dense, mode, cin, cout, Y, ldy, acc, like):
    g = _dense_geom(dense)
    s = _stream()
    ws = _dense_ws(s, int(hip.raw('es_dconv_workspace_floats')(g, mode, cin, cout)), like)
    call('es_dconv_fwd_bf16', Xh, ldx, Wp, g, mode, cin, cout, Y, ldy, acc, P(ws), ws.numel() if ws is not None else 0, s)


def _dense_wgrad(xh, ldx, gh, ldy, dense, transposed, cin, cout, dW, acc, sw, like):
    """weight gradient on the dense engine, on the weight-gradient stream `sw`; launches of few tiles over many rows (the 2-D 3x3
    layers) slice the rows through that stream's workspace"""
    g = _dense_geom(dense)
    ws = _dense_ws(sw, int(hip.raw('es_dconv_wgrad_workspace_floats')(g, transposed, cin, cout)), like)
    call('es_dconv_wgrad_ws_bf16', xh, ldx, gh, ldy, g, transposed, cin, cout, dW, acc, P(ws), ws.numel() if ws is not None else 0, sw)


class _Maps:
    """(nbr, inv) of a convolution, built on first use"""
    __slots__ = ('fn', 'val')

    def __init__(self, fn):
        self.fn, self.val = fn, None

    def get(self):
        if self.val is None:
            self.val = self.fn()
        return self.val


# ------------------------------------------------------------------ convolution
def conv(x, w, nbr, inv, n_out, bias=None, need_dx=True, bias_from=0, dense=None, maps=None):
    """y[j] = sum_k x[nbr[j,k]] @ w[k] (+bias).  w.d: (K,Cin,Cout).  nbr None -> identity (K=1).
    bias_from: first output column whose bias is trainable (earlier columns keep a zero bias).
    dense / maps: see "dense-volume convolutions" above (nbr / inv are then taken from maps() where a map kernel runs)."""
    K, cin, cout = w.d.shape
    n_in = x.d.shape[0]
    y = Var(empty((n_out, cout), x.d))
    bf = PRECISION[0] == 'bf16' and cin >= 16
    if maps is not None:
        maps = _Maps(maps) if not isinstance(maps, _Maps) else maps
    x16 = x.d.dtype == torch.bfloat16                  # bf16 activation rows (the image backbone's outputs under ACT16): their own shadow
    if x16:
        assert bf and _ld(x.d) == cin, 'bf16 input rows need the bf16 kernels and contiguous rows'
        x.dh = x.d
    dn = dense if (bf and _ld(x.d) == cin and dense_ok(dense, 0, cin, cout)) else None
    if dn is None and maps is not None:
        nbr, inv = maps.get()
    if dn is not None:
        if bias:                                  # rows pre-filled with the bias, the launch accumulates (the tile's accumulators fill the
            y.d.copy_(bias.d.expand_as(y.d))      # register file: an epilogue that also held the bias spilled)
        _dense_launch(P(x.shadow()), cin, P(w.bf16()[1]), dn, 0, cin, cout, P(y.d), cout, 1 if bias else 0, x.d)
    elif bf and (x16 or (SHADOW[0] and _ld(x.d) == cin and _use_shadow(n_in, cin, K, cin, cout))):
        if _halo_ok(nbr, n_out, n_in, cin, K, cin, cout):
            _halo_launch(P(x.shadow()), cin, P(w.bf16()[1]), nbr, n_out, n_in, K, cin, cout, P(bias.d) if bias else 0, P(y.d), cout, 0)
        else:
            _fwd_bf16(P(x.shadow()), 1, cin, P(w.bf16()[1]), P(nbr), n_out, n_in, K, cin, cout, P(bias.d) if bias else 0, P(y.d),
                      cout, 0, x.d)
    elif bf:
        _fwd_bf16(P(x.d), 0, _ld(x.d), P(w.bf16()[1]), P(nbr), n_out, n_in, K, cin, cout, P(bias.d) if bias else 0, P(y.d),
                  cout, 0, x.d)
    else:
        call('es_spconv_fwd', P(x.d), _ld(x.d), P(w.d), P(nbr), n_out, n_in, K, cin, cout, P(bias.d) if bias else 0,
             P(y.d), cout, 0, 0, _stream())

    def bwd():
        if y.g is None:
            return
        if DEBUG_GRADS is not None:
            DEBUG_GRADS[id(y)] = y.g.clone()
        _conv_backward(x, w, nbr, inv, n_out, y, y.g, bias, bias_from, need_dx, bf, dense=dense, maps=maps)
    TAPE.add(bwd)
    x.fresh = False
    y.fresh = True
    return y


def _conv_backward(x, w, nbr, inv, n_out, y, gy, bias, bias_from, need_dx, bf, gate=None, dense=None, maps=None, img=None):
    """wgrad (+ bias grad) and dgrad of a convolution whose output gradient is the row matrix `gy`.
    gate: folded-BN scale of x's producer -- the dgrad launch then also applies that layer's ReLU mask and BN scale
    (x is its only consumer), leaving x.g as the gradient of the producer's raw conv output."""
    K, cin, cout = w.d.shape
    n_in = x.d.shape[0]
    s = _stream()
    rec = None
    # dense engine: weight gradient / data gradient by address arithmetic where the library takes the shape
    dn_ok = bf and dense is not None and gate is None and _ld(gy) == cout and _ld(x.d) == cin and WGRAD_BF16[0]
    dn_w = dense if (dn_ok and w.g is not None and dense_ok(dense, 2, cin, cout)) else None
    dn_d = dense if (dn_ok and need_dx and x.rg and dense_ok(dense, 1, cin, cout)) else None
    if maps is not None and (DEBUG_CONV is not None or (w.g is not None and dn_w is None) or (need_dx and x.rg and dn_d is None)):
        nbr, inv = maps.get()
    if DEBUG_CONV is not None:
        rec = dict(x=x.d, w=w, nbr=nbr, n_out=n_out, gy=gy.clone(), bf=bool(bf), gate=gate, need_dx=bool(need_dx and x.rg),
                   before=(x.g.clone() if (x.g is not None and need_dx and x.rg) else None), bias=bias, bias_from=bias_from)
        DEBUG_CONV.append(rec)
    # bf16 shadow of the output gradient: gather source of the data-gradient launch and, with the input's shadow from
    # the forward pass, of the weight-gradient launch (made here, before the weight-gradient stream forks)
    gh = None
    if dn_w is not None or dn_d is not None or (
            bf and cout >= 16 and SHADOW[0] and need_dx and x.rg and gate is None and _ld(gy) == cout
            and _use_shadow(n_out, cout, K, cout, cin)):
        gh = y.grad_shadow() if gy is y.g else _cast_rows(gy)
    if dn_w is not None:
        x.shadow()                               # (made by the forward launch; never on the weight-gradient stream)
    if w.g is not None or (bias is not None and bias.g is not None):
        sw = _wgrad_stream(gy, x.d, gh, x.dh)
    iw = _img_wgrad_floats(img, K, cin, cout, x, gy) if (w.g is not None and bf and dn_w is None) else 0
    # 1x1 layers on contiguous rows (identity map), bf16 activation rows, f32 gradient rows: streaming kernel (csrc/imgwgrad.hip)
    rw = 0
    if (w.g is not None and bf and dn_w is None and not iw and ROWS_WGRAD[0] and K == 1 and nbr is None and n_in == n_out
            and x.dh is not None and gy.dtype == torch.float32):
        rw = int(hip.raw('es_rows_wgrad1_workspace_floats')(n_out, cin, cout))
    if dn_w is not None:
        _dense_wgrad(P(x.dh), cin, P(gh), cout, dn_w, 0, cin, cout, P(w.g), _first_write(P(w.g)), sw, gh)
    elif iw:
        _img_wgrad(sw, P(w.g), x.dh, _ld(x.dh), gy, _ld(gy), img, cin, iw)
    elif rw:
        ws = _WGRAD_WS.get(sw)
        if ws is None or ws.numel() < rw:
            if ws is not None:
                _KEEP.append(ws)
            ws = _WGRAD_WS[sw] = torch.empty(max(rw, 1 << 22), dtype=torch.float32, device=gy.device)
        call('es_rows_wgrad1_bf16', P(x.dh), _ld(x.dh), P(gy), _ld(gy), n_out, cin, cout, P(w.g), _first_write(P(w.g)), P(ws), ws.numel(), sw)
    elif w.g is not None and bf and WGRAD_BF16[0] and ((SHADOW[0] and WGRAD_SHADOW[0] and (gh is not None or x.dh is not None))
                                                  or x.d.dtype == torch.bfloat16):      # (bf16 activation rows ARE their shadow: ES_SHADOW=0 must not strand them)
        xs, ys = x.dh if x.dh is not None else x.d, gh if gh is not None else gy
        _wgrad('es_spconv_wgrad_bf16_src', sw, P(w.g), P(xs), int(x.dh is not None), _ld(xs), P(ys), int(gh is not None), _ld(ys),
               P(nbr), n_out, n_in, K, cin, cout)
    elif w.g is not None:
        assert x.d.dtype == torch.float32, 'bf16 activation rows reach the weight gradient through their shadow (x.dh)'
        _wgrad('es_spconv_wgrad_bf16' if (bf and WGRAD_BF16[0]) else 'es_spconv_wgrad', sw, P(w.g), P(x.d), _ld(x.d), P(gy),
               _ld(gy), P(nbr), n_out, n_in, K, cin, cout)
    if bias is not None and bias.g is not None:
        # column sums of gy (round 4: one deterministic launch on the weight-gradient stream; was an f32 GEMM against a column of ones)
        nb, dst = cout - bias_from, bias.g.data_ptr() + 4 * bias_from
        key = (sw, 'colsum')
        ws = _TICKET_WS.get(key)
        need = int(hip.raw('es_colsum_workspace_floats')(n_out, nb))
        if ws is None or ws.numel() < need or ws.device != gy.device:      # (a cached buffer of another device is never handed out)
            if ws is not None:
                _KEEP.append(ws)
            ws = _TICKET_WS[key] = torch.zeros(max(need, 1 << 16), dtype=torch.float32, device=gy.device)
            if sw != _stream():                  # the zero fill was queued on the compute stream, AFTER the weight-gradient stream forked:
                ev = torch.cuda.Event()          # order the fill before the first launch that reads the ticket
                ev.record(hip.stream_obj())
                hip._stream_of(sw).wait_event(ev)
        call('es_colsum', gy.data_ptr() + 4 * bias_from, _ld(gy), n_out, nb, dst, _first_write(dst), P(ws), ws.numel(), sw)
    if need_dx and x.rg and gate is not None:
        assert x.g is None and bf, 'gated dgrad: x must have exactly one consumer'
        x.g, x.gated = torch.empty(x.d.shape, dtype=torch.float32, device=x.d.device), True
        if (IMG_CONV[0] and img is not None and K == 9 and img[3] == 1 and cin == cout and x.d.dtype == torch.bfloat16
                and gy.dtype == torch.float32 and _ld(x.d) == cin
                and hip.raw('es_img_conv3_supported')(img[0], img[1], img[2], cin, 1, 1) == 1):
            # gated data gradient of a 3x3 image layer by address arithmetic (csrc/imgconv.hip, mode 1)
            call('es_img_conv3_bf16', P(gy), _ld(gy), P(w.bf16()[0]), img[0], img[1], img[2], cin, 1, 1, P(gate), 0, P(x.d), _ld(x.d), 3,
                 P(x.g), 0, _ld(x.g), s)
        elif x.d.dtype == torch.bfloat16:        # the gate operand is the bf16 activation (only its sign is read)
            call('es_spconv_fwd_bf16_io', P(gy), 0, _ld(gy), P(w.bf16()[0]), P(inv), n_in, n_out, K, cout, cin, P(gate), 0,
                 P(x.d), 1, _ld(x.d), 3, P(x.g), 0, _ld(x.g), s)
        else:
            call('es_spconv_fwd_bf16_affine', P(gy), _ld(gy), P(w.bf16()[0]), P(inv), n_in, n_out, K, cout, cin, P(gate), 0,
                 P(x.d), _ld(x.d), 3, P(x.g), _ld(x.g), s)
    elif need_dx and x.rg:
        g, acc = _grad_target(x, x.d)
        if dn_d is not None:
            _dense_launch(P(gh), cout, P(w.bf16()[0]), dn_d, 1, cin, cout, P(g), _ld(g), acc, gh)
        elif gh is not None and _halo_ok(inv, n_in, n_out, cout, K, cout, cin):
            _halo_launch(P(gh), cout, P(w.bf16()[0]), inv, n_in, n_out, K, cout, cin, 0, P(g), _ld(g), acc)
        elif gh is not None:
            _fwd_bf16(P(gh), 1, cout, P(w.bf16()[0]), P(inv), n_in, n_out, K, cout, cin, 0, P(g), _ld(g), acc, gh)
        elif bf and cout >= 16:
            _fwd_bf16(P(gy), 0, _ld(gy), P(w.bf16()[0]), P(inv), n_in, n_out, K, cout, cin, 0, P(g), _ld(g), acc, gy)
        else:
            call('es_spconv_fwd', P(gy), _ld(gy), P(w.d), P(inv), n_in, n_out, K, cout, cin, 0, P(g), _ld(g), 1, acc, s)
    if rec is not None and rec['need_dx']:
        rec['after'] = x.g.clone()               # (gradient buffer after this launch; `before` = what it accumulated onto)


def conv_affine(x, w, nbr, inv, n_out, scale, shift, act=1, res=None, need_dx=True, sole_consumer=False, out_bf16=False, img=None):
    """conv -> frozen-BN affine (+ residual) (+ ReLU) of the 2-D backbone.  In bf16 mode this is ONE launch (affine
    fused into the conv epilogue); in f32 mode conv() followed by affine_act().
    sole_consumer: promise that x feeds nothing but this conv; if x itself came out of a fused conv+BN+ReLU, its
    ReLU/BN backward is then folded into this conv's data-gradient launch.
    out_bf16: store the output rows in bf16 (ACT16); x / res may themselves be bf16 row matrices.
    img: (n_img, H, W, stride) when the map is a 3x3 / pad 1 image-grid map: the weight gradient may then run by address arithmetic
    (csrc/imgwgrad.hip) instead of through the map."""
    K, cin, cout = w.d.shape
    n_in = x.d.shape[0]
    if PRECISION[0] != 'bf16':
        return affine_act(conv(x, w, nbr, inv, n_out, need_dx=need_dx), scale, shift, act=act, res=res)
    h16 = torch.bfloat16
    y16 = bool(out_bf16 and ACT16[0] and cout % 4 == 0)
    x16, r16 = x.d.dtype == h16, (res is not None and res.d.dtype == h16)
    y = Var(empty((n_out, cout), x.d, dtype=h16 if y16 else torch.float32))
    # round 6: 3x3 layers on an image grid by address arithmetic (csrc/imgconv.hip) where the library takes the shape
    ic = (IMG_CONV[0] and img is not None and K == 9 and x16 and res is None and act in (0, 1) and cin == cout and _ld(x.d) == cin
          and hip.raw('es_img_conv3_supported')(img[0], img[1], img[2], cin, img[3], 0) == 1)
    if ic:
        call('es_img_conv3_bf16', P(x.d), cin, P(w.bf16()[1]), img[0], img[1], img[2], cin, img[3], 0, P(scale), P(shift), 0, 0, act,
             P(y.d), int(y16), cout, _stream())
    elif x16 or r16 or y16:
        call('es_spconv_fwd_bf16_io', P(x.d), int(x16), _ld(x.d), P(w.bf16()[1]), P(nbr), n_out, n_in, K, cin, cout, P(scale),
             P(shift), P(res.d) if res is not None else 0, int(r16), _ld(res.d) if res is not None else 0, act, P(y.d),
             int(y16), cout, _stream())
    else:
        call('es_spconv_fwd_bf16_affine', P(x.d), _ld(x.d), P(w.bf16()[1]), P(nbr), n_out, n_in, K, cin, cout, P(scale),
             P(shift), P(res.d) if res is not None else 0, _ld(res.d) if res is not None else 0, act, P(y.d), cout, _stream())
    if y16:
        y.dh = y.d                              # bf16 rows are their own gather shadow
    if x16:
        x.dh = x.d

    if act == 1 and res is None:
        y.gate = scale
    gate = x.gate if sole_consumer else None

    def bwd():
        if y.g is None:
            return
        if DEBUG_GRADS is not None:
            DEBUG_GRADS[id(y)] = y.g.clone()
        if y.gated:                             # the consumer's dgrad launch already applied mask and scale
            gconv = y.g
        else:
            gr = accr = 0
            if res is not None and res.rg:
                t, accr = _grad_target(res, res.d)
                gr = P(t)
            gconv = torch.empty((n_out, cout), dtype=torch.float32, device=y.d.device)   # gradient w.r.t. the conv output
            call('es_affine_act_bwd_yh' if y.d.dtype == h16 else 'es_affine_act_bwd', P(y.g), P(y.d), P(scale), n_out, cout, act,
                 P(gconv), 0, gr, accr, _stream())
        _conv_backward(x, w, nbr, inv, n_out, y, gconv, None, 0, need_dx, True, gate, img=img)
    TAPE.add(bwd)
    return y


def gen_conv_transpose(x, w):
    """MinkowskiGenerativeConvolutionTranspose(k=2,s=2): y[8i+k] = x[i] @ w[k]; 8 row GEMMs into the
    (N, 8*Cout) view of the output."""
    K, cin, cout = w.d.shape
    n = x.d.shape[0]
    y = Var(empty((n * 8, cout), x.d))
    s = _stream()
    bf = PRECISION[0] == 'bf16'
    fused = False
    if bf and GEN_FUSED[0] and x.d.dtype == torch.float32:      # all eight taps in one launch
        fused = hip.raw('es_gen_transpose_fwd_bf16')(P(x.d), _ld(x.d), P(w.bf16()[1]), n, cin, cout, P(y.d), s) == 0
    for k in range(0 if not fused else 8, 8):
        if bf:
            call('es_spconv_fwd_bf16', P(x.d), 0, _ld(x.d), w.bf16()[1].data_ptr() + 2 * k * cin * cout, 0, n, n, 1, cin,
                 cout, 0, y.d.data_ptr() + 4 * k * cout, 8 * cout, 0, s)
        else:
            call('es_spconv_fwd', P(x.d), _ld(x.d), w.d.data_ptr() + 4 * k * cin * cout, 0, n, n, 1, cin, cout, 0,
                 y.d.data_ptr() + 4 * k * cout, 8 * cout, 0, 0, s)

    def bwd():
        if y.g is None:
            return
        s = _stream()
        recs = None
        if DEBUG_CONV is not None:                     # the 8 taps as 8 identity-map K = 1 records sharing one data gradient
            before = x.g.clone() if (x.rg and x.g is not None) else None
            recs = [dict(x=x.d, w=ParamSlice(w, k), nbr=None, n_out=n, gy=y.g.view(n, 8, cout)[:, k].clone(), bf=bool(bf), gate=None,
                         need_dx=bool(x.rg), before=before, bias=None, bias_from=0, group=id(y), group_size=8) for k in range(8)]
            DEBUG_CONV.extend(recs)
        g, acc = _grad_target(x, x.d) if x.rg else (None, 0)
        sw = _wgrad_stream(y.g, x.d) if w.g is not None else s
        dfused = False
        if g is not None and bf and GEN_FUSED[0] and y.g.is_contiguous():      # (n * 8, cout) rows = (n, 8 cout)
            dfused = hip.raw('es_gen_transpose_dgrad_bf16')(P(y.g), P(w.bf16()[0]), n, cin, cout, P(g), _ld(g), acc, s) == 0
        for k in range(8):
            gy = y.g.data_ptr() + 4 * k * cout
            if w.g is not None:
                _wgrad('es_spconv_wgrad_bf16' if (bf and WGRAD_BF16[0]) else 'es_spconv_wgrad', sw,
                       w.g.data_ptr() + 4 * k * cin * cout, P(x.d), _ld(x.d), gy, 8 * cout, 0, n, n, 1, cin, cout)
            if dfused:
                continue
            if g is not None and bf:
                call('es_spconv_fwd_bf16', gy, 0, 8 * cout, w.bf16()[0].data_ptr() + 2 * k * cin * cout, 0, n, n, 1, cout,
                     cin, 0, P(g), _ld(g), 1 if (acc or k > 0) else 0, s)
            elif g is not None:
                call('es_spconv_fwd', gy, 8 * cout, w.d.data_ptr() + 4 * k * cin * cout, 0, n, n, 1, cout, cin, 0,
                     P(g), _ld(g), 1, 1 if (acc or k > 0) else 0, s)
        if recs is not None and x.rg:
            recs[-1]['after'] = x.g.clone()          # (after all eight taps)
    TAPE.add(bwd)
    return y


def conv_transpose_dense(x, w, dense):
    """nn.ConvTranspose3d(k=2, s=2) on a dense (B, X, Y, Z) grid (dense = (B, X, Y, Z, 2, 2, 0)): y (B*2X*2Y*2Z, Cout) in DENSE row order
    by the parity-class launch of the dense engine (csrc/dconv.hip mode 3) -- no generative-layout intermediate, no row permutation;
    backward: data gradient (mode 4: 8 taps gathered at 2 r + p) and weight gradient (mode 5).  Caller checks dense_ok(dense, 3 / 4 / 5)."""
    K, cin, cout = w.d.shape
    n = x.d.shape[0]
    y = Var(empty((n * 8, cout), x.d))
    _dense_launch(P(x.shadow()), cin, P(w.bf16()[1]), dense, 3, cin, cout, P(y.d), cout, 0, x.d)

    def bwd():
        if y.g is None:
            return
        gh = y.grad_shadow()
        rec_before = x.g.clone() if (DEBUG_CONV is not None and x.rg and x.g is not None) else None
        if w.g is not None:
            x.shadow()
            sw = _wgrad_stream(y.g, x.d, gh, x.dh)
            _dense_wgrad(P(x.dh), cin, P(gh), cout, dense, 1, cin, cout, P(w.g), _first_write(P(w.g)), sw, gh)
        if x.rg:
            g, acc = _grad_target(x, x.d)
            _dense_launch(P(gh), cout, P(w.bf16()[0]), dense, 4, cin, cout, P(g), _ld(g), acc, gh)
        if DEBUG_CONV is not None:                     # the 8 taps as 8 identity-map K = 1 records sharing one data gradient (as gen_conv_transpose)
            B, X, Y, Z = dense[:4]
            gv = y.g.view(B, X, 2, Y, 2, Z, 2, cout)
            recs = [dict(x=x.d, w=ParamSlice(w, k), nbr=None, n_out=n, gy=gv[:, :, k >> 2, :, (k >> 1) & 1, :, k & 1].reshape(n, cout).clone(),
                         bf=True, gate=None, need_dx=bool(x.rg), before=rec_before, bias=None, bias_from=0, group=id(y), group_size=8)
                    for k in range(8)]
            if x.rg:
                recs[-1]['after'] = x.g.clone()
            DEBUG_CONV.extend(recs)
    TAPE.add(bwd)
    return y


# ------------------------------------------------------------------ norms
def norm(x, weight, bias, seg_off, eps, act=0, res=None, running=None, momentum=0.1):
    """Train-mode batch norm (seg_off=[0,N]) / instance norm (seg_off=batch offsets) with fused
    residual add and activation (0 none, 1 ReLU, 2 ELU)."""
    n, C = x.d.shape
    nseg = len(seg_off) - 1
    so = iarr(seg_off)
    ws_n = hip.raw('es_norm_workspace_floats')(n, C, so, nseg) + 2 * nseg * C
    ws = empty((ws_n,), x.d)
    mean, invstd = empty((nseg, C), x.d), empty((nseg, C), x.d)
    y = Var(empty((n, C), x.d))
    rm, rv = (running if running is not None else (None, None))
    # bf16 mode: the apply passes also write the bf16 gather shadows their consumers would otherwise make with a cast launch
    # each (forward: of y, for the next convolution; backward: of the gradient handed to the producing convolution, when this
    # norm is that Var's only consumer) -- bit-identical to the casts they replace
    fuse = NORM_SHADOW[0] and PRECISION[0] == 'bf16' and SHADOW[0] and C >= 16 and C % 8 == 0
    private, x.fresh = bool(x.fresh), False
    if fuse:
        y.dh = empty((n, C), x.d, dtype=torch.bfloat16)
    call('es_norm_fwd', P(x.d), _ld(x.d), n, C, so, nseg, float(eps), P(weight.d), P(bias.d),
         P(res.d) if res is not None else 0, _ld(res.d) if res is not None else 0, act, P(rm), P(rv), float(momentum),
         P(mean), P(invstd), P(ws), P(y.d), C, P(y.dh) if fuse else 0, _stream())

    def bwd():
        if y.g is None:
            return
        s = _stream()
        if DEBUG_GRADS is not None:
            DEBUG_GRADS[id(y)] = y.g.clone()
        ws2 = empty((ws_n,), x.d)
        g, acc = _grad_target(x, x.d)
        gh = empty((n, C), x.d, dtype=torch.bfloat16) if (fuse and private and _ld(g) == C) else None
        call('es_norm_bwd', P(y.g), _ld(y.g), P(y.d), C, P(x.d), _ld(x.d), n, C, so, nseg, P(mean), P(invstd),
             P(weight.d), act, P(weight.g), P(bias.g), P(ws2), P(g), _ld(g), acc, P(gh), s)
        x.gh = gh
        if DEBUG_GRADS is not None:
            DEBUG_GRADS[('norm', id(y))] = dict(dx=g.clone(), dz=y.g.clone(), mean=mean.clone(), invstd=invstd.clone(),
                                                x=x.d.clone(), yd=y.d.clone(), acc=acc, n=n, C=C, act=act)
        if res is not None and res.rg:          # y.g now holds dz == gradient of the residual input
            if res.g is None:
                res.g = y.g
            else:
                call('es_axpy2d', P(res.g), _ld(res.g), P(y.g), _ld(y.g), n, C, 1.0, 1, s)
    TAPE.add(bwd)
    return y


def affine_act(x, scale, shift, act=1, res=None, need_dx=True):
    """frozen-BN affine (+residual) (+ReLU) for the 2-D backbone."""
    n, C = x.d.shape
    y = Var(empty((n, C), x.d))
    call('es_affine_act_fwd', P(x.d), P(scale), P(shift), P(res.d) if res is not None else 0, n, C, act, P(y.d),
         _stream())

    def bwd():
        if y.g is None:
            return
        gx = accx = gr = accr = 0
        if need_dx and x.rg:
            t, accx = _grad_target(x, x.d)
            gx = P(t)
        if res is not None and res.rg:
            t, accr = _grad_target(res, res.d)
            gr = P(t)
        if gx or gr:
            call('es_affine_act_bwd', P(y.g), P(y.d), P(scale), n, C, act, gx, accx, gr, accr, _stream())
    TAPE.add(bwd)
    return y


# ------------------------------------------------------------------ pooling / row moves
def maxpool(x, nbr, n_out, need_dx=True):
    C = x.d.shape[1]
    K = nbr.shape[1]
    y = Var(empty((n_out, C), x.d))
    arg = empty((n_out, C), x.d, torch.int32)
    call('es_maxpool_fwd', P(x.d), _ld(x.d), P(nbr), n_out, K, C, P(y.d), P(arg), _stream())

    def bwd():
        if y.g is None or not (need_dx and x.rg):
            return
        if x.g is None:
            x.g = torch.zeros_like(x.d)
        call('es_maxpool_bwd', P(y.g), P(arg), n_out, C, P(x.g), _ld(x.g), _stream())
    TAPE.add(bwd)
    return y


def gather_rows(x, idx):
    """y = x[idx] (MinkowskiPruning with idx = kept rows)."""
    n, C = idx.shape[0], x.d.shape[1]
    y = Var(empty((n, C), x.d))
    call('es_row_move', P(y.d), C, P(x.d), _ld(x.d), P(idx), n, C, 0, _stream())

    def bwd():
        if y.g is None or not x.rg:
            return
        if x.g is None:
            x.g = torch.zeros_like(x.d)
        call('es_row_move', P(x.g), _ld(x.g), P(y.g), _ld(y.g), P(idx), n, C, 1, _stream())
    TAPE.add(bwd)
    return y


def union_add(a, b, pos_a, pos_b, n):
    """sparse a + b on the coordinate union (rows pos_a / pos_b of the result)."""
    C = a.d.shape[1]
    y = Var(zeros((n, C), a.d))
    s = _stream()
    call('es_row_move', P(y.d), C, P(a.d), _ld(a.d), P(pos_a), a.d.shape[0], C, 2, s)
    call('es_row_move', P(y.d), C, P(b.d), _ld(b.d), P(pos_b), b.d.shape[0], C, 1, s)

    def bwd():
        if y.g is None:
            return
        s = _stream()
        for v, pos in ((a, pos_a), (b, pos_b)):
            if not v.rg:
                continue
            m = v.d.shape[0]
            if v.g is None:
                v.g = empty((m, C), a.d)
                call('es_row_move', P(v.g), C, P(y.g), _ld(y.g), P(pos), m, C, 0, s)
            else:
                t = empty((m, C), a.d)
                call('es_row_move', P(t), C, P(y.g), _ld(y.g), P(pos), m, C, 0, s)
                call('es_axpy2d', P(v.g), _ld(v.g), P(t), C, m, C, 1.0, 1, s)
    TAPE.add(bwd)
    return y


def copy_cols(dst, col, src):
    """dst[:, col:col+C] = src (raw tensors)."""
    n, C = src.shape
    call('es_axpy2d', dst.data_ptr() + 4 * col, dst.stride(0), P(src), src.stride(0), n, C, 1.0, 0, _stream())


def add_into(dst, src):
    n, C = src.shape
    call('es_axpy2d', P(dst), dst.stride(0), P(src), src.stride(0), n, C, 1.0, 1, _stream())


def add(a, b):
    """y = a + b of two row matrices of the same shape (dense residual / lateral sums)."""
    n, C = a.d.shape
    y = Var(empty((n, C), a.d))
    s = _stream()
    call('es_axpy2d', P(y.d), C, P(a.d), _ld(a.d), n, C, 1.0, 0, s)
    call('es_axpy2d', P(y.d), C, P(b.d), _ld(b.d), n, C, 1.0, 1, s)

    def bwd():
        if y.g is None:
            return
        for v in (a, b):
            if not v.rg:
                continue
            if v.g is None:
                v.g = y.g if v is a and not b.rg else y.g.clone()
            else:
                call('es_axpy2d', P(v.g), _ld(v.g), P(y.g), _ld(y.g), n, C, 1.0, 1, _stream())
    TAPE.add(bwd)
    return y


def upsample_add_(fine, coarse, n_img, Hf, Wf, Hc, Wc):
    """mmdet.FPN top-down step, in place: fine += nearest_upsample(coarse, size=(Hf, Wf)).  `fine` keeps its identity
    (its producer's backward sees the same gradient); the gradient reaching `coarse` is accumulated when the tape passes
    this point, i.e. after every later consumer of `fine` has contributed."""
    C = fine.d.shape[1]
    call('es_upsample_nearest_add_fwd', P(fine.d), P(coarse.d), n_img, Hf, Wf, Hc, Wc, C, _stream())

    def bwd():
        if fine.g is None or not coarse.rg:
            return
        g, acc = _grad_target(coarse, coarse.d)
        call('es_upsample_nearest_add_bwd', P(fine.g), P(g), n_img, Hf, Wf, Hc, Wc, C, acc, _stream())
    TAPE.add(bwd)
    return fine


# ------------------------------------------------------------------ transformer operators (grounding decoder, A19)
class ParamSlice:
    """tap j of a (K, Cin, Cout) Param as a (1, Cin, Cout) kernel of its own (the q / k / v blocks of an attention
    in-projection): shares storage, gradient and the parent's bf16 copies"""
    __slots__ = ('parent', 'j', 'd', 'g')

    def __init__(self, parent, j):
        self.parent, self.j = parent, j
        self.d = parent.d[j:j + 1]
        self.g = parent.g[j:j + 1] if parent.g is not None else None

    def bf16(self):
        n, t = self.parent.bf16()
        return n[self.j:self.j + 1], t[self.j:self.j + 1]


def linear(x, w, b=None, need_dx=True):
    """y = x @ W + b as a row GEMM on the convolution engine (w: Param / ParamSlice (1, Cin, Cout))"""
    return conv(x, w, None, None, x.d.shape[0], bias=b, need_dx=need_dx)


def relu_(x):
    """in-place ReLU on x.d; the gradient of x is masked in place when the tape passes here"""
    call('es_relu_fwd', P(x.d), x.d.numel(), _stream())

    def bwd():
        if x.g is not None:
            call('es_relu_bwd', P(x.g), P(x.d), x.d.numel(), _stream())
    TAPE.add(bwd)
    return x


def layernorm(x, w, b, res=None, eps=1e-5):
    """y = LayerNorm(x (+ res)) over channels; w, b: Params (C,)"""
    n, C = x.d.shape
    assert x.d.is_contiguous() and (res is None or res.d.is_contiguous())
    y = Var(empty((n, C), x.d))
    z = empty((n, C), x.d) if res is not None else x.d
    mean, rstd = empty((n,), x.d), empty((n,), x.d)
    call('es_layernorm_fwd', P(x.d), P(res.d) if res is not None else 0, n, C, P(w.d), P(b.d), float(eps), P(y.d),
         P(z) if res is not None else 0, P(mean), P(rstd), _stream())

    def bwd():
        if y.g is None:
            return
        s = _stream()
        tgt = [v for v in ((x, res) if res is not None else (x,)) if v.rg]
        dz = torch.empty_like(y.d)
        ws, nws = ticket_ws(hip.raw('es_layernorm_bwd_workspace_floats')(n, C), y.d)
        rec = None
        if DEBUG_OPS is not None:
            rec = dict(kind='ln', dy=y.g.clone(), z=z.clone(), w=w.d.clone(), mean=mean.clone(), rstd=rstd.clone(), n=n, C=C,
                       wp=w, bp=b, dw0=w.g.clone(), db0=b.g.clone())
            DEBUG_OPS.append(rec)
        call('es_layernorm_bwd', P(y.g), P(z), n, C, P(w.d), P(mean), P(rstd), P(dz), 0, P(w.g), P(b.g), P(ws), nws, s)
        if rec is not None:
            rec.update(dz=dz.clone(), dw1=w.g.clone(), db1=b.g.clone())
        owned = False                            # dz may be handed to exactly one input as its gradient buffer
        for v in tgt:
            if v.g is None:
                v.g = dz if not owned else dz.clone()
                owned = True
            else:
                call('es_axpy2d', P(v.g), _ld(v.g), P(dz), C, n, C, 1.0, 1, s)
    TAPE.add(bwd)
    return y


def attention(q, k, v, B, H, Lq, Lk, klen=None):
    """softmax(q k^T / sqrt(32)) v per (sample, head); q (B*Lq, H*32), k / v (B*Lk, H*32) projected row matrices (Vars,
    possibly column slices of wider buffers); klen: device int32 (B,) valid key counts or None"""
    bf = 1 if PRECISION[0] == 'bf16' else 0
    o = Var(empty((B * Lq, H * 32), q.d))
    lse = empty((B * H * Lq,), q.d)
    call('es_attn_fwd', P(q.d), _ld(q.d), P(k.d), _ld(k.d), P(v.d), _ld(v.d), B, H, Lq, Lk, P(klen), P(o.d), H * 32, P(lse), bf,
         _stream())

    def bwd():
        if o.g is None:
            return
        delta = empty((B * H * Lq,), q.d)
        gs = []
        for t in (q, k, v):
            if t.g is None:
                t.g = torch.empty_like(t.d)
                gs.append(0)
            else:
                gs.append(1)
        assert gs[0] == gs[1] == gs[2], 'attention inputs must be fresh projections (single consumer)'
        rec = None
        if DEBUG_OPS is not None:
            rec = dict(kind='attn', q=q.d.clone(), k=k.d.clone(), v=v.d.clone(), o=o.d.clone(), do=o.g.clone(), lse=lse.clone(),
                       klen=None if klen is None else klen.clone(), B=B, H=H, Lq=Lq, Lk=Lk, bf=bf, acc=gs[0],
                       before=[t.g.clone() for t in (q, k, v)] if gs[0] else None)
            DEBUG_OPS.append(rec)
        call('es_attn_bwd', P(q.d), _ld(q.d), P(k.d), _ld(k.d), P(v.d), _ld(v.d), P(o.d), H * 32, P(o.g), _ld(o.g), P(lse), B, H,
             Lq, Lk, P(klen), P(delta), P(q.g), _ld(q.g), P(k.g), _ld(k.g), P(v.g), _ld(v.g), gs[0], bf, _stream())
        if rec is not None:
            rec.update(delta=delta.clone(), dq=q.g.clone(), dk=k.g.clone(), dv=v.g.clone())
    TAPE.add(bwd)
    return o
