"""ctypes binding of libes_hip.so (the C-ABI drop-in boundary, include/es_hip.h).

The prototypes are parsed from the header so the header stays the single source of truth.
There is no fallback: if the shared library is missing or a symbol is absent this module
raises at import, and every call checks the returned status.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libes_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'es_hip.h')

_CT = {'int': ctypes.c_int, 'float': ctypes.c_float, 'double': ctypes.c_double, 'size_t': ctypes.c_size_t}


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes], [argnames])} for every prototype in es_hip.h."""
    src = open(path).read()
    src = re.sub(r'/\*.*?\*/', ' ', src, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(int|size_t)\s+(es_\w+)\s*\(([^)]*)\)\s*;', src):
        ret, name, args = m.group(1), m.group(2), m.group(3)
        at, an = [], []
        for a in args.split(','):
            a = a.strip()
            if not a or a == 'void':
                continue
            an.append(re.findall(r'(\w+)\s*$', a)[0])
            if '*' in a:
                at.append(ctypes.c_void_p)
            else:
                toks = [t for t in a.split() if t != 'const']
                at.append(_CT[toks[0]])
        out[name] = (_CT[ret], at, an)
    return out


PROTOS = parse_header()
CONSTS = {k: int(v) for k, v in re.findall(r'#define\s+(ES_\w+)\s+(\d+)', open(HEADER_PATH).read())}

if not os.path.exists(LIB_PATH):
    raise ImportError(f'{LIB_PATH} is missing: build it with `make -C embodiedscan_amd/csrc` '
                      '(or __graft_entry__.build()); there is no CPU fallback for the HIP path')
_lib = ctypes.CDLL(LIB_PATH)
_fn = {}
for _name, (_ret, _at, _an) in PROTOS.items():
    _f = getattr(_lib, _name)          # AttributeError if the .so does not export a declared symbol
    _f.restype, _f.argtypes = _ret, _at
    _fn[_name] = _f


if os.environ.get('ES_PINGPONG') is not None:          # A/B switch of the ping-pong LDS conv kernels (default: on)
    _fn['es_set_option'](1, int(os.environ['ES_PINGPONG']))
if os.environ.get('ES_WGRAD_HUGE') is not None:        # A/B switch of the 256x256 weight-gradient tile (default: on)
    _fn['es_set_option'](2, int(os.environ['ES_WGRAD_HUGE']))
if os.environ.get('ES_ROWGEMM') is not None:           # A/B switch of the K = 1 streaming row-GEMM kernel (default: on)
    _fn['es_set_option'](3, int(os.environ['ES_ROWGEMM']))
# tuning sweeps without a rebuild (tools/gpu_sweep.sh): weight-gradient slice targets, workspace cap, forward tap-split threshold
for _key, _env in ((4, 'ES_WG_BIG_TARGET'), (5, 'ES_WG_BIG_ROWS'), (6, 'ES_WG_SMALL_TARGET'), (7, 'ES_WG_CAP_MB'), (8, 'ES_FWD_SPLIT_WGS'),
                  (10, 'ES_DMA'), (11, 'ES_DMA_MIN_CIN'), (12, 'ES_RG128_MIN_CIN'), (14, 'ES_WGRAD_TR'), (15, 'ES_NORM_CB_ROWS'),
                  (16, 'ES_SPLIT_FOLD'), (19, 'ES_RG128_MIN_WGS'), (20, 'ES_WSHARE'), (21, 'ES_NARROW'), (23, 'ES_RG320'), (24, 'ES_LIN_SMALL'), (25, 'ES_EXPAND'), (26, 'ES_EXPAND_WGS')):
    if os.environ.get(_env) is not None:
        _fn['es_set_option'](_key, int(os.environ[_env]))


class HipError(RuntimeError):
    pass


PROFILE = None     # bench.py installs {'names': set, 'records': list, 'event': callable} to time chosen kernels
PAIRS = {}         # kernel-map device pointer -> device scalar with its number of valid (output, tap) pairs


def register_map(nbr):
    """while profiling, remember how many valid pairs a kernel map holds (algorithmic flops of its launches)."""
    if PROFILE is not None and nbr is not None:
        PAIRS[nbr.data_ptr()] = (nbr >= 0).sum()
    return nbr


_EXT_STREAMS = {}


def _stream_of(handle):
    """torch view of a raw hipStream_t (events must be recorded on the stream the kernel is launched on)"""
    import torch
    st = _EXT_STREAMS.get(handle)
    if st is None:
        st = _EXT_STREAMS[handle] = torch.cuda.ExternalStream(handle) if handle else torch.cuda.default_stream()
    return st


def call(name, *args):
    prof = PROFILE
    if prof is not None and name in prof['names']:
        e0, e1 = prof['event'](), prof['event']()
        st = _stream_of(args[-1])                 # every entry point takes its hipStream_t last
        e0.record(st)
        rc = _fn[name](*args)
        e1.record(st)
        prof['records'].append((name, e0, e1, args))          # bench.py resolves PAIRS[map pointer] right after
    else:
        rc = _fn[name](*args)
    if rc != 0:
        raise HipError(f'{name} failed with status {rc}')


def raw(name):
    return _fn[name]


_STREAM = [None]
_STREAM_OBJ = [None]


def stream():
    """current HIP stream handle (cached: torch.cuda.current_stream() costs ~7 us per call, and a train step makes
    ~500 launches); call refresh_stream() after switching streams."""
    if _STREAM[0] is None:
        refresh_stream()
    return _STREAM[0]


def refresh_stream():
    import torch
    _STREAM_OBJ[0] = torch.cuda.current_stream()
    _STREAM[0] = _STREAM_OBJ[0].cuda_stream
    return _STREAM[0]


def stream_obj():
    """the torch.cuda.Stream behind stream() (cached the same way)"""
    if _STREAM_OBJ[0] is None:
        refresh_stream()
    return _STREAM_OBJ[0]


def P(t):
    """device/host pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def iarr(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def parr(ptrs):
    """host array of device pointers"""
    return (ctypes.c_void_p * len(ptrs))(*[int(v) for v in ptrs])


def farr(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])
