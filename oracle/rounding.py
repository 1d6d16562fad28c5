"""bf16-operand mode of the CPU oracle.  TEST ORACLE (imported by tests/ and tools/ only).

The HIP path's bf16 mode rounds the two operands of every matrix product to bf16 (RNE) and accumulates in f32: forward
y = r(x) * r(w), data gradient dx = r(dy) * r(w)^T, weight gradient dw = r(x)^T * r(dy) (embodiedscan_amd/engine.py: conv
/ _conv_backward; layers with fewer than 16 input channels stay on the exact-f32 path).  With `MODE[0] = 'bf16'` the
oracle's products -- sparse / dense convolutions, Linear layers, the head's output GEMMs -- follow exactly that rule through
one autograd Function, so that a bf16 gradient of the HIP path can be compared with its OWN arithmetic specification at a
tolerance that only has to absorb the summation order instead of the bf16-vs-f32 gap (0.25 / 0.6 gates)."""
import torch

MODE = [None]          # None: exact f32 (the pinned oracle); 'bf16': operands rounded as described above
MIN_CIN = 16
ACT16 = [True]         # bf16 mode: the image backbone STORES its activations in bf16 (embodiedscan_amd.engine.ACT16); the backward
                       # treats the stored value as the activation (straight-through: no gradient of the rounding)


def r(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _Rounded(torch.autograd.Function):
    """y = fn(r(x), r(w)); backward: dx = d fn / dx applied to r(gy) with the weights r(w); dw = d fn / dw applied to r(gy)
    at the rounded input -- fn must be bilinear in (x, w) (every product of the model is)"""

    @staticmethod
    def forward(ctx, fn, round_dgrad, x, w):
        xr, wr = r(x.detach()), r(w.detach())
        ctx.fn, ctx.round_dgrad = fn, round_dgrad
        ctx.save_for_backward(xr, wr)
        with torch.no_grad():
            return fn(xr, wr)

    @staticmethod
    def backward(ctx, gy):
        xr, wr = ctx.saved_tensors
        gr = r(gy)
        dx = dw = None
        with torch.enable_grad():
            if ctx.needs_input_grad[2]:
                xl = xr.clone().requires_grad_(True)
                # the data-gradient launch falls back to exact f32 when the OUTPUT has fewer than 16 channels
                dx, = torch.autograd.grad(ctx.fn(xl, wr if ctx.round_dgrad else wr), xl, gr if ctx.round_dgrad else gy)
            if ctx.needs_input_grad[3]:
                wl = wr.clone().requires_grad_(True)
                dw, = torch.autograd.grad(ctx.fn(xr, wl), wl, gr)
        return None, None, dx, dw


def act(t):
    """activation-storage rounding of the image backbone in bf16 mode (value r(t), gradient of the identity)"""
    if MODE[0] != 'bf16' or not ACT16[0]:
        return t
    return t + (r(t.detach()) - t.detach())


def op(fn, x, w, cin, cout=None):
    """fn(x, w) under the current mode; cin / cout: channel counts that decide whether the HIP path uses the bf16 cores"""
    if MODE[0] != 'bf16' or cin < MIN_CIN:
        return fn(x, w)
    return _Rounded.apply(fn, cout is None or cout >= MIN_CIN, x, w)


class bf16_operands:
    """with bf16_operands(): ... -- run the oracle in bf16-operand mode; act16: the image backbone stores bf16 activations
    (mv-3ddet / grounder: yes; the occupancy detector, whose FPN takes f32 rows: no)"""

    def __init__(self, act16=True):
        self.act16 = act16

    def __enter__(self):
        import torch.nn.functional as F
        self.prev, MODE[0] = MODE[0], 'bf16'
        self.prev_act, ACT16[0] = ACT16[0], self.act16
        # Linear layers (reg branches, FFN, text_feat_map, attention in / out projections -- F.multi_head_attention_forward
        # resolves `linear` in torch.nn.functional at call time): E.linear runs them on the convolution engine
        self.linear = F.linear
        orig = self.linear

        def linear(x, w, b=None):
            y = op(lambda a, c: orig(a, c), x, w, w.shape[1], w.shape[0])
            return y if b is None else y + b
        F.linear = linear
        return self

    def __exit__(self, *exc):
        import torch.nn.functional as F
        F.linear = self.linear
        MODE[0] = self.prev
        ACT16[0] = self.prev_act
