"""Thin tensor-level wrappers over the C-ABI (fgn_amd.lib).

PyTorch is plumbing here: it owns device memory and the HIP stream; every op validates
operand shapes on the host (a mis-shaped launch can fault the GPU), passes raw device
pointers plus the current stream to libfgn_hip.so, and raises on any error.  Tensors
are NHWC fp32 and must be contiguous.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import lib as _lib


# Live HIP-event timing of the convolution launches (bench.py roofline): when set to a ConvProfile, every conv
# launch appends a record dict(kind, kernel, e0, e1, flop_direct, flop_issued, n_img, n_img_dev, shape):
#   kind        'conv' (one fgn_conv2d launch, incl. its split-K reduce), 'wg_in' / 'wg_gemm' / 'wg_out' (the three
#               kernels of a Winograd layer, bracketed separately)
#   kernel      name of the device kernel as rocprofv3 reports it (from fgn_conv2d_kernel_id)
#   flop_*      per image: direct-convolution FLOPs of the layer / MFMA FLOPs actually issued (0 for the transforms)
PROFILE = None

# conv3 + shortcut conv of the first block of a stride-1 stage as one dual-operand K loop (conv1x1_dual).  A/B knob.
FUSED_SHORTCUT = os.environ.get('FGN_FUSED_SHORTCUT', '1') != '0'
# Arithmetic of the GEMM-shaped launches (1x1 / stride 1 convolutions, the fused conv3 + shortcut, the Winograd GEMMs):
# 'x3' = conv_pw_x3_kernel, every f32 product as six bf16 MFMA products of exact three-way splits, f32 accumulation
# (csrc/conv_pw_x3.h; the packers below then also build the weights' bf16-plane image); 'f32' = the f32-input MFMA kernels;
# 'h2' (default) = conv_pw_h2_kernel: three f16 MFMA products of two-way splits of the power-of-two scaled operands per f32
# product (csrc/conv_pw_h2.h; the packers then build the weights' f16-plane image instead), on the same launches.
GEMM_MATH = os.environ.get('FGN_GEMM_MATH', 'h2')
# by row tile (fgn_x3_row_tile); template arguments: waves along M, 32-row blocks per wave, terms, LDS stages, 16x16x32 MFMA
X3_KERNELS = {64: 'conv_pw_x3_kernel<2, 1, 6, 2, true>', 128: 'conv_pw_x3_kernel<2, 2, 6, 2, true>'}
# by tile (fgn_h2_row_tile: 64 / 128 rows x 128 columns, 264 = 128 rows x 64 columns); template arguments: waves along M,
# waves along N, 32-row blocks per wave, LDS stages, implicit-GEMM loader (3x3 / strided convolutions)
H2_KERNELS = {64: 'conv_pw_h2_kernel<2, 2, 1, 2, %s>', 128: 'conv_pw_h2_kernel<2, 2, 2, 2, %s>', 264: 'conv_pw_h2_kernel<4, 1, 1, 2, %s>'}


def x3_kernel(rows: int, cout: int, k: int, grp_rows: int = 0, grp_valid: int = 0) -> str:
    """Name (as rocprofv3 reports it) of the conv_pw_x3_kernel instance a GEMM of this shape is launched on."""
    return X3_KERNELS.get(_lib.load().fgn_x3_row_tile(rows, cout, k, grp_rows, grp_valid), 'conv_pw_x3_kernel<?>')


def h2_kernel(rows: int, cout: int, k: int, grp_rows: int = 0, grp_valid: int = 0, im2col: bool = False) -> str:
    """The same for conv_pw_h2_kernel."""
    return H2_KERNELS.get(_lib.load().fgn_h2_row_tile(rows, cout, k, grp_rows, grp_valid), 'conv_pw_h2_kernel<?, %s>') % ('true' if im2col else 'false')


class gemm_math:
    """``with ops.gemm_math('f32'):`` - the layers PACKED inside use that arithmetic (None: no change).  Training packs its
    head layers under 'f32': the optimizer rewrites their weights in place every step and a plane image would have to
    be re-derived each time."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        global GEMM_MATH
        self.prev = GEMM_MATH
        if self.mode is not None:
            GEMM_MATH = self.mode
        return self

    def __exit__(self, *exc):
        global GEMM_MATH
        GEMM_MATH = self.prev
        return False
FUSED_SHORTCUT_STRIDES = tuple(int(v) for v in os.environ.get('FGN_FUSED_SHORTCUT_STRIDES', '1,2').split(','))

_TILES = {1: (128, 128, 64, 64, 2), 2: (64, 128, 32, 64, 3), 3: (128, 64, 64, 32, 3), 4: (64, 64, 32, 32, 4)}


def kernel_name(kid: int) -> str:
    """fgn_conv2d_kernel_id -> the kernel name in a rocprofv3 kernel trace."""
    mode = kid % 10
    bm, bn, wm, wn, mw = _TILES[kid // 10]
    if mode == 4:
        return 'conv_pw_persist_kernel'
    if mode == 3:
        return f'conv_igemm_kernel<{bm}, {bn}, {wm}, {wn}, *, {mw}>'
    return f'conv_igemm_dma_kernel<{bm}, {bn}, {wm}, {wn}, 2, {mw}, {mode}>'


class ZeroArena:
    """One zero-filled allocation per episode from which the small zero-initialised outputs of the selection
    / head kernels are carved (counters, logits of RoIs beyond the device count, ...): one fill kernel
    instead of about ten 5 us ones on the critical path.  Opened with ``ops.arena``."""

    def __init__(self, device, nbytes: int = 2 << 20):
        self.buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        self.off = 0

    def take(self, shape, dtype):
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        start = (self.off + 255) // 256 * 256
        if start + n > self.buf.numel():
            return None
        self.off = start + n
        return self.buf[start:start + n].view(dtype).view(*shape)


_ARENA: Optional[ZeroArena] = None


class arena:
    """``with ops.arena(device):`` - the zero arena of one episode; closed on every exit path, so an error raised
    mid-episode cannot leave a stale arena (allocated on another stream) for later ``ops.zeros`` callers.  Arenas do
    not nest and belong to one caller thread (the detector is single-caller, as the reference's is)."""

    def __init__(self, device):
        self.device = device

    def __enter__(self):
        global _ARENA
        if _ARENA is not None:
            raise _lib.FgnHipError('ops.arena: an episode is already being queued in this process')
        _ARENA = ZeroArena(self.device)
        return _ARENA

    def __exit__(self, *exc):
        global _ARENA
        _ARENA = None
        return False


def zeros(shape, device, dtype=torch.float32) -> torch.Tensor:
    """torch.zeros, served from the episode's arena when one is open on this device."""
    a = _ARENA
    if a is not None and a.buf.device == torch.device(device):
        t = a.take(tuple(shape), dtype)
        if t is not None:
            return t
    return torch.zeros(tuple(shape), device=device, dtype=dtype)


class ConvProfile(list):
    """Record list for the ``PROFILE`` hook with a pool of pre-created timing events: creating HIP
    timing events inside a timed region stalls now and then (tens of ms when the runtime grows its
    event pool), so a benchmark creates them up front with ``reserve``."""

    def __init__(self):
        super().__init__()
        self.pool = []

    def reserve(self, n_events: int):
        self.pool.extend(torch.cuda.Event(enable_timing=True) for _ in range(n_events))
        return self

    def event(self):
        """A timing event recorded on the current stream."""
        e = self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def arm(self):
        """Arm the library so that the next convolution-family kernel launched by this thread stamps a fresh
        (start, stop) event pair with its own start and end (fgn_profile_next_launch)."""
        pair = []
        for _ in range(2):
            if self.pool:
                pair.append(self.pool.pop())        # pool events have been recorded once: their handles exist
            else:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                pair.append(e)
        _lib.check(_lib.load().fgn_profile_next_launch(pair[0].cuda_event, pair[1].cuda_event), 'fgn_profile_next_launch')
        return pair


# ---- launch records of the dominant kernel that work inside a replayed hipGraph (include/fgn_hip.h: fgn_profile_stamps) ----
def new_stamp_records(capacity: int, device) -> torch.Tensor:
    """``capacity`` launch records (fgn_profile_stamp_words() x uint64 each, held as int64): {start, sum, shards arrived,
    executions, shortest, longest, ...}."""
    t = torch.zeros((capacity, _lib.load().fgn_profile_stamp_words()), dtype=torch.int64, device=device)
    t[:, 4] = -1
    return t


def arm_stamps(records: Optional[torch.Tensor]) -> int:
    """Arm (or, with None, disarm) the calling thread: every launch of conv_pw_persist_kernel takes the next record.
    Returns the number of records handed out since the previous call."""
    if records is None:
        return int(_lib.load().fgn_profile_stamps(None, 0))
    _chk(records, 'records', torch.int64)
    if records.dim() != 2 or records.shape[1] != _lib.load().fgn_profile_stamp_words():
        raise _lib.FgnHipError('arm_stamps: records must be [capacity, fgn_profile_stamp_words()] int64')
    return int(_lib.load().fgn_profile_stamps(_ptr(records), records.shape[0]))


def reset_stamps(records: torch.Tensor) -> None:
    """Forget the executions recorded so far (call with no launch of the graph in flight)."""
    records[:, 1] = 0
    records[:, 3] = 0
    records[:, 4] = -1
    records[:, 5] = 0


def read_stamps(records: torch.Tensor, n: int) -> list:
    """[{executions, total_us, min_us, max_us}] of the first ``n`` records (10 ns ticks -> microseconds)."""
    r = records[:n].cpu().numpy().view(np.uint64)
    return [dict(executions=int(x[3]), total_us=float(x[1]) * 0.01,
                 min_us=(float(x[4]) * 0.01 if x[3] else None), max_us=(float(x[5]) * 0.01 if x[3] else None)) for x in r]


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def phase_signal(counter: torch.Tensor) -> None:
    """Bump the int32 device counter from the current stream (inside a captured graph: on every replay)."""
    _chk(counter, 'counter', torch.int32)
    _lib.check(_lib.load().fgn_phase_signal(_ptr(counter), _stream()), 'fgn_phase_signal')


def phase_wait(counter: torch.Tensor, target: int, timeout_us: int = 50000) -> None:
    """Hold the current stream until the counter has reached ``target`` (or the timeout has passed)."""
    _chk(counter, 'counter', torch.int32)
    _lib.check(_lib.load().fgn_phase_wait(_ptr(counter), int(target), int(timeout_us), _stream()), 'fgn_phase_wait')


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, name: str, dtype=torch.float32):
    if not t.is_cuda:
        raise _lib.FgnHipError(f'{name} must be a device tensor (no CPU fallback)')
    if t.dtype != dtype:
        raise _lib.FgnHipError(f'{name} must be {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise _lib.FgnHipError(f'{name} must be contiguous')


def _f4(vals):
    return (C.c_float * 4)(*[float(v) for v in vals])


# --------------------------------------------------------------------------------------
# convolution
# --------------------------------------------------------------------------------------
@dataclass
class ConvLayer:
    """A packed convolution: weights [cout_pad, KH*KW*Cin (padded to x32)], folded
    per-channel epilogue scale/shift (eval-mode BN or bias)."""
    w: torch.Tensor
    scale: Optional[torch.Tensor]
    shift: Optional[torch.Tensor]
    cin: int
    cout: int
    cout_pad: int
    kh: int
    kw: int
    stride: int
    pad: int
    relu: bool
    w3: Optional[torch.Tensor] = None      # bf16-plane image of w (pack_x3) for conv_pw_x3_kernel, or None
    wh: Optional[torch.Tensor] = None      # f16-plane image of w (pack_h2) for conv_pw_h2_kernel, or None

    def to(self, device):
        self.w = self.w.to(device)
        self.scale = None if self.scale is None else self.scale.to(device)
        self.shift = None if self.shift is None else self.shift.to(device)
        self.w3 = None if self.w3 is None else self.w3.to(device)
        self.wh = None if self.wh is None else self.wh.to(device)
        return self


def _x3_ok(cin: int, cout: int) -> bool:
    """Whether a layer gets a bf16-plane image: the shapes conv_pw_x3_kernel takes and wins on (fgn_x3_row_tile decides per
    launch; a layer whose 128-column tiles would be under 70 % real channels never passes it)."""
    npad = (cout + 127) // 128 * 128
    return GEMM_MATH == 'x3' and cin % 32 == 0 and cin >= 64 and cout % 4 == 0 and cout * 10 >= npad * 7


def _h2_ok(cin: int, cout: int) -> bool:
    """The same for the f16-plane image of conv_pw_h2_kernel (GEMM_MATH 'h2': the launches 'x3' would take, and - on its
    64-column tile - layers of 48 .. 64 output channels; fgn_h2_row_tile decides per launch)."""
    npad = (cout + 127) // 128 * 128
    return GEMM_MATH == 'h2' and cin % 32 == 0 and cin >= 64 and cout % 4 == 0 and (cout * 10 >= npad * 7 or 48 <= cout <= 64)


def _h2_conv_ok(cin: int, cout: int, kh: int, kw: int) -> bool:
    """Whether a 3x3 (any stride) or strided 1x1 convolution gets an f16-plane image for the implicit-GEMM form of
    conv_pw_h2_kernel (fgn_conv2d_pair_h2_nhwc_f32): Cin / 32 a power of two."""
    ct = cin // 32
    return _h2_ok(cin, cout) and (kh, kw) in ((1, 1), (3, 3)) and ct & (ct - 1) == 0


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor] = None, bn: Optional[dict] = None,
              stride: int = 1, pad: int = 0, relu: bool = False, eps: float = 1e-5,
              pad_cin_to: Optional[int] = None) -> ConvLayer:
    """weight [Cout,Cin,KH,KW] (torch layout) -> ConvLayer.  ``bn`` = dict(weight, bias,
    running_mean, running_var) folds eval-mode BatchNorm into scale/shift."""
    weight = weight.detach().float()
    cout, cin, kh, kw = weight.shape
    if pad_cin_to is not None and pad_cin_to > cin:
        weight = torch.cat([weight, weight.new_zeros(cout, pad_cin_to - cin, kh, kw)], dim=1)
        cin = pad_cin_to
    cout_pad = (cout + 127) // 128 * 128
    if cin == 4:
        # stem layout of the LDS-DMA kernel: one 32-float K-tile per filter row = 8 pixels x 4 channels
        # (columns beyond kw are zero)
        if kw > 8:
            raise _lib.FgnHipError('pack_conv: a 4-channel (image) conv needs kw <= 8')
        w = weight.new_zeros(cout_pad, kh, 8, 4)
        w[:cout, :, :kw, :] = weight.permute(0, 2, 3, 1)
        w = w.reshape(cout_pad, kh * 32)
    else:
        k = kh * kw * cin
        k_pad = (k + 31) // 32 * 32
        w = weight.new_zeros(cout_pad, k_pad)
        w[:cout, :k] = weight.permute(0, 2, 3, 1).reshape(cout, k)
    scale = shift = None
    if bn is not None:
        scale = (bn['weight'].float() / torch.sqrt(bn['running_var'].float() + eps))
        shift = bn['bias'].float() - bn['running_mean'].float() * scale
        if bias is not None:
            shift = shift + bias.float() * scale
    elif bias is not None:
        shift = bias.detach().float().clone()
    pw = kh == 1 and kw == 1 and stride == 1 and pad == 0
    w3 = pack_x3(w) if (pw and _x3_ok(cin, cout)) else None
    wh = pack_h2(w) if ((pw and _h2_ok(cin, cout)) or (not pw and cin != 4 and _h2_conv_ok(cin, cout, kh, kw))) else None
    return ConvLayer(w.contiguous(), None if scale is None else scale.contiguous(),
                     None if shift is None else shift.contiguous(), cin, cout, cout_pad, kh, kw,
                     stride, pad, relu, w3, wh)


def conv2d(x: torch.Tensor, layer: ConvLayer, residual: Optional[torch.Tensor] = None,
           in_scale: Optional[torch.Tensor] = None, n_img_dev: Optional[torch.Tensor] = None,
           n_img: Optional[int] = None, a_img_div: int = 1, out: Optional[torch.Tensor] = None,
           tile_hint: int = 0) -> torch.Tensor:
    """x [n_in, H, W, Cin] -> y [n_img, Ho, Wo, Cout]; n_img defaults to n_in * a_img_div."""
    _chk(x, 'x')
    n_in, H, W, cin = x.shape
    if cin != layer.cin:
        raise _lib.FgnHipError(f'conv2d: Cin {cin} != layer Cin {layer.cin}')
    if n_img is None:
        n_img = n_in * a_img_div
    if n_img > n_in * a_img_div:
        raise _lib.FgnHipError('conv2d: n_img exceeds the input batch')
    ho = (H + 2 * layer.pad - layer.kh) // layer.stride + 1
    wo = (W + 2 * layer.pad - layer.kw) // layer.stride + 1
    if out is None:
        out = torch.empty((n_img, ho, wo, layer.cout), device=x.device, dtype=torch.float32)
    else:
        _chk(out, 'out')
        if tuple(out.shape) != (n_img, ho, wo, layer.cout):
            raise _lib.FgnHipError('conv2d: bad out shape')
    if residual is not None:
        _chk(residual, 'residual')
        if residual.shape != out.shape:
            raise _lib.FgnHipError('conv2d: residual shape mismatch')
    if in_scale is not None:
        _chk(in_scale, 'in_scale')
        if tuple(in_scale.shape) != (n_img, cin):
            raise _lib.FgnHipError(f'conv2d: in_scale must be [{n_img},{cin}], got {tuple(in_scale.shape)}')
    if n_img_dev is not None:
        _chk(n_img_dev, 'n_img_dev', torch.int32)
    prof = PROFILE
    L = _lib.load()
    if prof is not None:
        e0, e1 = prof.arm()
    pw = layer.kh == 1 and layer.kw == 1 and layer.stride == 1 and layer.pad == 0
    if layer.wh is not None and not pw and in_scale is None and a_img_div == 1 and tile_hint == 0 and residual is None and \
            n_img_dev is None and x.numel() * 4 < 0x7fffff00 and out.numel() < (1 << 31) and \
            L.fgn_h2_row_tile(n_img * ho * wo, layer.cout, layer.kh * layer.kw * cin, 0, 0) > 0:
        # 3x3 / strided convolution as an implicit GEMM on conv_pw_h2_kernel (one tensor)
        rc = L.fgn_conv2d_pair_h2_nhwc_f32(_ptr(x), n_img, H, W, None, 0, 1, 1, layer.wh.data_ptr(), _ptr(out), None,
                                           _ptr(layer.scale), _ptr(layer.shift), cin, layer.cout, layer.cout_pad, layer.kh,
                                           layer.kw, layer.stride, layer.pad, int(layer.relu), _stream())
        _lib.check(rc, 'fgn_conv2d_pair_h2_nhwc_f32')
        if prof is not None:
            k = layer.kh * layer.kw * cin
            flop = 2.0 * ho * wo * layer.cout * k
            prof.append(dict(kind='conv', kernel=h2_kernel(n_img * ho * wo, layer.cout, k, im2col=True), math='h2', e0=e0, e1=e1,
                             flop_direct=flop, flop_issued=flop, n_img=n_img, n_img_dev=None, gemm=(1, ho * wo, layer.cout, k),
                             residual=False, shape=(n_img, H, W, cin, layer.cout, layer.kh, layer.stride)))
        return out
    if pw and (layer.w3 is not None or layer.wh is not None) and in_scale is None and a_img_div == 1 and tile_hint == 0 and \
            x.numel() * 4 < 0x7fffff00 and out.numel() < (1 << 31) and \
            (L.fgn_h2_row_tile if layer.wh is not None else L.fgn_x3_row_tile)(n_img * ho * wo, layer.cout, cin, 0, 0) > 0:
        h2 = layer.wh is not None
        f = L.fgn_conv1x1_h2_nhwc_f32 if h2 else L.fgn_conv1x1_x3_nhwc_f32
        rc = f(_ptr(x), (layer.wh if h2 else layer.w3).data_ptr(), _ptr(out), _ptr(layer.scale), _ptr(layer.shift),
               _ptr(residual), _ptr(n_img_dev), n_img, H, W, cin, layer.cout, layer.cout_pad, int(layer.relu), _stream())
        _lib.check(rc, 'fgn_conv1x1_h2_nhwc_f32' if h2 else 'fgn_conv1x1_x3_nhwc_f32')
        if prof is not None:
            flop = 2.0 * ho * wo * layer.cout * cin
            prof.append(dict(kind='conv', kernel=(h2_kernel if h2 else x3_kernel)(n_img * ho * wo, layer.cout, cin),
                             math='h2' if h2 else 'x3', e0=e0, e1=e1, flop_direct=flop, flop_issued=flop,
                             n_img=n_img, n_img_dev=n_img_dev, gemm=(1, ho * wo, layer.cout, cin),
                             residual=residual is not None, shape=(n_img, H, W, cin, layer.cout, 1, 1)))
        return out
    ws_bytes = L.fgn_conv2d_workspace_bytes(n_img, H, W, cin, layer.cout, layer.kh, layer.kw, layer.stride,
                                            layer.pad, tile_hint)
    ws = torch.empty(ws_bytes, device=x.device, dtype=torch.uint8) if ws_bytes else None
    rc = L.fgn_conv2d_nhwc_f32(
        _ptr(x), _ptr(layer.w), _ptr(out), _ptr(layer.scale), _ptr(layer.shift), _ptr(residual),
        _ptr(in_scale), _ptr(n_img_dev), n_img, H, W, cin, layer.cout, layer.cout_pad, layer.kh, layer.kw,
        layer.stride, layer.pad, a_img_div, int(layer.relu), tile_hint, _ptr(ws), ws_bytes, _stream())
    _lib.check(rc, 'fgn_conv2d_nhwc_f32')
    if prof is not None:
        kid = L.fgn_conv2d_kernel_id(n_img, H, W, cin, layer.cout, layer.cout_pad, layer.kh, layer.kw, layer.stride,
                                     layer.pad, a_img_div, int(in_scale is not None), int(residual is not None),
                                     tile_hint)
        flop = 2.0 * ho * wo * layer.cout * layer.kh * layer.kw * (3 if cin == 4 else cin)
        prof.append(dict(kind='conv', kernel=kernel_name(kid), e0=e0, e1=e1, flop_direct=flop,
                         flop_issued=flop, n_img=n_img, n_img_dev=n_img_dev,
                         gemm=(1, ho * wo, layer.cout, layer.kh * layer.kw * cin),       # groups, rows per image, N, K
                         residual=residual is not None,
                         shape=(n_img, H, W, cin, layer.cout, layer.kh, layer.stride)))
    return out


def pack_x3(w: torch.Tensor, mfma32: bool = False) -> torch.Tensor:
    """w [G, N, K] (or [N, K]) f32 -> the weight image of ``conv_pw_x3_kernel`` (csrc/conv_pw_x3.h): every value as the
    EXACT sum of three bf16 values (p1 = w with the low 16 bits cleared, p2 the same of w - p1, p3 = w - p1 - p2), laid
    out [G][K / 32][plane][Npad][32] bf16 with N padded to 128, the k order inside a K-tile that of the kernel's MFMA
    operand (chunk g = k 4g..4g+3, 16+4g..16+4g+3) and the four 16-byte chunks of a 64-byte row XOR-ed with
    tau[(n >> 2) & 3], tau = (0, 3, 2, 1) - tile by tile the LDS image the kernel's LDS-DMA writes, conflict-free for its
    ds_read_b128.  ``mfma32``: the image of the v_mfma_f32_32x32x16_bf16 instances of the experiments build (k in order,
    XOR with (n >> 2) & 3).  uint8 tensor on w's device."""
    w = w.detach().float()
    if w.dim() == 2:
        w = w[None]
    G, N, K = w.shape
    if K % 32:
        raise _lib.FgnHipError('pack_x3: K must be a multiple of 32')
    npad = (N + 127) // 128 * 128
    wp = w.new_zeros(G, npad, K)
    wp[:, :N] = w
    planes = []
    r = wp
    for _ in range(3):
        hi = (r.contiguous().view(torch.int32) & -65536).view(torch.float32)
        planes.append((hi.view(torch.int32) >> 16).to(torch.int16))
        r = r - hi                                     # exact: hi holds the leading bits of r
    pl = torch.stack(planes, 1)                        # [G, 3, npad, K] bf16 bit patterns
    sh16 = not mfma32
    if sh16:      # v_mfma_f32_16x16x32_bf16: lane group g reads the f32 chunks g and g + 4 of an activation row (conflict-free
        # in ds_read_b128's lane groups), so chunk g of a weight row holds k = 4g..4g+3, 16+4g..16+4g+3 of the K-tile
        korder = torch.tensor([4 * g + j if j < 4 else 16 + 4 * g + j - 4 for g in range(4) for j in range(8)], device=w.device)
        pl = pl.view(G, 3, npad, K // 32, 32)[..., korder]
    pl = pl.reshape(G, 3, npad, K // 32, 4, 8)         # K -> (K-tile, chunk, 8)
    n = torch.arange(npad, device=w.device)
    swz = (n >> 2) & 3
    if sh16:
        swz = torch.tensor([0, 3, 2, 1], device=w.device)[swz]
    src = torch.arange(4, device=w.device)[None, :] ^ swz[:, None]                       # physical chunk c holds logical c ^ swz
    pl = torch.gather(pl, 4, src[None, None, :, None, :, None].expand(G, 3, npad, K // 32, 4, 8))
    img = pl.permute(0, 3, 1, 2, 4, 5).contiguous()    # [G, KT, 3, npad, 4, 8]
    return img.view(torch.uint8).reshape(-1)


def gemm_x3(x: torch.Tensor, image: torch.Tensor, cout: int, shift: Optional[torch.Tensor] = None,
            residual: Optional[torch.Tensor] = None, relu: bool = False, groups: int = 1, grp_valid: Optional[int] = None,
            bm: int = 0, nterms: int = 6, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [rows, K] (grouped: [groups, grp_rows, K]) times the ``pack_x3`` image -> [rows, cout] on conv_pw_x3_kernel."""
    _chk(x, 'x')
    K = x.shape[-1]
    rows = x.numel() // K
    grp_rows = rows // groups
    npad = (cout + 127) // 128 * 128
    L = _lib.load()
    if image.numel() != L.fgn_x3_image_bytes(K, npad, groups):
        raise _lib.FgnHipError('gemm_x3: image size does not match K / cout / groups')
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (cout,), device=x.device, dtype=torch.float32)
    rc = L.fgn_gemm_x3_f32(_ptr(x), image.data_ptr(), _ptr(out), _ptr(shift), _ptr(residual), rows, K, cout, npad, int(relu),
                           grp_rows, grp_rows if grp_valid is None else grp_valid, groups, bm, nterms, _stream())
    _lib.check(rc, 'fgn_gemm_x3_f32')
    return out


def pack_h2(w: torch.Tensor) -> torch.Tensor:
    """w [G, N, K] (or [N, K]) f32 -> the weight image of ``conv_pw_h2_kernel`` (csrc/conv_pw_h2.h): every output column n of
    a group scaled by the power of two that puts its largest |w| into [2^14, 2^15) (an all-zero column: 1), then every value
    as two f16 planes, hi = f16(w s) and lo = f16(w s - hi) (round to nearest; hi + lo = w s to within 2^-23 |w s|: one f32 ulp at worst), laid
    out [G][K / 32][plane][Npad][32] f16 with ``pack_x3``'s k order and chunk swizzle, followed by the inverse scales
    [G][Npad] f32.  uint8 tensor on w's device."""
    w = w.detach().float()
    if w.dim() == 2:
        w = w[None]
    G, N, K = w.shape
    if K % 32:
        raise _lib.FgnHipError('pack_h2: K must be a multiple of 32')
    npad = (N + 127) // 128 * 128
    wp = w.new_zeros(G, npad, K)
    wp[:, :N] = w
    amax = wp.abs().amax(dim=2)                                              # [G, npad]
    e = torch.floor(torch.log2(torch.where(amax > 0, amax, torch.ones_like(amax)).double()))
    e = torch.where((amax > 0) & torch.isfinite(amax), e, torch.full_like(e, 14.0)).clamp_(-100.0, 100.0)
    scale = torch.pow(torch.tensor(2.0, dtype=torch.float64, device=w.device), 14.0 - e).float()
    inv = torch.pow(torch.tensor(2.0, dtype=torch.float64, device=w.device), e - 14.0).float()
    ws = wp * scale[:, :, None]                                              # exact (power of two)
    hi = ws.to(torch.float16)
    lo = (ws - hi.float()).to(torch.float16)
    pl = torch.stack((hi, lo), 1).view(torch.int16)                          # [G, 2, npad, K] f16 bit patterns
    korder = torch.tensor([4 * g + j if j < 4 else 16 + 4 * g + j - 4 for g in range(4) for j in range(8)], device=w.device)
    pl = pl.view(G, 2, npad, K // 32, 32)[..., korder]
    pl = pl.reshape(G, 2, npad, K // 32, 4, 8)                               # K -> (K-tile, chunk, 8)
    n = torch.arange(npad, device=w.device)
    swz = torch.tensor([0, 3, 2, 1], device=w.device)[(n >> 2) & 3]
    src = torch.arange(4, device=w.device)[None, :] ^ swz[:, None]           # physical chunk c holds logical c ^ swz
    pl = torch.gather(pl, 4, src[None, None, :, None, :, None].expand(G, 2, npad, K // 32, 4, 8))
    img = pl.permute(0, 3, 1, 2, 4, 5).contiguous()                          # [G, KT, 2, npad, 4, 8]
    return torch.cat((img.view(torch.uint8).reshape(-1), inv.contiguous().view(torch.uint8).reshape(-1)))


def gemm_h2(x: torch.Tensor, image: torch.Tensor, cout: int, shift: Optional[torch.Tensor] = None,
            residual: Optional[torch.Tensor] = None, relu: bool = False, groups: int = 1, grp_valid: Optional[int] = None,
            bm: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [rows, K] (grouped: [groups, grp_rows, K]) times the ``pack_h2`` image -> [rows, cout] on conv_pw_h2_kernel."""
    _chk(x, 'x')
    K = x.shape[-1]
    rows = x.numel() // K
    grp_rows = rows // groups
    npad = (cout + 127) // 128 * 128
    L = _lib.load()
    if image.numel() != L.fgn_h2_image_bytes(K, npad, groups):
        raise _lib.FgnHipError('gemm_h2: image size does not match K / cout / groups')
    if out is None:
        out = torch.empty(tuple(x.shape[:-1]) + (cout,), device=x.device, dtype=torch.float32)
    rc = L.fgn_gemm_h2_f32(_ptr(x), image.data_ptr(), _ptr(out), _ptr(shift), _ptr(residual), rows, K, cout, npad, int(relu),
                           grp_rows, grp_rows if grp_valid is None else grp_valid, groups, bm, _stream())
    _lib.check(rc, 'fgn_gemm_h2_f32')
    return out


@dataclass
class DualConvLayer:
    """Two 1x1 / stride 1 convolutions with their eval-mode BatchNorms, packed for ONE K loop (``conv1x1_dual``):
    w [cout_pad, cin1 + cin2] with the two BN scales folded into its rows (fp64 fold, one rounding), shift = sum of the
    two BN shifts."""
    w: torch.Tensor
    shift: torch.Tensor
    cin1: int
    cin2: int
    cout: int
    cout_pad: int
    relu: bool
    w3: Optional[torch.Tensor] = None
    wh: Optional[torch.Tensor] = None

    def to(self, device):
        self.w, self.shift = self.w.to(device), self.shift.to(device)
        self.w3 = None if self.w3 is None else self.w3.to(device)
        self.wh = None if self.wh is None else self.wh.to(device)
        return self


def pack_conv_dual(w1: torch.Tensor, bn1: dict, w2: torch.Tensor, bn2: dict, relu: bool = True, eps: float = 1e-5) -> DualConvLayer:
    """w1 [Cout,Cin1,1,1] + bn1 (a bottleneck's conv3 / bn3), w2 [Cout,Cin2,1,1] + bn2 (its shortcut conv / bn)."""
    cout, cin1 = w1.shape[:2]
    cin2 = w2.shape[1]
    if tuple(w1.shape[2:]) != (1, 1) or tuple(w2.shape[2:]) != (1, 1) or w2.shape[0] != cout or cin1 % 32 or cin2 % 32 or cout % 4:
        raise _lib.FgnHipError('pack_conv_dual: two 1x1 kernels with one Cout, Cin1 / Cin2 multiples of 32, Cout % 4 == 0')
    rows, shift = [], 0.0
    for w, bn in ((w1, bn1), (w2, bn2)):
        sc = bn['weight'].double() / torch.sqrt(bn['running_var'].double() + eps)
        shift = shift + (bn['bias'].double() - bn['running_mean'].double() * sc)
        rows.append(w.detach().double().reshape(cout, -1) * sc[:, None])
    cout_pad = (cout + 127) // 128 * 128
    wp = torch.zeros(cout_pad, cin1 + cin2, dtype=torch.float32)
    wp[:cout] = torch.cat(rows, 1).float()
    return DualConvLayer(wp.contiguous(), shift.float().contiguous(), cin1, cin2, cout, cout_pad, relu,
                         pack_x3(wp) if _x3_ok(cin1 + cin2, cout) else None, pack_h2(wp) if _h2_ok(cin1 + cin2, cout) else None)


def strided_rows(shapes, stride: int, device) -> torch.Tensor:
    """Row table of a 1x1 / stride ``stride`` convolution over NHWC tensors that lie one behind the other in one buffer
    (``shapes`` = [(n, H, W), ...]): output row m (images, then output rows, then output columns, tensor after tensor) ->
    the input row it reads.  int32 on ``device``; for ``conv1x1_dual(..., x2_rows=)``."""
    out, base = [], 0
    for n, H, W in shapes:
        ho, wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        img = torch.arange(n, dtype=torch.int64)[:, None, None] * (H * W)
        oy = torch.arange(ho, dtype=torch.int64)[None, :, None] * (stride * W)
        ox = torch.arange(wo, dtype=torch.int64)[None, None, :] * stride
        out.append((base + img + oy + ox).reshape(-1))
        base += n * H * W
    return torch.cat(out).to(torch.int32).to(device)


def conv1x1_dual(x1: torch.Tensor, x2: torch.Tensor, layer: DualConvLayer, out: Optional[torch.Tensor] = None,
                 x2_rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x1 [..., Cin1] -> relu?(x1 W1^T + x2' W2^T + shift) [..., Cout], x2' = x2 over the same leading dims (rows), or -
    with ``x2_rows`` (int32 [rows], ``strided_rows``) - row x2_rows[m] of x2 [..., Cin2] for output row m."""
    _chk(x1, 'x1')
    _chk(x2, 'x2')
    rows = x1.numel() // layer.cin1
    if x1.shape[-1] != layer.cin1 or x2.shape[-1] != layer.cin2 or (x2_rows is None and x1.shape[:-1] != x2.shape[:-1]):
        raise _lib.FgnHipError('conv1x1_dual: operand shapes inconsistent with the layer')
    x2_total = x2.numel() // layer.cin2
    if x2_rows is not None:
        _chk(x2_rows, 'x2_rows', torch.int32)
        if x2_rows.numel() != rows:
            raise _lib.FgnHipError('conv1x1_dual: x2_rows must hold one input row per output row')
    shape = tuple(x1.shape[:-1]) + (layer.cout,)
    if out is None:
        out = torch.empty(shape, device=x1.device, dtype=torch.float32)
    else:
        _chk(out, 'out')
        if tuple(out.shape) != shape:
            raise _lib.FgnHipError('conv1x1_dual: bad out shape')
    prof = PROFILE
    L = _lib.load()
    if prof is not None:
        e0, e1 = prof.arm()
    use_x3 = (layer.w3 is not None or layer.wh is not None) and max(x1.numel(), x2.numel()) * 4 < 0x7fffff00 and \
        (L.fgn_h2_row_tile if layer.wh is not None else L.fgn_x3_row_tile)(rows, layer.cout, layer.cin1 + layer.cin2, 0, 0) > 0
    use_h2 = use_x3 and layer.wh is not None
    if use_x3:
        f = L.fgn_conv1x1_dual_h2_nhwc_f32 if use_h2 else L.fgn_conv1x1_dual_x3_nhwc_f32
        rc = f(_ptr(x1), _ptr(x2), _ptr(x2_rows), x2_total, (layer.wh if use_h2 else layer.w3).data_ptr(), _ptr(out),
               _ptr(layer.shift), rows, layer.cin1, layer.cin2, layer.cout, layer.cout_pad, int(layer.relu), _stream())
    else:
        rc = L.fgn_conv1x1_dual_nhwc_f32(_ptr(x1), _ptr(x2), _ptr(x2_rows), x2_total, _ptr(layer.w), _ptr(out), _ptr(layer.shift),
                                         rows, layer.cin1, layer.cin2, layer.cout, layer.cout_pad, int(layer.relu), _stream())
    _lib.check(rc, 'fgn_conv1x1_dual_nhwc_f32')
    if prof is not None:
        k = layer.cin1 + layer.cin2
        flop = 2.0 * rows * layer.cout * k
        prof.append(dict(kind='conv', kernel=h2_kernel(rows, layer.cout, k) if use_h2 else x3_kernel(rows, layer.cout, k) if use_x3 else 'conv_pw_persist_kernel',
                         math='h2' if use_h2 else 'x3' if use_x3 else 'f32', e0=e0, e1=e1, flop_direct=flop, flop_issued=flop,
                         n_img=1, n_img_dev=None, gemm=(1, rows, layer.cout, k), residual=False,
                         shape=(1, rows, 1, k, layer.cout, 1, 1)))
    return out


def conv2d_pair(x0: torch.Tensor, x1: torch.Tensor, layer: ConvLayer, out0: Optional[torch.Tensor] = None,
                out1: Optional[torch.Tensor] = None):
    """The same convolution (weights, folded BN, ReLU) on two NHWC tensors of different geometry in ONE launch - the
    query map and the support maps of a backbone layer that strides over the spatial structure (3x3 / stride 2, the
    1x1 / stride 2 shortcut, the stem).  Per tensor the arithmetic of ``conv2d`` without split-K.  -> (y0, y1)."""
    _chk(x0, 'x0')
    _chk(x1, 'x1')
    if x0.shape[3] != layer.cin or x1.shape[3] != layer.cin:
        raise _lib.FgnHipError(f'conv2d_pair: Cin {x0.shape[3]} / {x1.shape[3]} != layer Cin {layer.cin}')
    outs = []
    for x, out in ((x0, out0), (x1, out1)):
        n, H, W, _ = x.shape
        ho = (H + 2 * layer.pad - layer.kh) // layer.stride + 1
        wo = (W + 2 * layer.pad - layer.kw) // layer.stride + 1
        if out is None:
            out = torch.empty((n, ho, wo, layer.cout), device=x.device, dtype=torch.float32)
        else:
            _chk(out, 'out')
            if tuple(out.shape) != (n, ho, wo, layer.cout):
                raise _lib.FgnHipError('conv2d_pair: bad out shape')
        outs.append(out)
    prof = PROFILE
    L = _lib.load()
    if prof is not None:
        e0, e1 = prof.arm()
    rows = sum(o.shape[0] * o.shape[1] * o.shape[2] for o in outs)
    k = layer.kh * layer.kw * layer.cin
    span = abs(x0.data_ptr() - x1.data_ptr()) + max(x0.numel(), x1.numel()) * 4
    if layer.wh is not None and layer.cin != 4 and span < 0x7fffff00 and rows * layer.cout < (1 << 31) and \
            L.fgn_h2_row_tile(rows, layer.cout, k, 0, 0) > 0:
        # the implicit GEMM of both tensors on conv_pw_h2_kernel
        rc = L.fgn_conv2d_pair_h2_nhwc_f32(_ptr(x0), x0.shape[0], x0.shape[1], x0.shape[2], _ptr(x1), x1.shape[0], x1.shape[1],
                                           x1.shape[2], layer.wh.data_ptr(), _ptr(outs[0]), _ptr(outs[1]), _ptr(layer.scale),
                                           _ptr(layer.shift), layer.cin, layer.cout, layer.cout_pad, layer.kh, layer.kw,
                                           layer.stride, layer.pad, int(layer.relu), _stream())
        _lib.check(rc, 'fgn_conv2d_pair_h2_nhwc_f32')
        if prof is not None:
            flop = 2.0 * layer.cout * k * rows
            prof.append(dict(kind='conv', kernel=h2_kernel(rows, layer.cout, k, im2col=True), math='h2', e0=e0, e1=e1,
                             flop_direct=flop, flop_issued=flop, n_img=1, n_img_dev=None, gemm=(1, rows, layer.cout, k),
                             shape=(x0.shape[0] + x1.shape[0], x0.shape[1], x0.shape[2], layer.cin, layer.cout, layer.kh,
                                    layer.stride)))
        return outs[0], outs[1]
    rc = L.fgn_conv2d_pair_nhwc_f32(_ptr(x0), _ptr(outs[0]), x0.shape[0], x0.shape[1], x0.shape[2],
                                    _ptr(x1), _ptr(outs[1]), x1.shape[0], x1.shape[1], x1.shape[2],
                                    _ptr(layer.w), _ptr(layer.scale), _ptr(layer.shift), layer.cin, layer.cout,
                                    layer.cout_pad, layer.kh, layer.kw, layer.stride, layer.pad, int(layer.relu), _stream())
    _lib.check(rc, 'fgn_conv2d_pair_nhwc_f32')
    if prof is not None:
        per_px = 2.0 * layer.cout * layer.kh * layer.kw * (3 if layer.cin == 4 else layer.cin)
        flop = per_px * sum(o.shape[0] * o.shape[1] * o.shape[2] for o in outs)
        prof.append(dict(kind='conv', kernel='conv_igemm_dma_pair_kernel<64, 64, 32, 32, 2, 4, %d>' % (2 if layer.cin == 4 else 0),
                         e0=e0, e1=e1, flop_direct=flop, flop_issued=flop, n_img=1, n_img_dev=None,
                         gemm=(1, sum(o.shape[0] * o.shape[1] * o.shape[2] for o in outs), layer.cout,
                               layer.kh * layer.kw * (3 if layer.cin == 4 else layer.cin)),
                         shape=(x0.shape[0] + x1.shape[0], x0.shape[1], x0.shape[2], layer.cin, layer.cout, layer.kh,
                                layer.stride)))
    return outs[0], outs[1]


# --------------------------------------------------------------------------------------
# Winograd F(2x2,3x3) form of 3x3 / stride 1 / pad 1 convolutions
# --------------------------------------------------------------------------------------
@dataclass
class WinogradLayer:
    """U [m_in^2, cout_pad, Cin] = G w G^T with the per-channel epilogue scale folded in; shift [Cout].
    ``m`` = output tile edge: 4 = F(4x4,3x3), 36 tile positions; 2 = F(2x2,3x3), 16 positions."""
    u: torch.Tensor
    shift: Optional[torch.Tensor]
    cin: int
    cout: int
    cout_pad: int
    relu: bool
    m: int = 2
    u3: Optional[torch.Tensor] = None      # bf16-plane image of u (pack_x3), or None
    uh: Optional[torch.Tensor] = None      # f16-plane image of u (pack_h2), or None

    @property
    def groups(self) -> int:
        return (self.m + 2) ** 2

    def to(self, device):
        self.u = self.u.to(device)
        self.shift = None if self.shift is None else self.shift.to(device)
        self.u3 = None if self.u3 is None else self.u3.to(device)
        self.uh = None if self.uh is None else self.uh.to(device)
        return self


_WG_G = {
    2: torch.tensor([[1.0, 0.0, 0.0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0.0, 0.0, 1.0]], dtype=torch.float64),
    # Cook-Toom, points {0, 1, -1, 1/2, -2, inf} (csrc/winograd.hip holds the matching B^T / A^T)
    4: torch.tensor([[1.0, 0.0, 0.0], [1 / 3, 1 / 3, 1 / 3], [-1 / 3, 1 / 3, -1 / 3], [-16 / 15, -8 / 15, -4 / 15],
                     [1 / 15, -2 / 15, 4 / 15], [0.0, 0.0, 1.0]], dtype=torch.float64),
}
WINOGRAD_M = 4       # default output tile edge
_WG_MIN_CIN = int(os.environ.get('FGN_WG_MIN_CIN', '128'))    # F(4x4) from this many input channels (A/B knob: see winograd_pays)


def pack_winograd(weight: torch.Tensor, bias: Optional[torch.Tensor] = None, bn: Optional[dict] = None,
                  relu: bool = False, eps: float = 1e-5, m: Optional[int] = None) -> WinogradLayer:
    """weight [Cout,Cin,3,3] -> WinogradLayer (transform in fp64, stored fp32)."""
    m = WINOGRAD_M if m is None else m
    cout, cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or cin % 32 != 0 or cout % 4 != 0 or m not in _WG_G:
        raise _lib.FgnHipError('pack_winograd: needs a 3x3 kernel, Cin % 32 == 0, Cout % 4 == 0, m in (2, 4)')
    w = weight.detach().double()
    scale = shift = None
    if bn is not None:
        scale = bn['weight'].double() / torch.sqrt(bn['running_var'].double() + eps)
        shift = bn['bias'].double() - bn['running_mean'].double() * scale
        if bias is not None:
            shift = shift + bias.double() * scale
        w = w * scale[:, None, None, None]
    elif bias is not None:
        shift = bias.detach().double()
    G = _WG_G[m].to(w.device)
    u = torch.einsum('ai,ocij,bj->aboc', G, w, G)                            # [m+2,m+2,Cout,Cin]
    cout_pad = (cout + 127) // 128 * 128
    up = torch.zeros((m + 2) ** 2, cout_pad, cin, dtype=torch.float32, device=w.device)
    up[:, :cout] = u.reshape((m + 2) ** 2, cout, cin).float()
    return WinogradLayer(up.contiguous(), None if shift is None else shift.float().contiguous(), cin, cout,
                         cout_pad, relu, m, pack_x3(up) if _x3_ok(cin, cout) else None, pack_h2(up) if _h2_ok(cin, cout) else None)


_WG_G_DEV: dict = {}


def repack_conv_(layer: ConvLayer, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> ConvLayer:
    """``pack_conv`` of updated weights INTO an existing packed layer (no BatchNorm fold, Cin != 4): one strided copy -
    what a training step needs after its optimizer update.  The padding rows / columns keep the zeros of the first pack."""
    cout, cin, kh, kw = weight.shape
    if (cout, cin, kh, kw) != (layer.cout, layer.cin, layer.kh, layer.kw) or cin == 4 or layer.scale is not None:
        raise _lib.FgnHipError('repack_conv_: layer and weight do not match (or the layer folds a BatchNorm)')
    k = kh * kw * cin
    if layer.w.shape[1] == k:                      # rows are exactly K long: one strided copy
        layer.w[:cout].view(cout, kh, kw, cin).copy_(weight.detach().permute(0, 2, 3, 1))
    else:                                          # K padded to a multiple of 32: through a contiguous temporary
        layer.w[:cout, :k].copy_(weight.detach().permute(0, 2, 3, 1).reshape(cout, k))
    if bias is not None:
        layer.shift.copy_(bias.detach())
    if layer.w3 is not None:
        layer.w3.copy_(pack_x3(layer.w))
    if layer.wh is not None:
        layer.wh.copy_(pack_h2(layer.w))
    return layer


def repack_winograd_(layer: WinogradLayer, weight: torch.Tensor, bias: Optional[torch.Tensor] = None) -> WinogradLayer:
    """``pack_winograd`` of updated weights INTO an existing layer: one kernel (``fgn_winograd_pack_weights_f32``, fp64
    arithmetic on the device, one rounding - the host transform's arithmetic) instead of ~40 torch kernels."""
    cout, cin, kh, kw = weight.shape
    if (cout, cin, kh, kw) != (layer.cout, layer.cin, 3, 3):
        raise _lib.FgnHipError('repack_winograd_: layer and weight do not match')
    w = weight.detach()
    _chk(w, 'weight')
    key = (layer.m, w.device)
    G = _WG_G_DEV.get(key)
    if G is None:
        G = _WG_G_DEV[key] = _WG_G[layer.m].to(w.device).contiguous()
    rc = _lib.load().fgn_winograd_pack_weights_f32(_ptr(w), _ptr(G), _ptr(layer.u), cout, cin, layer.cout_pad, layer.m,
                                                   _stream())
    _lib.check(rc, 'fgn_winograd_pack_weights_f32')
    if bias is not None:
        layer.shift.copy_(bias.detach())
    if layer.u3 is not None:
        layer.u3.copy_(pack_x3(layer.u))
    if layer.uh is not None:
        layer.uh.copy_(pack_h2(layer.u))
    return layer


def _wg_tiles(H: int, W: int, m: int) -> int:
    return ((H + m - 1) // m) * ((W + m - 1) // m)


def winograd_fits(n_img: int, H: int, W: int, cin: int, cout: int, m: Optional[int] = None) -> bool:
    """The grouped GEMM addresses V / Mo through 32-bit buffer offsets: [groups * t_pad, C] must stay below 2 GiB."""
    m = WINOGRAD_M if m is None else m
    t_pad = (n_img * _wg_tiles(H, W, m) + 63) // 64 * 64
    rows = (m + 2) ** 2 * t_pad
    return rows * cin * 4 < 0x7fffff00 and rows * cout < (1 << 31)


def winograd_pays(n_img: int, H: int, W: int, cin: int, cout: int, m: Optional[int] = None) -> bool:
    """Measured on MI355X (tools/wg_bench.py, direct -> F(2x2) -> F(4x4)): AG-RPN conv 2.23 -> 0.95 -> 0.57 ms,
    shared_head 3x3 on 300 RoIs 0.65 -> 0.43 -> 0.29 ms, layer1 (64 channels, 200x334) 65 -> 70 -> 57 us; the
    9-RoI support head (441 pixels) loses to the launch costs of the three kernels.  Both forms need about a thousand
    output pixels and 128 input channels.  (Alone, F(4x4) wins at 64 channels too - layer1: 57 us against 65 for the direct
    form - but its three launches move 5.5x the bytes, and in the pipelined step, beside the other episode's GEMMs, the
    direct form wins: round 5, same box, three interleaved pairs 203.2-204.0 -> 204.4-204.8 img/s.)"""
    m = WINOGRAD_M if m is None else m
    return cin >= (_WG_MIN_CIN if m == 4 else 128) and n_img * H * W >= 1024 and winograd_fits(n_img, H, W, cin, cout, m)


def conv3x3_winograd(x: torch.Tensor, layer: WinogradLayer, in_scale: Optional[torch.Tensor] = None,
                     a_img_div: int = 1, n_img_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [n_in,H,W,Cin] -> relu?(conv3x3(x[i // a_img_div] * in_scale[i]) + shift) [n_in*a_img_div,H,W,Cout]."""
    _chk(x, 'x')
    n_in, H, W, cin = x.shape
    if cin != layer.cin:
        raise _lib.FgnHipError(f'conv3x3_winograd: Cin {cin} != layer Cin {layer.cin}')
    n_img = n_in * a_img_div
    if in_scale is not None:
        _chk(in_scale, 'in_scale')
        if tuple(in_scale.shape) != (n_img, cin):
            raise _lib.FgnHipError('conv3x3_winograd: in_scale must be [n_img, Cin]')
    if n_img_dev is not None:
        _chk(n_img_dev, 'n_img_dev', torch.int32)
    tiles = _wg_tiles(H, W, layer.m)
    L = _lib.load()
    f_in, f_out = (L.fgn_winograd4_input_f32, L.fgn_winograd4_output_f32) if layer.m == 4 else \
        (L.fgn_winograd_input_f32, L.fgn_winograd_output_f32)
    G = layer.groups
    t_pad = L.fgn_winograd_t_pad(n_img * tiles)
    V = torch.empty((G, t_pad, cin), device=x.device, dtype=torch.float32)
    Mo = torch.empty((G, t_pad, layer.cout), device=x.device, dtype=torch.float32)
    y = torch.empty((n_img, H, W, layer.cout), device=x.device, dtype=torch.float32)
    prof = PROFILE
    st = _stream()
    ev = [] if prof is not None else None
    if ev is not None:
        ev.append(prof.arm())
    use_x3 = (layer.u3 is not None or layer.uh is not None) and \
        (L.fgn_h2_row_tile if layer.uh is not None else L.fgn_x3_row_tile)(G * t_pad, layer.cout, cin, t_pad, n_img * tiles) > 0
    use_h2 = use_x3 and layer.uh is not None
    use_x3 = use_x3 and not use_h2
    _lib.check(f_in(_ptr(x), _ptr(in_scale), _ptr(V), _ptr(n_img_dev), n_img, a_img_div, H, W, cin, t_pad, st),
               'fgn_winograd_input_f32')
    if ev is not None:
        ev.append(prof.arm())
    if use_h2:
        _lib.check(L.fgn_winograd_gemm_h2_f32(_ptr(V), layer.uh.data_ptr(), _ptr(Mo), _ptr(n_img_dev), n_img, tiles, t_pad, cin,
                                              layer.cout, layer.cout_pad, G, st), 'fgn_winograd_gemm_h2_f32')
    elif use_x3:
        _lib.check(L.fgn_winograd_gemm_x3_f32(_ptr(V), layer.u3.data_ptr(), _ptr(Mo), _ptr(n_img_dev), n_img, tiles, t_pad, cin,
                                              layer.cout, layer.cout_pad, G, st), 'fgn_winograd_gemm_x3_f32')
    else:
        _lib.check(L.fgn_winograd_gemm_f32(_ptr(V), _ptr(layer.u), _ptr(Mo), _ptr(n_img_dev), n_img, tiles, t_pad, cin,
                                           layer.cout, layer.cout_pad, G, st), 'fgn_winograd_gemm_f32')
    if ev is not None:
        ev.append(prof.arm())
    _lib.check(f_out(_ptr(Mo), _ptr(y), _ptr(layer.shift), _ptr(n_img_dev), n_img, H, W, layer.cout, t_pad,
                     int(layer.relu), st), 'fgn_winograd_output_f32')
    if ev is not None:
        # direct-convolution FLOPs of the layer (what the reference's formulation executes) are booked on the GEMM
        # record; the MFMA work actually issued is (m+2)^2 products per m x m output tile instead of 9 m^2
        shape = (n_img, H, W, cin, layer.cout, 3, 1)
        common = dict(n_img=n_img, n_img_dev=n_img_dev, shape=shape)
        kin, kout = 'wg_input_kernel', 'wg_output_kernel'
        if layer.m == 4:     # the template instances of the F(4x4) transforms (csrc/winograd.hip): <vector width, eager>
            vi, vo = L.fgn_winograd4_variant(n_img * tiles, cin, 0), L.fgn_winograd4_variant(n_img * tiles, layer.cout, 1)
            kin = 'wg4_input_kernel<%d, %s>' % (vi // 10, 'true' if vi % 10 else 'false')
            kout = 'wg4_output_kernel<%d>' % (vo // 10)
        prof.append(dict(kind='wg_in', kernel=kin, e0=ev[0][0], e1=ev[0][1], flop_direct=0.0, flop_issued=0.0,
                         **common))
        # the grouped GEMM is a point-wise launch over [groups * t_pad] rows
        gid = L.fgn_conv2d_kernel_id(G * t_pad, 1, 1, cin, layer.cout, layer.cout_pad, 1, 1, 1, 0, 1, 0, 0, 4)   # 64x64 tile
        prof.append(dict(kind='wg_gemm', kernel=h2_kernel(G * t_pad, layer.cout, cin, t_pad, n_img * tiles) if use_h2 else
                         x3_kernel(G * t_pad, layer.cout, cin, t_pad, n_img * tiles) if use_x3 else kernel_name(gid),
                         math='h2' if use_h2 else 'x3' if use_x3 else 'f32', e0=ev[1][0], e1=ev[1][1],
                         flop_direct=2.0 * H * W * layer.cout * 9 * cin,
                         flop_issued=2.0 * G * tiles * layer.cout * cin, gemm=(G, tiles, layer.cout, cin), **common))
        prof.append(dict(kind='wg_out', kernel=kout, e0=ev[2][0], e1=ev[2][1], flop_direct=0.0,
                         flop_issued=0.0, **common))
    return y


def conv3x3_winograd_multi(xs, layer: WinogradLayer, outs) -> None:
    """The F(4x4) / F(2x2) convolution of SEVERAL inputs with one weight set through ONE grouped GEMM: every input
    [n_i, H_i, W_i, Cin] is transformed into its own tile range of a shared V, the GEMM runs over all tiles, every
    output ``outs[i]`` [n_i, H_i, W_i, Cout] (views of a caller-owned buffer) is transformed out of its range of Mo.
    (The transforms address tile t of position g at ((g * t_pad + t) * C): a tile offset is a pointer offset.)"""
    L = _lib.load()
    f_in, f_out = (L.fgn_winograd4_input_f32, L.fgn_winograd4_output_f32) if layer.m == 4 else \
        (L.fgn_winograd_input_f32, L.fgn_winograd_output_f32)
    cin, cout, G = layer.cin, layer.cout, layer.groups
    tiles = []
    for x, y in zip(xs, outs):
        _chk(x, 'x')
        _chk(y, 'out')
        if x.shape[-1] != cin or tuple(y.shape) != tuple(x.shape[:-1]) + (cout,):
            raise _lib.FgnHipError('conv3x3_winograd_multi: operand shapes inconsistent')
        tiles.append(x.shape[0] * _wg_tiles(x.shape[1], x.shape[2], layer.m))
    total = sum(tiles)
    t_pad = L.fgn_winograd_t_pad(total)
    if G * t_pad * max(cin, cout) * 4 >= 0x7fffff00:
        raise _lib.FgnHipError('conv3x3_winograd_multi: V / Mo exceed the 2 GiB descriptors')
    dev = xs[0].device
    V = torch.empty((G, t_pad, cin), device=dev, dtype=torch.float32)
    Mo = torch.empty((G, t_pad, cout), device=dev, dtype=torch.float32)
    st = _stream()
    pair = layer.m == 4 and len(xs) == 2          # one transform launch for both tensors
    prof = PROFILE if pair else None              # (the per-tensor form is not instrumented: tests / F(2x2) only)
    ev = []
    if prof is not None:
        ev.append(prof.arm())
    use_x3 = (layer.u3 is not None or layer.uh is not None) and \
        (L.fgn_h2_row_tile if layer.uh is not None else L.fgn_x3_row_tile)(G * t_pad, cout, cin, t_pad, total) > 0
    use_h2 = use_x3 and layer.uh is not None
    use_x3 = use_x3 and not use_h2
    if pair:
        (n0, h0, w0, _), (n1, h1, w1, _) = xs[0].shape, xs[1].shape
        _lib.check(L.fgn_winograd4_input2_f32(_ptr(xs[0]), n0, h0, w0, _ptr(xs[1]), n1, h1, w1, _ptr(V), cin, t_pad, st),
                   'fgn_winograd4_input2_f32')
    else:
        off = 0
        for x, n_t in zip(xs, tiles):
            n, H, W, _ = x.shape
            _lib.check(f_in(_ptr(x), None, V.data_ptr() + off * cin * 4, None, n, 1, H, W, cin, t_pad, st),
                       'fgn_winograd_input_f32')
            off += n_t
    if prof is not None:
        ev.append(prof.arm())
    if use_h2:
        _lib.check(L.fgn_winograd_gemm_h2_f32(_ptr(V), layer.uh.data_ptr(), _ptr(Mo), None, 1, total, t_pad, cin, cout,
                                              layer.cout_pad, G, st), 'fgn_winograd_gemm_h2_f32')
    elif use_x3:
        _lib.check(L.fgn_winograd_gemm_x3_f32(_ptr(V), layer.u3.data_ptr(), _ptr(Mo), None, 1, total, t_pad, cin, cout,
                                              layer.cout_pad, G, st), 'fgn_winograd_gemm_x3_f32')
    else:
        _lib.check(L.fgn_winograd_gemm_f32(_ptr(V), _ptr(layer.u), _ptr(Mo), None, 1, total, t_pad, cin, cout, layer.cout_pad,
                                           G, st), 'fgn_winograd_gemm_f32')
    if prof is not None:
        ev.append(prof.arm())
    if pair:
        _lib.check(L.fgn_winograd4_output2_f32(_ptr(Mo), _ptr(layer.shift), _ptr(outs[0]), n0, h0, w0, _ptr(outs[1]), n1, h1,
                                               w1, cout, t_pad, int(layer.relu), st), 'fgn_winograd4_output2_f32')
    else:
        off = 0
        for y, n_t in zip(outs, tiles):
            n, H, W, _ = y.shape
            _lib.check(f_out(Mo.data_ptr() + off * cout * 4, _ptr(y), _ptr(layer.shift), None, n, H, W, cout, t_pad,
                             int(layer.relu), st), 'fgn_winograd_output_f32')
            off += n_t
    if prof is not None:
        pixels = sum(x.shape[0] * x.shape[1] * x.shape[2] for x in xs)
        common = dict(n_img=1, n_img_dev=None, shape=(len(xs), pixels, 1, cin, cout, 3, 1))
        vi, vo = L.fgn_winograd4_variant(total, cin, 0), L.fgn_winograd4_variant(total, cout, 1)
        prof.append(dict(kind='wg_in', kernel='wg4_input_kernel<%d, %s>' % (vi // 10, 'true' if vi % 10 else 'false'),
                         e0=ev[0][0], e1=ev[0][1], flop_direct=0.0, flop_issued=0.0, **common))
        gid = L.fgn_conv2d_kernel_id(G * t_pad, 1, 1, cin, cout, layer.cout_pad, 1, 1, 1, 0, 1, 0, 0, 4)
        prof.append(dict(kind='wg_gemm', kernel=h2_kernel(G * t_pad, cout, cin, t_pad, total) if use_h2 else
                         x3_kernel(G * t_pad, cout, cin, t_pad, total) if use_x3 else kernel_name(gid),
                         math='h2' if use_h2 else 'x3' if use_x3 else 'f32', e0=ev[1][0], e1=ev[1][1],
                         flop_direct=2.0 * pixels * cout * 9 * cin, flop_issued=2.0 * G * total * cout * cin,
                         gemm=(G, total, cout, cin), **common))
        prof.append(dict(kind='wg_out', kernel='wg4_output_kernel<%d>' % (vo // 10), e0=ev[2][0], e1=ev[2][1],
                         flop_direct=0.0, flop_issued=0.0, **common))


# --------------------------------------------------------------------------------------
# spatial ops
# --------------------------------------------------------------------------------------
def nchw3_to_nhwc4(x: torch.Tensor) -> torch.Tensor:
    _chk(x, 'x')
    n, c, h, w = x.shape
    if c != 3:
        raise _lib.FgnHipError('nchw3_to_nhwc4: expects 3 channels')
    y = torch.empty((n, h, w, 4), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().fgn_nchw3_to_nhwc4_f32(_ptr(x), _ptr(y), n, h, w, _stream()),
               'fgn_nchw3_to_nhwc4_f32')
    return y


def maxpool3x3s2(x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(x, 'x')
    n, h, w, c = x.shape
    shape = (n, (h - 1) // 2 + 1, (w - 1) // 2 + 1, c)
    if out is None:
        y = torch.empty(shape, device=x.device, dtype=torch.float32)
    else:
        _chk(out, 'out')
        if tuple(out.shape) != shape:
            raise _lib.FgnHipError('maxpool3x3s2: bad out shape')
        y = out
    _lib.check(_lib.load().fgn_maxpool3x3s2_nhwc_f32(_ptr(x), _ptr(y), n, h, w, c, _stream()),
               'fgn_maxpool3x3s2_nhwc_f32')
    return y


def group_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, groups: int, eps: float = 1e-5,
               relu: bool = False, residual: Optional[torch.Tensor] = None, inplace: bool = False) -> torch.Tensor:
    """torch.nn.GroupNorm over NHWC x [n,H,W,C] (+ residual) (+ ReLU)."""
    _chk(x, 'x')
    _chk(gamma, 'gamma')
    _chk(beta, 'beta')
    n, h, w, c = x.shape
    if gamma.numel() != c or beta.numel() != c or c % groups != 0:
        raise _lib.FgnHipError('group_norm: bad gamma/beta/groups')
    if residual is not None:
        _chk(residual, 'residual')
        if residual.shape != x.shape:
            raise _lib.FgnHipError('group_norm: bad residual shape')
    lib = _lib.load()
    nbytes = lib.fgn_group_norm_workspace_bytes(n, h * w, c, groups)
    ws = torch.empty(max(nbytes // 8, 1), device=x.device, dtype=torch.float64)
    y = x if inplace else torch.empty_like(x)
    _lib.check(lib.fgn_group_norm_nhwc_f32(_ptr(x), _ptr(y), _ptr(gamma), _ptr(beta), _ptr(residual), _ptr(ws), nbytes,
                                           n, h * w, c, groups, float(eps), int(relu), _stream()),
               'fgn_group_norm_nhwc_f32')
    return y


def avgpool2x2(x: torch.Tensor) -> torch.Tensor:
    """AvgPool2d(2, 2, ceil_mode=True, count_include_pad=False) over NHWC."""
    _chk(x, 'x')
    n, h, w, c = x.shape
    y = torch.empty((n, (h + 1) // 2, (w + 1) // 2, c), device=x.device, dtype=torch.float32)
    _lib.check(_lib.load().fgn_avgpool2x2_nhwc_f32(_ptr(x), _ptr(y), n, h, w, c, _stream()), 'fgn_avgpool2x2_nhwc_f32')
    return y


def roi_align(fmap: torch.Tensor, rois: torch.Tensor, out_size: int, spatial_scale: float,
              sampling_ratio: int, aligned: bool, n_rois_dev: Optional[torch.Tensor] = None,
              post_shift: Optional[torch.Tensor] = None, relu: bool = False,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fmap [B,H,W,C], rois [R,5] -> [R,P,P,C] (+ post_shift[C], ReLU)."""
    _chk(fmap, 'fmap')
    _chk(rois, 'rois')
    b, h, w, c = fmap.shape
    if post_shift is not None:
        _chk(post_shift, 'post_shift')
        if post_shift.numel() != c:
            raise _lib.FgnHipError('roi_align: post_shift must be [C]')
    if rois.dim() != 2 or rois.shape[1] != 5:
        raise _lib.FgnHipError('roi_align: rois must be [R,5]')
    r = rois.shape[0]
    if out is None:
        out = torch.empty((r, out_size, out_size, c), device=fmap.device, dtype=torch.float32)
    else:
        _chk(out, 'out')
        if tuple(out.shape) != (r, out_size, out_size, c):
            raise _lib.FgnHipError('roi_align: bad out shape')
    if n_rois_dev is not None:
        _chk(n_rois_dev, 'n_rois_dev', torch.int32)
    rc = _lib.load().fgn_roi_align_nhwc_f32(_ptr(fmap), _ptr(rois), _ptr(out), _ptr(n_rois_dev), r, b, h, w, c,
                                            out_size, float(spatial_scale), sampling_ratio, int(aligned),
                                            _ptr(post_shift), int(relu), _stream())
    _lib.check(rc, 'fgn_roi_align_nhwc_f32')
    return out


def roi_align2(fmap: torch.Tensor, fmap2: torch.Tensor, rois: torch.Tensor, out_size: int, spatial_scale: float,
               sampling_ratio: int, aligned: bool, n_rois_dev: Optional[torch.Tensor] = None,
               post_shift2: Optional[torch.Tensor] = None, relu2: bool = False):
    """Two maps of one spatial size pooled at the same RoIs by ONE launch: -> ([R,P,P,C], [R,P,P,C2] (+ post_shift2,
    ReLU)); identical bytes to two ``roi_align`` calls."""
    _chk(fmap, 'fmap')
    _chk(fmap2, 'fmap2')
    _chk(rois, 'rois')
    b, h, w, c = fmap.shape
    c2 = fmap2.shape[3]
    if tuple(fmap2.shape[:3]) != (b, h, w) or rois.dim() != 2 or rois.shape[1] != 5:
        raise _lib.FgnHipError('roi_align2: operand shapes inconsistent')
    if post_shift2 is not None:
        _chk(post_shift2, 'post_shift2')
        if post_shift2.numel() != c2:
            raise _lib.FgnHipError('roi_align2: post_shift2 must be [C2]')
    if n_rois_dev is not None:
        _chk(n_rois_dev, 'n_rois_dev', torch.int32)
    r = rois.shape[0]
    out = torch.empty((r, out_size, out_size, c), device=fmap.device, dtype=torch.float32)
    out2 = torch.empty((r, out_size, out_size, c2), device=fmap.device, dtype=torch.float32)
    rc = _lib.load().fgn_roi_align2_nhwc_f32(_ptr(fmap), _ptr(fmap2), _ptr(rois), _ptr(out), _ptr(out2), _ptr(n_rois_dev),
                                             r, b, h, w, c, c2, out_size, float(spatial_scale), sampling_ratio,
                                             int(aligned), _ptr(post_shift2), int(relu2), _stream())
    _lib.check(rc, 'fgn_roi_align2_nhwc_f32')
    return out, out2


def roi_align_mask(mask_u8: torch.Tensor, rois: torch.Tensor, out_size: int, spatial_scale: float,
                   sampling_ratio: int, aligned: bool) -> torch.Tensor:
    """mask [B,H,W] uint8 -> [R,P,P] fp32."""
    _chk(mask_u8, 'mask', torch.uint8)
    _chk(rois, 'rois')
    b, h, w = mask_u8.shape
    r = rois.shape[0]
    out = torch.empty((r, out_size, out_size), device=mask_u8.device, dtype=torch.float32)
    rc = _lib.load().fgn_roi_align_mask_u8(_ptr(mask_u8), _ptr(rois), _ptr(out), r, b, h, w, out_size,
                                           float(spatial_scale), sampling_ratio, int(aligned), _stream())
    _lib.check(rc, 'fgn_roi_align_mask_u8')
    return out


def support_class_vectors(x: torch.Tensor, weights: Optional[torch.Tensor], n_groups: int, k: int) -> torch.Tensor:
    """x [n_groups*k, P.., C] -> [n_groups, C] mean over (k, P) of x * weights."""
    _chk(x, 'x')
    c = x.shape[-1]
    p = x[0].numel() // c
    if x.shape[0] != n_groups * k:
        raise _lib.FgnHipError('support_class_vectors: leading dim != n_groups*k')
    if weights is not None:
        _chk(weights, 'weights')
        if weights.numel() != n_groups * k * p:
            raise _lib.FgnHipError('support_class_vectors: weights shape mismatch')
    out = torch.empty((n_groups, c), device=x.device, dtype=torch.float32)
    rc = _lib.load().fgn_support_class_vectors_f32(_ptr(x), _ptr(weights), _ptr(out), n_groups, k, p, c, _stream())
    _lib.check(rc, 'fgn_support_class_vectors_f32')
    return out


def scale_channels(x: torch.Tensor, v: torch.Tensor, div: int) -> torch.Tensor:
    """x [n_in, ..., C], v [n_in*div, C] -> [n_in*div, ..., C] = x[n // div] * v[n]."""
    _chk(x, 'x')
    _chk(v, 'v')
    n_out, c = v.shape
    if x.shape[0] * div != n_out or x.shape[-1] != c:
        raise _lib.FgnHipError('scale_channels: operand shapes inconsistent')
    out = torch.empty((n_out,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
    p = x[0].numel() // c
    _lib.check(_lib.load().fgn_scale_channels_f32(_ptr(x), _ptr(v), _ptr(out), n_out, div, p, c, _stream()),
               'fgn_scale_channels_f32')
    return out


def support_kmean(x: torch.Tensor, n_groups: int, k: int) -> torch.Tensor:
    _chk(x, 'x')
    if x.shape[0] != n_groups * k:
        raise _lib.FgnHipError('support_kmean: leading dim != n_groups*k')
    c = x.shape[-1]
    p = x[0].numel() // c
    out = torch.empty((n_groups,) + tuple(x.shape[1:]), device=x.device, dtype=torch.float32)
    rc = _lib.load().fgn_support_kmean_f32(_ptr(x), _ptr(out), n_groups, k, p, c, _stream())
    _lib.check(rc, 'fgn_support_kmean_f32')
    return out


def gather_support_vectors(table: torch.Tensor, labels: torch.Tensor, rois: Optional[torch.Tensor],
                           n_ways: int, n_dev: Optional[torch.Tensor] = None) -> torch.Tensor:
    _chk(table, 'table')
    _chk(labels, 'labels', torch.int64)
    n, c = labels.shape[0], table.shape[-1]
    if rois is not None:
        _chk(rois, 'rois')
        if rois.shape[0] < n or rois.shape[1] != 5:
            raise _lib.FgnHipError('gather_support_vectors: rois must be [>=n,5]')
    out = zeros((n, c), table.device)
    rc = _lib.load().fgn_gather_support_vectors_f32(_ptr(table), _ptr(labels), _ptr(rois), _ptr(out), _ptr(n_dev),
                                                    n, n_ways, c, _stream())
    _lib.check(rc, 'fgn_gather_support_vectors_f32')
    return out


def relation_gn_head(q: torch.Tensor, s: torch.Tensor, rois: torch.Tensor, gn_w, gn_b, fc_w, fc_b,
                     n_ways: int, gn_groups: int, eps: float, n_rois_dev: Optional[torch.Tensor] = None,
                     rel_out: Optional[torch.Tensor] = None):
    """q [R,7,7,C]; s [B*N,7,7,C]; rois [R,5] -> cls_raw [R*N,2], reg_raw [R*N,4].  ``rel_out`` (parity tests):
    [R*N,7,7,C] receives the relation feature map the reference materialises (fgn_roi_head.py:274)."""
    for t, nm in ((q, 'Q'), (s, 'S'), (rois, 'rois'), (gn_w, 'gn_w'), (gn_b, 'gn_b'), (fc_w, 'fc_w'), (fc_b, 'fc_b')):
        _chk(t, nm)
    r, p, _, c = q.shape
    if s.shape[1:] != q.shape[1:] or s.shape[0] % n_ways or rois.shape[0] < r or fc_w.shape != (6, c):
        raise _lib.FgnHipError('relation_gn_head: operand shapes inconsistent')
    if rel_out is not None:
        _chk(rel_out, 'rel_out')
        if tuple(rel_out.shape) != (r * n_ways, p, p, c):
            raise _lib.FgnHipError('relation_gn_head: rel_out must be [R*N,P,P,C]')
    cls = zeros((r * n_ways, 2), q.device)
    reg = zeros((r * n_ways, 4), q.device)
    L = _lib.load()
    scratch = torch.empty(L.fgn_relation_gn_head_scratch_bytes(r, n_ways, c), device=q.device, dtype=torch.uint8)
    rc = L.fgn_relation_gn_head_f32(_ptr(q), _ptr(s), _ptr(rois), _ptr(gn_w), _ptr(gn_b), _ptr(fc_w),
                                    _ptr(fc_b), _ptr(cls), _ptr(reg), _ptr(n_rois_dev), r, n_ways, c,
                                    gn_groups, p, float(eps), _ptr(rel_out), _ptr(scratch), _stream())
    _lib.check(rc, 'fgn_relation_gn_head_f32')
    return cls, reg


# --------------------------------------------------------------------------------------
# selection stages
# --------------------------------------------------------------------------------------
MAX_RATIO = float(np.float32(abs(math.log(16.0 / 1000.0))))


def base_anchors(scales, ratios, base_size) -> np.ndarray:
    """mmdet AnchorGenerator base anchors (ratio-major, centre 0, not rounded), fp32."""
    f = np.float32
    w = h = f(base_size)
    ratios = np.asarray(ratios, f)
    scales = np.asarray(scales, f)
    h_ratios = np.sqrt(ratios)
    w_ratios = (f(1) / h_ratios).astype(f)
    ws = (w * w_ratios[:, None] * scales[None, :]).reshape(-1).astype(f)
    hs = (h * h_ratios[:, None] * scales[None, :]).reshape(-1).astype(f)
    c = f(0.0) * w
    return np.stack([c - f(0.5) * ws, c - f(0.5) * hs, c + f(0.5) * ws, c + f(0.5) * hs], -1).astype(f)


def rpn_merge(head: torch.Tensor, batch: int, n_ways: int, n_anchors: int):
    """head [B*N, h, w, CH] -> logits, scores [B, h*w*A], deltas [B, h*w*A, 4]."""
    _chk(head, 'head')
    bn, h, w, ch = head.shape
    if bn != batch * n_ways:
        raise _lib.FgnHipError('rpn_merge: leading dim != batch*n_ways')
    n = h * w * n_anchors
    logits = torch.empty((batch, n), device=head.device, dtype=torch.float32)
    scores = torch.empty_like(logits)
    deltas = torch.empty((batch, n, 4), device=head.device, dtype=torch.float32)
    rc = _lib.load().fgn_rpn_merge_f32(_ptr(head), _ptr(logits), _ptr(scores), _ptr(deltas), batch, n_ways, h * w,
                                       n_anchors, ch, _stream())
    _lib.check(rc, 'fgn_rpn_merge_f32')
    return logits, scores, deltas


def rpn_proposals(scores: torch.Tensor, deltas: torch.Tensor, anchors_base: torch.Tensor, feat_h: int, feat_w: int,
                  stride: int, img_h: int, img_w: int, means, stds, nms_pre: int, min_bbox_size: float,
                  iou_thr: float, max_per_img: int, debug_topk: bool = False, with_rois: bool = False):
    """-> proposals [B,max_per_img,5] (x1,y1,x2,y2,score), n_props [B] int32 (+ with_rois: the same boxes as
    [B*max_per_img,5] RoIs (image index, box) = bbox2roi, fgn_roi_head.py:556)."""
    _chk(scores, 'scores')
    _chk(deltas, 'deltas')
    _chk(anchors_base, 'anchors_base')
    batch, n_total = scores.shape
    a = anchors_base.shape[0]
    if n_total != feat_h * feat_w * a or tuple(deltas.shape) != (batch, n_total, 4):
        raise _lib.FgnHipError('rpn_proposals: operand shapes inconsistent')
    L = _lib.load()
    props = torch.empty((batch, max_per_img, 5), device=scores.device, dtype=torch.float32)
    rois = torch.empty((batch * max_per_img, 5), device=scores.device, dtype=torch.float32)   # bbox2roi of the proposals
    n_props = zeros((batch,), scores.device, torch.int32)
    n_sel = nms_pre if 0 < nms_pre < n_total else n_total
    if (n_sel > 8192 or max_per_img > 1024) and not debug_topk:
        # training sizes (train_cfg.rpn_proposal: 12000 -> 2000): the ranking runs as a global sort over many workgroups
        scratch = torch.empty(L.fgn_rpn_proposals_large_scratch_bytes(batch, n_total, nms_pre), device=scores.device,
                              dtype=torch.uint8)
        rc = L.fgn_rpn_proposals_large_f32(_ptr(scores), _ptr(deltas), _ptr(anchors_base), _ptr(scratch), _ptr(props),
                                           _ptr(rois), _ptr(n_props), batch, feat_h, feat_w, a, stride, float(img_h),
                                           float(img_w), _f4(means), _f4(stds), MAX_RATIO, nms_pre,
                                           float(min_bbox_size), float(iou_thr), max_per_img, _stream())
        _lib.check(rc, 'fgn_rpn_proposals_large_f32')
        return (props, n_props, rois) if with_rois else (props, n_props)
    scratch = torch.empty(L.fgn_rpn_proposals_scratch_bytes(batch, n_total, nms_pre), device=scores.device,
                          dtype=torch.uint8)
    dbg = None
    if debug_topk:
        dbg = torch.full((batch, 8192), -1, device=scores.device, dtype=torch.int32)
    # zero-filled workspace of the multi-workgroup pre-selection (histograms, counters): from the episode's zero arena
    pre_zeroed = zeros((L.fgn_rpn_proposals_zeroed_bytes(batch),), scores.device, torch.uint8)
    rc = L.fgn_rpn_proposals_f32(_ptr(scores), _ptr(deltas), _ptr(anchors_base), _ptr(scratch), _ptr(pre_zeroed),
                                 _ptr(props), _ptr(rois), _ptr(n_props), _ptr(dbg), batch, feat_h, feat_w, a, stride,
                                 float(img_h),
                                 float(img_w), _f4(means), _f4(stds), MAX_RATIO, nms_pre, float(min_bbox_size),
                                 float(iou_thr), max_per_img, _stream())
    _lib.check(rc, 'fgn_rpn_proposals_f32')
    if debug_topk:
        return props, n_props, dbg
    if with_rois:
        return props, n_props, rois
    return props, n_props


def det_post(rois: torch.Tensor, cls_raw: torch.Tensor, reg_raw: torch.Tensor, n_ways: int, img_h: int, img_w: int,
             means, stds, score_thr: float, iou_thr: float, max_per_img: int,
             n_rois_dev: Optional[torch.Tensor] = None, debug_scores: bool = False, img_index: Optional[int] = None,
             batch: int = 1, out=None):
    """-> det [max_per_img,5], labels, n_det (+ with ``img_index``: mask RoIs [max_per_img,5] = (img_index, box), the
    mask branch's bbox2roi, fgn_roi_head.py:654).  ``batch`` > 1: the RoIs of ``batch`` images stacked ([batch*R,5],
    ``n_rois_dev`` [batch] when given), one workgroup per image in one launch; outputs stacked the same way
    ([batch*max_per_img,5], ..., n_det [batch]), image i carrying index ``img_index + i``.  ``out``: caller-owned
    (det, labels, n_det) of those shapes, n_det ZERO on entry (the packed result record of the detector)."""
    for t, nm in ((rois, 'rois'), (cls_raw, 'cls_raw'), (reg_raw, 'reg_raw')):
        _chk(t, nm)
    if batch < 1 or rois.shape[0] % batch:
        raise _lib.FgnHipError('det_post: the RoI rows are not a multiple of batch')
    r = rois.shape[0] // batch
    if rois.shape[1] != 5 or tuple(cls_raw.shape) != (batch * r * n_ways, 2) or \
            tuple(reg_raw.shape) != (batch * r * n_ways, 4):
        raise _lib.FgnHipError('det_post: operand shapes inconsistent')
    if n_rois_dev is not None and n_rois_dev.numel() < batch:
        raise _lib.FgnHipError('det_post: n_rois_dev holds fewer counts than batch')
    L = _lib.load()
    scratch = torch.empty(batch * L.fgn_det_post_scratch_bytes(r, n_ways), device=rois.device, dtype=torch.uint8)
    if out is not None:
        det, lab, n_det = out
        _chk(det, 'out det'); _chk(lab, 'out labels', torch.int64); _chk(n_det, 'out n_det', torch.int32)
        if tuple(det.shape) != (batch * max_per_img, 5) or tuple(lab.shape) != (batch * max_per_img,) or n_det.numel() != batch:
            raise _lib.FgnHipError('det_post: bad out shapes')
    else:
        det = torch.empty((batch * max_per_img, 5), device=rois.device, dtype=torch.float32)
        lab = torch.empty((batch * max_per_img,), device=rois.device, dtype=torch.int64)
        n_det = zeros((batch,), rois.device, torch.int32)
    mrois = torch.empty((batch * max_per_img, 5), device=rois.device, dtype=torch.float32) \
        if img_index is not None else None
    # softmax scores [r, n_ways + 1] of the first image followed by 16 words of phase stamps (tools/post_time.py)
    dbg = torch.zeros((r * (n_ways + 1) + 16,), device=rois.device, dtype=torch.float32) if debug_scores else None
    rc = L.fgn_det_post_f32(_ptr(rois), _ptr(cls_raw), _ptr(reg_raw), _ptr(n_rois_dev), _ptr(scratch), _ptr(det),
                            _ptr(mrois), int(img_index or 0), _ptr(lab), _ptr(n_det), _ptr(dbg), batch, r, n_ways,
                            float(img_h), float(img_w), _f4(means),
                            _f4(stds), MAX_RATIO, float(score_thr), float(iou_thr), max_per_img, _stream())
    _lib.check(rc, 'fgn_det_post_f32')
    if debug_scores:
        det_post.last_stamps = dbg[r * (n_ways + 1):].view(torch.int32)
        return det, lab, n_det, dbg[:r * (n_ways + 1)].view(r, n_ways + 1)
    if img_index is not None:
        return det, lab, n_det, mrois
    return det, lab, n_det


def mask_logits(x: torch.Tensor, w: torch.Tensor, bias, roi_size: int,
                n_dev: Optional[torch.Tensor] = None, prob_out: Optional[torch.Tensor] = None):
    """x [D, P, P, 4*C] (deconv output, sub-position major) -> logits, prob [D, 2P, 2P].  ``bias``: a float, or a
    one-element device tensor the kernel reads itself (no host read of a parameter a training step has just updated)."""
    _chk(x, 'x')
    _chk(w, 'w')
    d = x.shape[0]
    c = w.numel()
    if x[0].numel() != roi_size * roi_size * 4 * c:
        raise _lib.FgnHipError('mask_logits: x shape inconsistent with weight')
    logits = zeros((d, 2 * roi_size, 2 * roi_size), x.device)
    if prob_out is not None:        # caller-owned, ZERO on entry (rows beyond the device count are not written)
        _chk(prob_out, 'prob_out')
        if tuple(prob_out.shape) != (d, 2 * roi_size, 2 * roi_size):
            raise _lib.FgnHipError('mask_logits: bad prob_out shape')
        prob = prob_out
    else:
        prob = zeros((d, 2 * roi_size, 2 * roi_size), x.device)
    bias_dev = None
    if isinstance(bias, torch.Tensor):
        _chk(bias, 'bias')
        bias_dev, bias = bias, 0.0
    rc = _lib.load().fgn_mask_logits_f32(_ptr(x), _ptr(w), float(bias), _ptr(bias_dev), _ptr(logits), _ptr(prob),
                                         _ptr(n_dev), d, roi_size, c, _stream())
    _lib.check(rc, 'fgn_mask_logits_f32')
    return logits, prob


def mask_paste(prob: torch.Tensor, boxes: torch.Tensor, img_h: int, img_w: int, thr: float,
               n_dev: Optional[torch.Tensor] = None, skip_empty: bool = True) -> torch.Tensor:
    """prob [D, M, M], boxes [D, >=4] (x1,y1,x2,y2 first) -> uint8 [D, H, W].  ``skip_empty``: mmdet's CPU paste
    (inside the integer-expanded box) / False: its CUDA paste (grid over the whole image); equal for thr >= 0.5."""
    _chk(prob, 'prob')
    _chk(boxes, 'boxes')
    d, m, _ = prob.shape
    if boxes.shape[0] != d or boxes.shape[1] < 4:
        raise _lib.FgnHipError('mask_paste: boxes shape mismatch')
    out = torch.empty((d, img_h, img_w), device=prob.device, dtype=torch.uint8)
    rc = _lib.load().fgn_mask_paste_u8(_ptr(prob), _ptr(boxes), boxes.shape[1], _ptr(out), _ptr(n_dev), d, img_h,
                                       img_w, m, float(thr), int(bool(skip_empty)), _stream())
    _lib.check(rc, 'fgn_mask_paste_u8')
    return out


RLE_TRANS_CAP = 16384     # transitions per detection kept on device
RLE_BYTE_CAP = 16384      # COCO string bytes per detection


def mask_rle(prob: torch.Tensor, boxes: torch.Tensor, img_h: int, img_w: int, thr: float,
             n_dev: Optional[torch.Tensor] = None, skip_empty: bool = True, out=None):
    """Fused paste + threshold + COCO RLE.  Returns (bytes [D,RLE_BYTE_CAP] u8, lens [D] i32,
    overflow [D] i32), all on device.  ``out``: caller-owned tensors of those shapes, lens / overflow ZERO on entry."""
    _chk(prob, 'prob')
    _chk(boxes, 'boxes')
    d, m, _ = prob.shape
    if boxes.shape[0] != d or boxes.shape[1] < 4:
        raise _lib.FgnHipError('mask_rle: boxes shape mismatch')
    dev = prob.device
    scratch = torch.empty((d, RLE_TRANS_CAP), device=dev, dtype=torch.int32)
    if out is not None:
        out, lens, ovf = out
        _chk(out, 'out bytes', torch.uint8); _chk(lens, 'out lens', torch.int32); _chk(ovf, 'out overflow', torch.int32)
        if tuple(out.shape) != (d, RLE_BYTE_CAP) or lens.numel() != d or ovf.numel() != d:
            raise _lib.FgnHipError('mask_rle: bad out shapes')
    else:
        out = torch.empty((d, RLE_BYTE_CAP), device=dev, dtype=torch.uint8)
        lens = zeros((d,), dev, torch.int32)
        ovf = zeros((d,), dev, torch.int32)
    rc = _lib.load().fgn_mask_rle(_ptr(prob), _ptr(boxes), boxes.shape[1], _ptr(scratch), _ptr(out), _ptr(lens),
                                  _ptr(ovf), _ptr(n_dev), d, img_h, img_w, m, float(thr), RLE_TRANS_CAP,
                                  RLE_BYTE_CAP, int(bool(skip_empty)), _stream())
    _lib.check(rc, 'fgn_mask_rle')
    return out, lens, ovf


def dense_mask_rle(masks: torch.Tensor, packed: bool = False):
    """COCO RLE of dense binary masks [n,H,W] (bool / uint8) on the device: the ground-truth masks of the query
    (``qry_isegmaps_rle``, fgn.py:298).  Returns (bytes [n,RLE_BYTE_CAP] u8, lens [n] i32, overflow [n] i32).
    ``packed``: the three are views of ONE allocation laid out [lens | overflow | bytes] (one device-to-host copy
    moves them all) and that allocation is returned as a fourth value."""
    if masks.dtype == torch.bool:
        masks = masks.view(torch.uint8)
    _chk(masks, 'masks', torch.uint8)
    if masks.dim() != 3:
        raise _lib.FgnHipError('dense_mask_rle: masks must be [n,H,W]')
    n, h, w = masks.shape
    dev = masks.device
    buf = None
    if packed:
        buf = torch.empty(n * (RLE_BYTE_CAP + 8), device=dev, dtype=torch.uint8)
        buf[:n * 8].zero_()
        lens, ovf = buf[:n * 4].view(torch.int32), buf[n * 4:n * 8].view(torch.int32)
        out = buf[n * 8:].view(n, RLE_BYTE_CAP)
    else:
        out = torch.empty((n, RLE_BYTE_CAP), device=dev, dtype=torch.uint8)
        lens = torch.zeros((n,), device=dev, dtype=torch.int32)
        ovf = torch.zeros((n,), device=dev, dtype=torch.int32)
    if n == 0:
        return (out, lens, ovf, buf) if packed else (out, lens, ovf)
    L = _lib.load()
    nbytes = L.fgn_dense_rle_scratch_bytes(n, h, w, RLE_TRANS_CAP)
    scratch = torch.empty(nbytes, device=dev, dtype=torch.uint8)
    _lib.check(L.fgn_dense_mask_rle(_ptr(masks), _ptr(scratch), nbytes, _ptr(out), _ptr(lens), _ptr(ovf), n, h, w,
                                    RLE_TRANS_CAP, RLE_BYTE_CAP, _stream()), 'fgn_dense_mask_rle')
    return (out, lens, ovf, buf) if packed else (out, lens, ovf)


# --------------------------------------------------------------------------------------
# forward_train pieces (csrc/train.hip)
# --------------------------------------------------------------------------------------
def box_assign(boxes: torch.Tensor, gts: torch.Tensor, pos_iou_thr: float, neg_iou_thr: float, min_pos_iou: float,
               match_low_quality: bool = True, inside: Optional[torch.Tensor] = None, with_overlaps: bool = False,
               out: Optional[torch.Tensor] = None):
    """MaxIoUAssigner: boxes [n, >=4], gts [k,4] -> gt_inds [n] int32 (-2 not a candidate, -1 ignored, 0 negative,
    i+1 positive of GT i) (+ max_overlaps [n]).  ``out``: an int32 [n] slice to write into."""
    _chk(boxes, 'boxes')
    n, k = boxes.shape[0], gts.shape[0]
    if k:
        _chk(gts, 'gts')
    if inside is not None:
        _chk(inside, 'inside', torch.uint8)
    L = _lib.load()
    if out is None:
        gt_inds = torch.empty((n,), device=boxes.device, dtype=torch.int32)
    else:
        _chk(out, 'out', torch.int32)
        if out.numel() != n:
            raise _lib.FgnHipError('box_assign: out must hold n entries')
        gt_inds = out
    mo = torch.zeros((n,), device=boxes.device, dtype=torch.float32) if with_overlaps else None
    scratch = torch.empty(L.fgn_box_assign_scratch_bytes(n, k), device=boxes.device, dtype=torch.uint8)
    rc = L.fgn_box_assign_f32(_ptr(boxes), boxes.shape[1], _ptr(inside), _ptr(gts) if k else None, n, k,
                              float(pos_iou_thr), float(neg_iou_thr), float(min_pos_iou), int(match_low_quality),
                              _ptr(scratch), _ptr(gt_inds), _ptr(mo), _stream())
    _lib.check(rc, 'fgn_box_assign_f32')
    return (gt_inds, mo) if with_overlaps else gt_inds


def bbox2delta(proposals: torch.Tensor, gts: torch.Tensor, means, stds) -> torch.Tensor:
    _chk(proposals, 'proposals')
    _chk(gts, 'gts')
    n = proposals.shape[0]
    if tuple(proposals.shape) != (n, 4) or tuple(gts.shape) != (n, 4):
        raise _lib.FgnHipError('bbox2delta: operands must be [n,4]')
    out = torch.empty((n, 4), device=proposals.device, dtype=torch.float32)
    rc = _lib.load().fgn_bbox2delta_f32(_ptr(proposals), _ptr(gts), _ptr(out), n, _f4(means), _f4(stds), _stream())
    _lib.check(rc, 'fgn_bbox2delta_f32')
    return out


def _loss_args(x, y, w):
    _chk(x, 'pred')
    _chk(y, 'target')
    if x.numel() != y.numel():
        raise _lib.FgnHipError('loss: pred / target sizes differ')
    if w is not None:
        _chk(w, 'weight')
        if w.numel() != x.numel():
            raise _lib.FgnHipError('loss: weight size differs')


def bce_logits_sum(x, y, w, avg_factor: float, y_threshold: float = -1.0) -> torch.Tensor:
    """sum_i w_i * BCEWithLogits(x_i, y_i) / avg_factor -> [1] (y binarised at y_threshold when >= 0)."""
    _loss_args(x, y, w)
    out = torch.empty((1,), device=x.device, dtype=torch.float32)
    rc = _lib.load().fgn_bce_logits_sum_f32(_ptr(x), _ptr(y), _ptr(w), x.numel(), float(y_threshold),
                                            float(avg_factor), _ptr(out), _stream())
    _lib.check(rc, 'fgn_bce_logits_sum_f32')
    return out


def smooth_l1_sum(pred, target, w, avg_factor: float, beta: float = 1.0) -> torch.Tensor:
    _loss_args(pred, target, w)
    out = torch.empty((1,), device=pred.device, dtype=torch.float32)
    rc = _lib.load().fgn_smooth_l1_sum_f32(_ptr(pred), _ptr(target), _ptr(w), pred.numel(), float(beta),
                                           float(avg_factor), _ptr(out), _stream())
    _lib.check(rc, 'fgn_smooth_l1_sum_f32')
    return out


def softmax_ce_sum(logits, labels, w, avg_factor: float) -> torch.Tensor:
    _chk(logits, 'logits')
    _chk(labels, 'labels', torch.int64)
    n, c = logits.shape
    if labels.numel() != n or (w is not None and w.numel() != n):
        raise _lib.FgnHipError('softmax_ce_sum: operand sizes differ')
    if w is not None:
        _chk(w, 'weight')
    out = torch.empty((1,), device=logits.device, dtype=torch.float32)
    rc = _lib.load().fgn_softmax_ce_sum_f32(_ptr(logits), _ptr(labels), _ptr(w), n, c, float(avg_factor), _ptr(out),
                                            _stream())
    _lib.check(rc, 'fgn_softmax_ce_sum_f32')
    return out


def bn_train(x: torch.Tensor, gamma, beta, eps: float, momentum: float, running_mean=None, running_var=None,
             residual=None, relu: bool = False, inplace: bool = True):
    """BatchNorm2d in training mode on NHWC x [..., C]: -> (y, batch mean [C], biased batch variance [C]); the
    running estimates (optional) are updated in place."""
    _chk(x, 'x')
    c = x.shape[-1]
    p = x.numel() // c
    for t, nm in ((gamma, 'gamma'), (beta, 'beta'), (running_mean, 'running_mean'), (running_var, 'running_var'),
                  (residual, 'residual')):
        if t is not None:
            _chk(t, nm)
    if residual is not None and residual.numel() != x.numel():
        raise _lib.FgnHipError('bn_train: residual shape differs')
    L = _lib.load()
    mean = torch.empty((c,), device=x.device, dtype=torch.float32)
    var = torch.empty((c,), device=x.device, dtype=torch.float32)
    out = x if inplace else torch.empty_like(x)
    scratch = torch.empty(L.fgn_bn_train_scratch_bytes(c), device=x.device, dtype=torch.uint8)
    rc = L.fgn_bn_train_f32(_ptr(x), p, c, _ptr(gamma), _ptr(beta), float(eps), float(momentum), _ptr(running_mean),
                            _ptr(running_var), _ptr(residual), int(relu), _ptr(scratch), _ptr(mean), _ptr(var),
                            _ptr(out), _stream())
    _lib.check(rc, 'fgn_bn_train_f32')
    return out, mean, var


# --------------------------------------------------------------------------------------
# backward pieces of the trainable heads (csrc/train_bwd.hip)
# --------------------------------------------------------------------------------------
def bce_logits_grad(x, y, w, scale: float, y_threshold: float = -1.0) -> torch.Tensor:
    _loss_args(x, y, w)
    dx = torch.empty_like(x)
    rc = _lib.load().fgn_bce_logits_grad_f32(_ptr(x), _ptr(y), _ptr(w), x.numel(), float(y_threshold), float(scale),
                                             _ptr(dx), _stream())
    _lib.check(rc, 'fgn_bce_logits_grad_f32')
    return dx


def smooth_l1_grad(pred, target, w, scale: float, beta: float = 1.0) -> torch.Tensor:
    _loss_args(pred, target, w)
    d = torch.empty_like(pred)
    rc = _lib.load().fgn_smooth_l1_grad_f32(_ptr(pred), _ptr(target), _ptr(w), pred.numel(), float(beta), float(scale),
                                            _ptr(d), _stream())
    _lib.check(rc, 'fgn_smooth_l1_grad_f32')
    return d


def softmax_ce_grad(logits, labels, w, scale: float) -> torch.Tensor:
    _chk(logits, 'logits')
    _chk(labels, 'labels', torch.int64)
    if w is not None:
        _chk(w, 'weight')
    n, c = logits.shape
    d = torch.empty_like(logits)
    rc = _lib.load().fgn_softmax_ce_grad_f32(_ptr(logits), _ptr(labels), _ptr(w), n, c, float(scale), _ptr(d), _stream())
    _lib.check(rc, 'fgn_softmax_ce_grad_f32')
    return d


def relu_backward(dy: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """dy * [y > 0] (y: the ReLU's output)."""
    _chk(dy, 'dy')
    _chk(y, 'y')
    if dy.numel() != y.numel():
        raise _lib.FgnHipError('relu_backward: operand sizes differ')
    out = torch.empty_like(dy)
    rc = _lib.load().fgn_relu_backward_f32(_ptr(dy), _ptr(y), _ptr(out), dy.numel(), _stream())
    _lib.check(rc, 'fgn_relu_backward_f32')
    return out


def colsum(x: torch.Tensor, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """x [..., C] -> [C] = sum over all leading dims (fp64 partials, fixed order)."""
    _chk(x, 'x')
    c = x.shape[-1]
    r = x.numel() // c if c else 0
    L = _lib.load()
    if out is None:
        out = torch.empty((c,), device=x.device, dtype=torch.float32)
        accumulate = False
    else:
        _chk(out, 'out')
    scratch = torch.empty(L.fgn_colsum_scratch_bytes(c), device=x.device, dtype=torch.uint8)
    rc = L.fgn_colsum_f32(_ptr(x), r, c, _ptr(scratch), _ptr(out), int(accumulate), _stream())
    _lib.check(rc, 'fgn_colsum_f32')
    return out


def bn_train_backward(x_pre, y_post, dy, mean, var, gamma, eps: float, want_g: bool = False):
    """-> dx, dgamma, dbeta (+ g = dy masked by the ReLU, when ``want_g``)."""
    for t, nm in ((x_pre, 'x_pre'), (dy, 'dy'), (mean, 'mean'), (var, 'var'), (gamma, 'gamma')):
        _chk(t, nm)
    if y_post is not None:
        _chk(y_post, 'y_post')
    c = x_pre.shape[-1]
    p = x_pre.numel() // c
    if dy.numel() != x_pre.numel() or (y_post is not None and y_post.numel() != x_pre.numel()):
        raise _lib.FgnHipError('bn_train_backward: operand sizes differ')
    L = _lib.load()
    dx = torch.empty_like(x_pre)
    g = torch.empty_like(x_pre) if want_g else None
    dgamma = torch.empty((c,), device=x_pre.device, dtype=torch.float32)
    dbeta = torch.empty((c,), device=x_pre.device, dtype=torch.float32)
    scratch = torch.empty(L.fgn_bn_train_backward_scratch_bytes(c), device=x_pre.device, dtype=torch.uint8)
    rc = L.fgn_bn_train_backward_f32(_ptr(x_pre), _ptr(y_post), _ptr(dy), _ptr(mean), _ptr(var), _ptr(gamma), float(eps),
                                     p, c, _ptr(scratch), _ptr(dx), _ptr(g), _ptr(dgamma), _ptr(dbeta), _stream())
    _lib.check(rc, 'fgn_bn_train_backward_f32')
    return (dx, dgamma, dbeta, g) if want_g else (dx, dgamma, dbeta)


def relation_gn_head_backward(q, s, rois, gn_w, gn_b, fc_w, d_out6, n_ways: int, gn_groups: int, eps: float):
    """-> dQ [R,7,7,C], dZ [R*N,7,7,C], pooled [R*N,C], dgamma [C], dbeta [C]."""
    for t, nm in ((q, 'Q'), (s, 'S'), (rois, 'rois'), (gn_w, 'gn_w'), (gn_b, 'gn_b'), (fc_w, 'fc_w'), (d_out6, 'd_out6')):
        _chk(t, nm)
    r, ps, _, c = q.shape
    if tuple(d_out6.shape) != (r * n_ways, 6) or tuple(fc_w.shape) != (6, c):
        raise _lib.FgnHipError('relation_gn_head_backward: operand shapes inconsistent')
    dq = torch.empty_like(q)
    dz = torch.empty((r * n_ways, ps, ps, c), device=q.device, dtype=torch.float32)
    pooled = torch.empty((r * n_ways, c), device=q.device, dtype=torch.float32)
    dga = torch.empty((r, c), device=q.device, dtype=torch.float32)
    dbe = torch.empty((r, c), device=q.device, dtype=torch.float32)
    rc = _lib.load().fgn_relation_gn_head_backward_f32(_ptr(q), _ptr(s), _ptr(rois), _ptr(gn_w), _ptr(gn_b), _ptr(fc_w),
                                                       _ptr(d_out6), _ptr(dq), _ptr(dz), _ptr(pooled), _ptr(dga),
                                                       _ptr(dbe), r, n_ways, c, gn_groups, ps, float(eps), _stream())
    _lib.check(rc, 'fgn_relation_gn_head_backward_f32')
    return dq, dz, pooled, colsum(dga), colsum(dbe)


def mask_logits_backward(up: torch.Tensor, dlogit: torch.Tensor, w: torch.Tensor, roi_size: int):
    """up [D,P,P,4*C], dlogit [D,2P,2P], w [C] -> d_up [D,P,P,4*C], dw [C]."""
    _chk(up, 'up')
    _chk(dlogit, 'dlogit')
    _chk(w, 'w')
    d, c = up.shape[0], w.numel()
    if up[0].numel() != roi_size * roi_size * 4 * c or tuple(dlogit.shape) != (d, 2 * roi_size, 2 * roi_size):
        raise _lib.FgnHipError('mask_logits_backward: operand shapes inconsistent')
    d_up = torch.empty_like(up)
    part = torch.empty((d, c), device=up.device, dtype=torch.float32)
    rc = _lib.load().fgn_mask_logits_backward_f32(_ptr(up), _ptr(dlogit), _ptr(w), _ptr(d_up), _ptr(part), d, roi_size, c,
                                                  _stream())
    _lib.check(rc, 'fgn_mask_logits_backward_f32')
    return d_up, colsum(part)


def im2col3x3(x: torch.Tensor) -> torch.Tensor:
    """x [n,H,W,C] -> [n*H*W, 9*C] (3x3, stride 1, pad 1; column = tap*C + ci)."""
    _chk(x, 'x')
    n, h, w, c = x.shape
    out = torch.empty((n * h * w, 9 * c), device=x.device, dtype=torch.float32)
    rc = _lib.load().fgn_im2col3x3_f32(_ptr(x), _ptr(out), n, h, w, c, _stream())
    _lib.check(rc, 'fgn_im2col3x3_f32')
    return out


def gemm_tn(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a [R,M], b [R,N] -> a^T b [M,N] (the weight-gradient product: reduction over the rows) on the MFMA kernel."""
    _chk(a, 'a')
    _chk(b, 'b')
    r, m = a.shape
    n = b.shape[1]
    if b.shape[0] != r:
        raise _lib.FgnHipError('gemm_tn: operands must share the row count')
    L = _lib.load()
    out = torch.empty((m, n), device=a.device, dtype=torch.float32)
    wsb = L.fgn_gemm_tn_workspace_bytes(r, m, n)
    ws = torch.empty(wsb, device=a.device, dtype=torch.uint8) if wsb else None
    rc = L.fgn_gemm_tn_f32(_ptr(a), _ptr(b), _ptr(out), r, m, n, _ptr(ws), _stream())
    _lib.check(rc, 'fgn_gemm_tn_f32')
    return out


def gemm_small(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False) -> torch.Tensor:
    """a [M,K] (or [K,M] with ``trans_a``) x b [K,N] -> [M,N] on the one-thread-per-output kernel: the products whose
    shapes the MFMA kernels do not take (a dimension that is not a multiple of 4 / 32).  Operands may be row-strided
    2-D views (unit stride along the last dimension)."""
    for t, nm in ((a, 'a'), (b, 'b')):
        if not t.is_cuda or t.dtype != torch.float32 or t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
            raise _lib.FgnHipError(f'gemm_small: {nm} must be a 2-D fp32 device tensor with unit inner stride')
    k, m = (a.shape[0], a.shape[1]) if trans_a else (a.shape[1], a.shape[0])
    if b.shape[0] != k:
        raise _lib.FgnHipError('gemm_small: inner dimensions differ')
    n = b.shape[1]
    out = torch.empty((m, n), device=a.device, dtype=torch.float32)
    lda = a.stride(0) if a.shape[0] > 1 else a.shape[1]
    ldb = b.stride(0) if b.shape[0] > 1 else n
    rc = _lib.load().fgn_gemm_small_f32(_ptr(a), _ptr(b), _ptr(out), m, n, k, lda, ldb, n, int(trans_a), _stream())
    _lib.check(rc, 'fgn_gemm_small_f32')
    return out


def adagrad_multi(params, grads, states, lrs, weight_decay: float, eps: float = 1e-10, table: Optional[dict] = None) -> None:
    """``adagrad_step`` for a list of tensors in one launch (each with its own learning rate).  ``table``: a dict the
    caller keeps between steps - the checked parameter / state pointer arrays are built once per (tensors, lrs) and only
    the gradient pointers are refreshed (the parameters of a ``Trainer`` never move)."""
    import ctypes as C
    k = len(params)
    if not (len(grads) == len(states) == len(lrs) == k):
        raise _lib.FgnHipError('adagrad_multi: list lengths differ')
    if not k:
        return
    key = (tuple(p.data_ptr() for p in params), tuple(s.data_ptr() for s in states), tuple(float(v) for v in lrs))
    if table is None or table.get('key') != key:
        for p, s in zip(params, states):
            _chk(p, 'param'); _chk(s, 'state')
            if not (p.is_contiguous() and s.is_contiguous()) or s.numel() != p.numel():
                raise _lib.FgnHipError('adagrad_multi: parameters and states must be contiguous and of equal size')
        built = dict(key=key, pa=(C.c_void_p * k)(*key[0]), sa=(C.c_void_p * k)(*key[1]),
                     na=(C.c_longlong * k)(*[p.numel() for p in params]), la=(C.c_float * k)(*key[2]),
                     ga=(C.c_void_p * k)(), numel=[p.numel() for p in params])
        if table is None:
            table = built
        else:
            table.clear(); table.update(built)
    ga = table['ga']
    for i, g in enumerate(grads):
        if not g.is_cuda or g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != table['numel'][i]:
            raise _lib.FgnHipError('adagrad_multi: gradients must be contiguous fp32 device tensors of the parameters\' sizes')
        ga[i] = g.data_ptr()
    rc = _lib.load().fgn_adagrad_multi_f32(table['pa'], ga, table['sa'], table['na'], table['la'], k, float(weight_decay),
                                           float(eps), _stream())
    _lib.check(rc, 'fgn_adagrad_multi_f32')


def adagrad_step(param: torch.Tensor, grad: torch.Tensor, state: torch.Tensor, lr: float, weight_decay: float,
                 eps: float = 1e-10) -> None:
    for t, nm in ((param, 'param'), (grad, 'grad'), (state, 'state')):
        _chk(t, nm)
    if grad.numel() != param.numel() or state.numel() != param.numel():
        raise _lib.FgnHipError('adagrad_step: operand sizes differ')
    rc = _lib.load().fgn_adagrad_step_f32(_ptr(param), _ptr(grad), _ptr(state), param.numel(), float(lr),
                                          float(weight_decay), float(eps), _stream())
    _lib.check(rc, 'fgn_adagrad_step_f32')
