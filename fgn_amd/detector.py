"""FGN detector: the drop-in boundary of the hot path.

Mirrors the reference's detector API (subprojects/sp02_omniiseg_fgn_mmdet/fgn.py:28-303):
``FGN(n_ways, k_shots, backbone=..., rpn_head=..., roi_head=..., train_cfg=..., test_cfg=...)``,
``forward(return_loss, **batch)``, ``simple_test(**batch, rescale=True)`` with the
FewShotISEG batch dict in and the per-image result dicts out (same keys, numpy values,
YXYX boxes, COCO RLE masks).  The constructor accepts the reference's mmcv-style config
dicts (fgn_r50_c4_densecl.py:13-186) unchanged.

Host code is Python/PyTorch plumbing (device memory, streams).  Every arithmetic step
runs in libfgn_hip.so (hand-written gfx950 HIP kernels, include/fgn_hip.h); there is no
eager/CPU fallback: without the library or without a GPU ``simple_test`` raises.

Data layout in HBM: all feature maps NHWC fp32; RoI tensors [R,7,7,C] (a RoI is an
"image" of the conv kernel); weights pre-packed [CoutPad][KH][KW][Cin]; eval-mode BN
folded into a per-channel scale/shift epilogue.  Data-dependent counts (proposals,
detections) stay in device int32 counters consumed by the kernels, so an episode runs
without host synchronisation until the final device-to-host copy of the results.
"""
from __future__ import annotations

import copy
import os
from collections import OrderedDict
from typing import Dict, List, Optional

import numpy as np
import torch

from . import config as _config
from . import ops, rle
from .weights import backbone_state_from_pretrained, init_state_dict


# ------------------------------------------------------------------------------------------
# reference-style (mmcv) config -> flat config
# ------------------------------------------------------------------------------------------
_R50_BLOCKS = (3, 4, 6, 3)


def normalise_config(n_ways, k_shots, backbone=None, rpn_head=None, roi_head=None, test_cfg=None,
                     train_cfg=None) -> dict:
    """Accept either this package's flat dicts or the reference's mmcv config dicts."""
    cfg = _config.fgn_r50_c4_config(n_ways, k_shots)
    if backbone:
        if 'stage_blocks' in backbone:
            cfg['backbone'].update(backbone)
        else:   # mmdet ResNet dict (fgn_r50_c4_densecl.py:15-42); layer4 is dropped (main.py:403-405)
            depth = backbone.get('depth', 50)
            if depth not in (18, 50):
                raise NotImplementedError('ResNet-50-C4 (both reference configs) and the ResNet-18 extension are built')
            last = max(backbone.get('out_indices', (2,)))
            norm = backbone.get('norm_cfg', dict(type='BN'))
            if norm.get('type', 'BN') not in ('BN', 'GN'):
                raise NotImplementedError(f"norm_cfg {norm.get('type')!r}")
            if depth == 18 and (norm.get('type', 'BN') != 'BN' or backbone.get('deep_stem') or backbone.get('avg_down')):
                raise NotImplementedError('ResNet-18 extension: frozen-BN BasicBlock stages only')
            cfg['backbone'].update(depth=depth, block='basic' if depth == 18 else 'bottleneck')
            cfg['backbone'].update(stage_blocks=((2, 2, 2, 2) if depth == 18 else _R50_BLOCKS)[:last + 1],
                                   stage_planes=(64, 128, 256, 512)[:last + 1],
                                   strides=tuple(backbone.get('strides', (1, 2, 2, 2)))[:last + 1],
                                   # fgn_r50_c4_scratch.py:16-23
                                   deep_stem=bool(backbone.get('deep_stem', False)),
                                   avg_down=bool(backbone.get('avg_down', False)),
                                   norm=norm.get('type', 'BN'), gn_groups=norm.get('num_groups', 32),
                                   ref_num_stages=int(backbone.get('num_stages', 4)))
    if rpn_head:
        r = cfg['rpn_head']
        if 'anchor_generator' in rpn_head:
            ag, bc = rpn_head['anchor_generator'], rpn_head.get('bbox_coder', {})
            r.update(in_channels=rpn_head.get('in_channels', r['in_channels']),
                     feat_channels=rpn_head.get('feat_channels', r['feat_channels']),
                     anchor_scales=tuple(ag['scales']), anchor_ratios=tuple(ag['ratios']),
                     anchor_stride=ag['strides'][0],
                     target_means=tuple(bc.get('target_means', r['target_means'])),
                     target_stds=tuple(bc.get('target_stds', r['target_stds'])))
        else:
            r.update(rpn_head)
    if roi_head:
        h = cfg['roi_head']
        if 'bbox_roi_extractor' in roi_head:
            ex = roi_head['bbox_roi_extractor']
            h.update(roi_out_size=ex['roi_layer']['output_size'],
                     roi_sampling_ratio=ex['roi_layer'].get('sampling_ratio', 0),
                     featmap_stride=ex['featmap_strides'][0])
            bh, mh = roi_head.get('bbox_head', {}), roi_head.get('mask_head', {})
            bc = bh.get('bbox_coder', {})
            h['bbox_head'].update(in_channels=bh.get('in_channels', 1024), num_classes=bh.get('num_classes', 1),
                                  target_means=tuple(bc.get('target_means', (0., 0., 0., 0.))),
                                  target_stds=tuple(bc.get('target_stds', (.1, .1, .2, .2))))
            h['mask_head'].update({k: mh[k] for k in ('num_convs', 'in_channels', 'conv_out_channels',
                                                      'num_classes') if k in mh})
        else:
            for k, v in roi_head.items():
                if isinstance(v, dict) and k in h:
                    h[k].update(v)
                else:
                    h[k] = v
    if test_cfg:
        t = cfg['test_cfg']
        for part in ('rpn', 'rcnn'):
            src = dict(test_cfg.get(part, {}))
            if 'nms' in src:
                src['nms_iou_threshold'] = src.pop('nms')['iou_threshold']
            t[part].update(src)
    if train_cfg:
        # mmdet layout (fgn_r50_c4_densecl.py:131-171): assigner / sampler sub-dicts are flattened
        t = cfg['train_cfg']
        for part in ('rpn', 'rcnn', 'rpn_proposal'):
            src = dict(train_cfg.get(part, {}))
            for sub, typ in (('assigner', 'MaxIoUAssigner'), ('sampler', 'RandomSampler')):
                d = dict(src.pop(sub, {}))
                if d.pop('type', typ) != typ:
                    raise NotImplementedError(f'train_cfg.{part}.{sub}: only {typ} (both reference configs)')
                d.pop('ignore_iof_thr', None)
                src.update(d)
            if 'nms' in src:
                src['nms_iou_threshold'] = src.pop('nms')['iou_threshold']
            src.pop('debug', None)
            t[part].update(src)
    return cfg


# ------------------------------------------------------------------------------------------
class _Bottleneck:
    def __init__(self, sd, prefix, stride, eps, winograd=0):
        bn = lambda n: {k: sd[f'{prefix}.{n}.{k}'] for k in ('weight', 'bias', 'running_mean', 'running_var')}
        self.conv1 = ops.pack_conv(sd[prefix + '.conv1.weight'], bn=bn('bn1'), relu=True, eps=eps)
        self.conv2 = ops.pack_conv(sd[prefix + '.conv2.weight'], bn=bn('bn2'), stride=stride, pad=1, relu=True,
                                   eps=eps)
        # Winograd form of a stride-1 3x3 (F(4x4,3x3): 4x fewer MFMA passes); chosen per call by ops.winograd_pays
        w2 = sd[prefix + '.conv2.weight']
        self.conv2_wg = ops.pack_winograd(w2, bn=bn('bn2'), relu=True, eps=eps, m=winograd) \
            if winograd and stride == 1 and w2.shape[1] % 32 == 0 and w2.shape[0] % 4 == 0 else None
        self.conv3 = ops.pack_conv(sd[prefix + '.conv3.weight'], bn=bn('bn3'), relu=True, eps=eps)
        self.down = None
        self.conv3_dual = None
        if (prefix + '.downsample.0.weight') in sd:
            dbn = {k: sd[f'{prefix}.downsample.1.{k}'] for k in ('weight', 'bias', 'running_mean', 'running_var')}
            wd = sd[prefix + '.downsample.0.weight']
            self.down = ops.pack_conv(wd, bn=dbn, stride=stride, eps=eps)
            # first block of a stage: conv3 + bn3 and the 1x1 shortcut conv + bn as ONE dual-operand K loop - no launch
            # and no [rows, Cout] round trip for the shortcut (ops.conv1x1_dual).  Stride 1 (layer1.0): both read the
            # same pixels; stride 2 (layer2.0, layer3.0): the shortcut's rows come through a row table (_strided_rows)
            w3 = sd[prefix + '.conv3.weight']
            if ops.FUSED_SHORTCUT and stride in ops.FUSED_SHORTCUT_STRIDES and tuple(wd.shape[2:]) == (1, 1) and \
                    w3.shape[1] % 32 == 0 and wd.shape[1] % 32 == 0 and w3.shape[0] % 4 == 0:
                self.conv3_dual = ops.pack_conv_dual(w3, bn('bn3'), wd, dbn, relu=True, eps=eps)

    def layers(self):
        return [l for l in (self.conv1, self.conv2, self.conv3, self.down, self.conv2_wg, self.conv3_dual) if l is not None]

    def __call__(self, x, n_img_dev=None, y1=None):
        """``y1``: the output of conv1 (+BN+ReLU) when the caller already has it (shared_head entry)."""
        fused = self.conv3_dual is not None and n_img_dev is None
        idt = x if (self.down is None or fused) else ops.conv2d(x, self.down, n_img_dev=n_img_dev)
        y = y1 if y1 is not None else ops.conv2d(x, self.conv1, n_img_dev=n_img_dev)
        if self.conv2_wg is not None and \
                ops.winograd_pays(y.shape[0], y.shape[1], y.shape[2], self.conv2_wg.cin, self.conv2_wg.cout,
                                  self.conv2_wg.m):
            y = ops.conv3x3_winograd(y, self.conv2_wg, n_img_dev=n_img_dev)
        else:
            y = ops.conv2d(y, self.conv2, n_img_dev=n_img_dev)
        if fused:                                                              # relu(bn3(conv3(y)) + bn_d(conv_d(x)))
            rows = None if self.down.stride == 1 else _strided_rows([tuple(x.shape[:3])], self.down.stride, x.device)
            return ops.conv1x1_dual(y, x, self.conv3_dual, x2_rows=rows)
        return ops.conv2d(y, self.conv3, residual=idt, n_img_dev=n_img_dev)   # relu(bn3(conv3) + identity)


_ROW_TABLES: dict = {}


def _strided_rows(shapes, stride, device):
    """``ops.strided_rows`` cached per (geometry, stride, device): built once on the host, outside any graph capture of
    the same geometry (the eager run that precedes a capture fills the cache)."""
    key = (tuple(shapes), int(stride), str(device))
    t = _ROW_TABLES.get(key)
    if t is None:
        t = _ROW_TABLES[key] = ops.strided_rows(shapes, stride, device)
    return t


class _Pair:
    """Two NHWC tensors (query map, support maps) that live in ONE [M_q + M_s, C] buffer, so that the layers whose
    work does not depend on the spatial structure (1x1 / stride 1 convolutions, the grouped Winograd GEMM) run as one
    launch over all rows."""

    def __init__(self, q_shape, s_shape, c, device):
        self.mq = q_shape[0] * q_shape[1] * q_shape[2]
        self.ms = s_shape[0] * s_shape[1] * s_shape[2]
        self.buf = torch.empty((self.mq + self.ms, c), device=device, dtype=torch.float32)
        self.q = self.buf[:self.mq].view(*q_shape, c)
        self.s = self.buf[self.mq:].view(*s_shape, c)

    @property
    def flat(self):
        return self.buf.view(1, self.mq + self.ms, 1, self.buf.shape[1])

    def like(self, c, q_hw=None, s_hw=None):
        qs = self.q.shape[:3] if q_hw is None else (self.q.shape[0],) + tuple(q_hw)
        ss = self.s.shape[:3] if s_hw is None else (self.s.shape[0],) + tuple(s_hw)
        return _Pair(tuple(qs), tuple(ss), c, self.buf.device)


USE_CONV_PAIR = os.environ.get('FGN_CONV_PAIR', '1') != '0'      # A/B knob: 0 = one launch per tensor


def _conv_pair(xq, xs, layer, out_q, out_s):
    """A convolution that strides over the spatial structure on the query map and the support maps: one two-tensor
    launch (ops.conv2d_pair), or one launch each."""
    if USE_CONV_PAIR and layer.cout % 4 == 0 and (layer.cin == 4 or layer.cin % 32 == 0):
        ops.conv2d_pair(xq, xs, layer, out_q, out_s)
    else:
        ops.conv2d(xq, layer, out=out_q)
        ops.conv2d(xs, layer, out=out_s)


def _bottleneck_pair(blk: '_Bottleneck', x: _Pair) -> _Pair:
    """``_Bottleneck.__call__`` on a query / support pair: merged launches for conv1, conv3 (+ residual), the
    stride-1 downsample and the Winograd GEMM; two-tensor launches for the strided convolutions and the transforms."""
    stride = blk.conv2.stride
    out_hw = lambda t: ((t.shape[1] - 1) // stride + 1, (t.shape[2] - 1) // stride + 1)
    fused = blk.conv3_dual is not None
    if blk.down is None or fused:
        idt = x
    else:
        idt = x.like(blk.down.cout, out_hw(x.q), out_hw(x.s))
        if blk.down.stride == 1:
            ops.conv2d(x.flat, blk.down, out=idt.flat)
        else:
            _conv_pair(x.q, x.s, blk.down, idt.q, idt.s)
    y1 = x.like(blk.conv1.cout)
    ops.conv2d(x.flat, blk.conv1, out=y1.flat)
    y2 = x.like(blk.conv2.cout, out_hw(x.q), out_hw(x.s))
    wg = blk.conv2_wg
    if wg is not None and ops.winograd_pays(1, (y1.mq + y1.ms) // 64 + 1, 64, wg.cin, wg.cout, wg.m):
        ops.conv3x3_winograd_multi([y1.q, y1.s], wg, [y2.q, y2.s])
    else:
        _conv_pair(y1.q, y1.s, blk.conv2, y2.q, y2.s)
    out = y2.like(blk.conv3.cout)
    if fused:       # (stride 1: y2 and x hold the same rows; stride 2: the shortcut's rows through a row table)
        rows = None if stride == 1 else _strided_rows([tuple(x.q.shape[:3]), tuple(x.s.shape[:3])], stride, x.buf.device)
        ops.conv1x1_dual(y2.flat, x.flat, blk.conv3_dual, out=out.flat, x2_rows=rows)
    else:
        ops.conv2d(y2.flat, blk.conv3, residual=idt.flat, out=out.flat)
    return out


class _BasicBlock:
    """mmdet BasicBlock of the ResNet-18 extension (config.fgn_r18_c4_config): conv3x3(stride)+BN+ReLU, conv3x3+BN,
    + identity, ReLU.  conv1 takes the Winograd form when it has stride 1 and pays; conv2 carries the residual in
    the epilogue of the direct kernel."""

    def __init__(self, sd, prefix, stride, eps, winograd=0):
        bn = lambda n: {k: sd[f'{prefix}.{n}.{k}'] for k in ('weight', 'bias', 'running_mean', 'running_var')}
        w1 = sd[prefix + '.conv1.weight']
        self.conv1 = ops.pack_conv(w1, bn=bn('bn1'), stride=stride, pad=1, relu=True, eps=eps)
        self.conv1_wg = ops.pack_winograd(w1, bn=bn('bn1'), relu=True, eps=eps, m=winograd) \
            if winograd and stride == 1 and w1.shape[1] % 32 == 0 and w1.shape[0] % 4 == 0 else None
        self.conv2 = ops.pack_conv(sd[prefix + '.conv2.weight'], bn=bn('bn2'), pad=1, relu=True, eps=eps)
        self.down = None
        if (prefix + '.downsample.0.weight') in sd:
            dbn = {k: sd[f'{prefix}.downsample.1.{k}'] for k in ('weight', 'bias', 'running_mean', 'running_var')}
            self.down = ops.pack_conv(sd[prefix + '.downsample.0.weight'], bn=dbn, stride=stride, eps=eps)

    def layers(self):
        return [l for l in (self.conv1, self.conv1_wg, self.conv2, self.down) if l is not None]

    def __call__(self, x, n_img_dev=None):
        idt = x if self.down is None else ops.conv2d(x, self.down, n_img_dev=n_img_dev)
        wg = self.conv1_wg
        if wg is not None and ops.winograd_pays(x.shape[0], x.shape[1], x.shape[2], wg.cin, wg.cout, wg.m):
            y = ops.conv3x3_winograd(x, wg, n_img_dev=n_img_dev)
        else:
            y = ops.conv2d(x, self.conv1, n_img_dev=n_img_dev)
        return ops.conv2d(y, self.conv2, residual=idt, n_img_dev=n_img_dev)   # relu(bn2(conv2) + identity)


class _BottleneckGN:
    """Bottleneck of the from-scratch backbone (fgn_r50_c4_scratch.py:16-25): conv -> GroupNorm -> ReLU;
    the norm cannot fold into the conv epilogue, so every conv is followed by the GN passes.  Shortcut
    under avg_down: [2x2 average pool if strided] -> 1x1 conv (stride 1) -> GN."""

    def __init__(self, sd, prefix, stride, groups, eps, avg_down):
        self.groups, self.eps = groups, eps
        self.conv1 = ops.pack_conv(sd[prefix + '.conv1.weight'])
        self.conv2 = ops.pack_conv(sd[prefix + '.conv2.weight'], stride=stride, pad=1)
        self.conv3 = ops.pack_conv(sd[prefix + '.conv3.weight'])
        self.gn = [[sd[f'{prefix}.gn{i}.weight'].clone(), sd[f'{prefix}.gn{i}.bias'].clone()] for i in (1, 2, 3)]
        self.pooled = bool(avg_down) and stride != 1
        if self.pooled and stride != 2:
            raise NotImplementedError('avg_down shortcut: only stride 2')
        i = 1 if self.pooled else 0
        # (mmdet's ResLayer may put the AvgPool2d in front of a stride-1 shortcut too - kernel 1, the identity - which
        # shifts the conv / norm to downsample.1 / .2 in layer1 as well: unverifiable here (mmdet 2.18 is absent), so a
        # checkpoint in either key layout loads)
        if f'{prefix}.downsample.{i}.weight' not in sd and f'{prefix}.downsample.{1 - i}.weight' in sd:
            i = 1 - i
        self.down = None
        if f'{prefix}.downsample.{i}.weight' in sd:
            self.down = ops.pack_conv(sd[f'{prefix}.downsample.{i}.weight'], stride=1 if avg_down else stride)
            self.gn.append([sd[f'{prefix}.downsample.{i + 1}.weight'].clone(),
                            sd[f'{prefix}.downsample.{i + 1}.bias'].clone()])

    def layers(self):
        return [l for l in (self.conv1, self.conv2, self.conv3, self.down) if l is not None]

    def to(self, device):
        for l in self.layers():
            l.to(device)
        self.gn = [[t.float().contiguous().to(device) for t in pair] for pair in self.gn]
        return self

    def _norm(self, x, i, relu, residual=None):
        return ops.group_norm(x, self.gn[i][0], self.gn[i][1], self.groups, self.eps, relu=relu, residual=residual,
                              inplace=True)

    def __call__(self, x, n_img_dev=None):
        idt = x
        if self.down is not None:
            idt = self._norm(ops.conv2d(ops.avgpool2x2(x) if self.pooled else x, self.down), 3, False)
        y = self._norm(ops.conv2d(x, self.conv1), 0, True)
        y = self._norm(ops.conv2d(y, self.conv2), 1, True)
        return self._norm(ops.conv2d(y, self.conv3), 2, True, residual=idt)   # relu(gn3(conv3) + identity)


class _ResultRecord:
    """Every per-image output of one batch in ONE buffer (device, or its pinned host mirror): the selection / mask
    kernels write straight into its views and ONE device-to-host copy moves it (round 5: the nine small copies per
    episode it replaces were ~6 us blit kernels each on the caller stream).  Layout, 256-byte aligned fields:
    [n_dets | rle_len | rle_overflow | mask_prob] (the zero-initialised head) | det_bboxes | det_labels | rle bytes."""

    def __init__(self, batch: int, max_det: int, mask_size: int, device=None, pinned: bool = False):
        self.key = (batch, max_det, mask_size, ops.RLE_BYTE_CAP)
        off, spec = 0, []
        for name, shape, dtype in (('cnt', (batch, 1), torch.int32), ('rle_len', (batch, max_det), torch.int32),
                                   ('rle_ovf', (batch, max_det), torch.int32),
                                   ('prob', (batch, max_det, mask_size, mask_size), torch.float32), (None, None, None),
                                   ('det', (batch, max_det, 5), torch.float32), ('lab', (batch, max_det), torch.int64),
                                   ('rle', (batch, max_det, ops.RLE_BYTE_CAP), torch.uint8)):
            if name is None:
                self.zero_bytes = off
                continue
            n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
            spec.append((name, shape, dtype, off, n))
            off = (off + n + 255) // 256 * 256
        self.nbytes = off
        self.buf = torch.empty(off, dtype=torch.uint8, pin_memory=True) if pinned else \
            torch.empty(off, dtype=torch.uint8, device=device)
        for name, shape, dtype, o, n in spec:
            setattr(self, name, self.buf[o:o + n].view(dtype).view(*shape))

    def clear_head(self) -> None:
        self.buf[:self.zero_bytes].zero_()


class _GraphedEpisode:
    """The launch sequence of ``FGN._detect_eager`` for one input geometry, captured once as a
    hipGraph (both HIP streams of the path fork from and join the capture stream).  Inputs are copied
    into static device buffers (this copy is the H2D step of fgn.py:79-108 when the batch arrives on
    the host), outputs live in static buffers that the next replay overwrites - so the replay waits
    for the previous download; that download (one copy of the batch's result record) carries every tensor
    ``pack_results`` may read."""
    CODE_KEYS = ('vec', 'S', 'cat_mean_mp')

    def __init__(self, model, ins: dict, img_shape, support_code, dev, phase_counter=None):
        self.static = {k: torch.empty(v.shape, dtype=v.dtype, device=dev) for k, v in ins.items()}
        self.img_shape = torch.as_tensor(img_shape).cpu().clone()
        self.code = None
        if support_code is not None:
            self.code = dict(support_code)
            for k in self.CODE_KEYS:
                self.code[k] = support_code[k].clone()
        self.last_download = None
        for k, v in ins.items():
            self.static[k].copy_(v, non_blocking=True)
        args = lambda: (self.static['qry_img'], self.static.get('spp_imgs'), self.static.get('spp_bboxes'),
                        self.static.get('spp_isegmaps'), self.img_shape, self.code)
        # eager pass first: packs the weights, sets kernel attributes, sizes the allocator pools
        # (the eager run that precedes the capture sends no phase mark: every ``detect_device`` call bumps the caller's
        # counter exactly once - here through the replay that follows the capture)
        model._detect_eager(*args(), phase_counter=None)
        torch.cuda.current_stream().synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # launch records of the dominant kernel (``FGN.stamp_capacity``; bench.py's roofline over the whole timed window):
        # the record of every conv_pw_persist_kernel launch is baked into the captured launch, each replay adds its span
        self.stamps, self.stamp_count = None, 0
        if model.stamp_capacity:
            self.stamps = ops.new_stamp_records(int(model.stamp_capacity), dev)
            torch.cuda.current_stream().synchronize()
            ops.arm_stamps(self.stamps)
        # thread-local capture mode: only this thread's calls are checked against the capture, so nothing another
        # thread does (the watchdog of an RCCL process group polling the all-gather of the previous step) can
        # invalidate it.  (The global mode passed the same test on this torch build; FGN_GRAPH_CAPTURE_MODE selects.)
        try:
            with torch.cuda.graph(self.graph, capture_error_mode=os.environ.get('FGN_GRAPH_CAPTURE_MODE', 'thread_local')):
                self.outs = model._detect_eager(*args(), phase_counter=phase_counter)
        finally:
            if self.stamps is not None:
                self.stamp_count = ops.arm_stamps(None)

    def run(self, model, ins: dict, support_code, main) -> list:
        for k, v in ins.items():
            if v is not self.static[k]:                # (already uploaded straight into the static buffer: detect_device)
                self.static[k].copy_(v, non_blocking=True)
        if support_code is not None:
            for k in self.CODE_KEYS:
                if support_code[k].data_ptr() != self.code[k].data_ptr():
                    self.code[k].copy_(support_code[k], non_blocking=True)
        if self.last_download is not None:
            main.wait_event(self.last_download)       # static outputs are still being read by the copy stream
        self.graph.replay()
        # (the static outputs are views of the batch's result record: its one download carries everything pack_results
        # reads, the mask probabilities and boxes of the rare RLE-overflow fallback included)
        return [dict(d) for d in self.outs]          # detect_device queues the download and sets ``last_download``


class FGN(torch.nn.Module):
    """Fully Guided Network, inference path, on MI355X HIP kernels."""
    fp16_enabled = False
    subsampling_ratio = 16

    def __init__(self, n_ways: int, k_shots: int, backbone: Optional[dict] = None, rpn_head: Optional[dict] = None,
                 roi_head: Optional[dict] = None, train_cfg: Optional[dict] = None, test_cfg: Optional[dict] = None,
                 neck=None, pretrained=None, init_cfg=None, state_dict: Optional[dict] = None, seed: int = 0,
                 type: Optional[str] = None, **kwargs):
        super().__init__()
        if type not in (None, 'FGN'):
            raise ValueError(f'cannot build detector type {type!r}')
        self.n_ways, self.k_shots = int(n_ways), int(k_shots)
        self.cfg = normalise_config(self.n_ways, self.k_shots, backbone, rpn_head, roi_head, test_cfg, train_cfg)
        self.train_cfg = train_cfg
        self.test_cfg = self.cfg['test_cfg']
        # Region semantics of the mask paste (mmdet `_do_paste_mask`): 'cpu' = skip_empty=True, a mask is pasted inside the
        # integer-expanded box only - the CPU reference north_star names, this build's oracle and default; 'cuda' =
        # skip_empty=False, the grid spans the whole image - what the reference computes where it actually runs (cuda:0,
        # main.py:365).  Identical at the configured threshold 0.5 for boxes of positive width and height
        # (fgn_r50_c4_densecl.py:186: the value on the box edge is half the border pixel), up to box_w / 28 more pixels outside the box under 'cuda' below it.
        self.paste_semantics = 'cpu'
        if self.test_cfg['rcnn'].get('mask_thr_binary', 0.5) < 0.5:
            import warnings
            warnings.warn("mask_thr_binary < 0.5: mmdet's CPU and CUDA paste paths differ below 0.5 (skip_empty); this "
                          "detector follows the CPU path unless `paste_semantics = 'cuda'` is set (DESIGN.md section 2)",
                          stacklevel=2)
        self._sd = OrderedDict((k, self._canon(k, v)) for k, v in
                               (state_dict if state_dict is not None else init_state_dict(self.cfg, seed)).items())
        # mmcv `Pretrained` init_cfg of the backbone (fgn_r50_c4_densecl.py:39-41) / the deprecated `pretrained=` kwarg:
        # applied by init_weights(), as in the reference (main.py:431-434)
        ic = (backbone or {}).get('init_cfg') or init_cfg
        ic = ic[0] if isinstance(ic, (list, tuple)) and ic else ic
        self.backbone_pretrained = pretrained or (ic.get('checkpoint') if isinstance(ic, dict) and
                                                  ic.get('type') == 'Pretrained' else None) or None
        self._frozen_stages = int((backbone or {}).get('frozen_stages', 4)) if isinstance(backbone, dict) else 4
        self._packed_device = None
        self._PT = None                           # train-mode layers of the shared head (fgn_amd.train.pack_train)
        self.debug_trace: Optional[dict] = None   # set to {} to capture intermediates (tests)
        self.use_side_stream = True               # support branch on a second HIP stream
        self.use_graphs = False                   # replay a captured hipGraph per input geometry
        self._use_winograd = ops.WINOGRAD_M       # Winograd form of the 3x3 / stride 1 convs: 4 = F(4x4,3x3), 2 = F(2x2,3x3), 0 = direct
        self.use_roi_commute = True               # shared_head conv1 on the feature map, RoIAlign after (set before first use)
        # query + support maps through SHARED backbone launches (1x1 / stride 1 convolutions and the grouped Winograd GEMM
        # over all rows of both; frozen-BN bottleneck backbones): half the launches of the two passes, no support-only
        # split-K.  Round 2 measured this 5 % slower with one episode in flight (the support stream filled the query
        # branch's idle phases); with two episodes in flight the other episode does that, and it is 7 % faster
        # (same-box A/B, r03: 6.07 -> 5.72 ms).  Results differ from the separate passes in the last bits only
        # (split-K plans and Winograd-vs-direct choices depend on the row count).
        self.use_merged_backbone = True
        # the support RoIs ride through the shared head in the box head's RoI batch (same weights; eval-mode BatchNorm is
        # per sample): the ~30 small launches of count_spp's own shared-head pass (9 RoIs, 441 rows: three direct-form
        # 3x3 convolutions at 0.08 of the MFMA peak while they queue behind the persistent GEMMs, r03 kernel stats)
        # disappear into 3 % more rows of the 300-RoI launches.  Step time equal within run-to-run noise (same-box A/B,
        # r04: B = 1 189.9-191.3 vs 188.6-190.9 img/s, B = 4 211.1-211.8 vs 209.2-211.0, B = 8 214.6 vs 215.5): on since
        # round 4 for the launch count.  False = the separate pass, byte-identical to `encode_supports`.
        self.use_merged_support_head = True
        self.transfer_mode = 0                    # 0: upload + copy stream per caller; 1 / 2: see transfer_stream()
        # one device-to-host copy per batch (every output lands in one `_ResultRecord`) and, under graph replay with the
        # transfers on the caller stream, host inputs copied straight into the graph's static buffers (round 5: 15 -> 6
        # small copies per episode on the caller stream); False = the per-field copies of rounds 1-4 (A/B knob)
        self.use_packed_transfers = True
        self.stamp_capacity = 0                   # > 0: captured graphs carry launch records of the dominant kernel (ops.read_stamps)
        self._graphs: dict = {}
        self._streams: dict = {}                  # (role, caller stream) -> HIP stream: 'side', 'copy', 'upload'
        self._pinned: dict = {}                   # (batch, max_det, byte cap) -> list of pinned host slots

    def _skip_empty(self) -> bool:
        if self.paste_semantics not in ('cpu', 'cuda'):
            raise ValueError(f"paste_semantics must be 'cpu' or 'cuda', got {self.paste_semantics!r}")
        return self.paste_semantics == 'cpu'

    @property
    def use_winograd(self) -> int:
        return self._use_winograd

    @use_winograd.setter
    def use_winograd(self, on) -> None:
        """False / 0: direct implicit GEMM for every 3x3; True: the default Winograd form (F(4x4,3x3)); 2 or 4: that
        output tile edge."""
        m = ops.WINOGRAD_M if on is True else int(on)
        if m not in (0, 2, 4):
            raise ValueError('use_winograd: False, True, 2 or 4')
        if m != self._use_winograd:               # the packed layers (and any captured graph) depend on it
            self._use_winograd = m
            self._packed_device = None
            self._PT = None
            self._graphs = {}

    @classmethod
    def from_config(cls, model_cfg: dict, **kw) -> 'FGN':
        """``build_detector(cfg.model, ...)`` equivalent (main.py:390-394)."""
        model_cfg = copy.deepcopy(dict(model_cfg))
        model_cfg.pop('type', None)
        model_cfg.update(kw)
        return cls(**model_cfg)

    # --- weights --------------------------------------------------------------------------
    def _trainer_alive(self):
        ref = getattr(self, '_trainer', None)
        return ref() if ref is not None else None

    def _source_sd(self) -> dict:
        """The weights every (re-)pack starts from: the state dict, overlaid with the device-resident master weights
        and running statistics of a live ``fgn_amd.train.Trainer`` - so a re-pack after training steps (device change,
        ``use_winograd`` setter, ``_packed_device`` reset) never falls back to the initial heads."""
        tr = self._trainer_alive()
        if tr is None:
            return self._sd
        sd = OrderedDict(self._sd)
        sd.update(tr.W)
        sd.update(tr.buffers)
        return sd

    def state_dict(self, *a, **k):
        tr = self._trainer_alive()
        return OrderedDict(self._sd) if tr is None else OrderedDict(tr.state_dict())

    @staticmethod
    def _canon(name: str, v: torch.Tensor) -> torch.Tensor:
        return v.detach().cpu() if name.endswith('num_batches_tracked') else v.detach().float().cpu()

    @property
    def backbone(self):
        """What main.py:402-405 reads off ``model.backbone`` right after ``build_detector``: ``frozen_stages`` (from the
        reference's config dict; 4 = the DenseCL configuration, -1 = from scratch), the list of stage names it shortens
        to drop res5 (this detector is C4 by construction: three stages), and ``eval()`` (BatchNorm is folded from the
        running statistics at pack time: there is no train-mode backbone to switch)."""
        from types import SimpleNamespace
        ns = SimpleNamespace(frozen_stages=self._frozen_stages, norm_eval=bool(self.cfg['backbone'].get('norm_eval', True)),
                             res_layers=[f'layer{i + 1}' for i in range(len(self.cfg['backbone']['stage_blocks']) + 1)])
        ns.eval = lambda: ns
        ns.train = lambda mode=True: ns
        return ns

    def init_weights(self) -> None:
        """mmcv ``BaseModule.init_weights`` as far as this path needs it: the backbone's ``Pretrained`` init_cfg
        (fgn_r50_c4_densecl.py:39-41; main.py:431-434).  Without one the seeded initialisation stands."""
        if self.backbone_pretrained:
            self.load_backbone_pretrained(self.backbone_pretrained)

    def load_backbone_pretrained(self, ckpt) -> dict:
        """Load a backbone-only checkpoint (path, or state dict with un-prefixed torchvision / mmcv ResNet keys) into
        ``backbone.*`` only, non-strict like mmcv's ``load_checkpoint(backbone, ..., strict=False)``: entries the C4
        backbone does not have (``layer4.*``, ``fc.*``) are reported as unexpected, absent ones as missing (the
        optional ``num_batches_tracked`` buffers aside).  Heads keep their weights."""
        src = backbone_state_from_pretrained(ckpt)
        own = [k for k in self._sd if k.startswith('backbone.')]
        picked = {k: src[k] for k in own if k in src}
        for k, v in picked.items():
            if tuple(v.shape) != tuple(self._sd[k].shape):
                raise ValueError(f'shape mismatch for {k}: {tuple(v.shape)} vs {tuple(self._sd[k].shape)}')
        report = dict(loaded=len(picked),
                      missing=[k for k in own if k not in src and not k.endswith('num_batches_tracked')],
                      unexpected=[k for k in src if k not in self._sd])
        if not picked:
            raise KeyError('no backbone tensor found in the checkpoint (expected keys like conv1.weight, layer1.0.conv1.weight)')
        self.load_state_dict(picked, strict=False)
        return report

    def load_state_dict(self, state_dict, strict: bool = True):
        sd = state_dict.get('state_dict', state_dict)
        missing = [k for k in self._sd if k not in sd and not k.endswith('num_batches_tracked')]
        if strict and missing:
            raise KeyError(f'missing keys in state_dict: {missing[:5]} ...')
        for k in self._sd:
            if k in sd:
                if tuple(sd[k].shape) != tuple(self._sd[k].shape):
                    raise ValueError(f'shape mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(self._sd[k].shape)}')
                self._sd[k] = self._canon(k, sd[k])
        self._packed_device = None
        self._PT = None
        self._graphs = {}
        tr = self._trainer_alive()
        if tr is not None:              # a live trainer adopts the loaded weights (its Adagrad sums are kept)
            tr.adopt(self._sd)

    def _pack(self, device):
        """Fold BN, re-layout weights for the kernels and move them to ``device``."""
        sd, cfg = self._source_sd(), self.cfg
        eps = cfg['backbone']['bn_eps']
        P = {}
        bb = cfg['backbone']
        gn = bb.get('norm', 'BN') == 'GN'
        bnp = lambda name: {k: sd[f'{name}.{k}'] for k in ('weight', 'bias', 'running_mean', 'running_var')}
        # stem: list of (conv, GroupNorm params or None); BatchNorm(eval) folds into the conv epilogue
        if bb.get('deep_stem'):
            P['stem'] = []
            for i, stride in enumerate((2, 1, 1)):
                w, nm = sd[f'backbone.stem.{3 * i}.weight'], f'backbone.stem.{3 * i + 1}'
                kw = dict(stride=stride, pad=1, pad_cin_to=4 if i == 0 else None)
                if gn:
                    P['stem'].append((ops.pack_conv(w, **kw), [sd[nm + '.weight'].clone(), sd[nm + '.bias'].clone()]))
                else:
                    P['stem'].append((ops.pack_conv(w, bn=bnp(nm), relu=True, eps=eps, **kw), None))
        elif gn:
            P['stem'] = [(ops.pack_conv(sd['backbone.conv1.weight'], stride=2, pad=3, pad_cin_to=4),
                          [sd['backbone.gn1.weight'].clone(), sd['backbone.gn1.bias'].clone()])]
        else:
            P['stem'] = [(ops.pack_conv(sd['backbone.conv1.weight'], bn=bnp('backbone.bn1'), stride=2, pad=3,
                                        relu=True, eps=eps, pad_cin_to=4), None)]
        P['stages'] = []
        for li, (nblk, stride) in enumerate(zip(bb['stage_blocks'], bb['strides'])):
            P['stages'].append([
                _BottleneckGN(sd, f'backbone.layer{li + 1}.{b}', stride if b == 0 else 1, bb.get('gn_groups', 32),
                              eps, bb.get('avg_down', False)) if gn else
                (_BasicBlock if bb.get('block', 'bottleneck') == 'basic' else _Bottleneck)(
                    sd, f'backbone.layer{li + 1}.{b}', stride if b == 0 else 1, eps, winograd=self.use_winograd)
                for b in range(nblk)])
        P.update(self._pack_heads(sd))
        P.update(self._pack_shared(sd))
        rp = cfg['rpn_head']
        P['anchors'] = torch.from_numpy(ops.base_anchors(rp['anchor_scales'], rp['anchor_ratios'],
                                                         rp['anchor_stride']))

        def mv(o):
            if isinstance(o, torch.Tensor):
                return o.float().contiguous().to(device)
            if isinstance(o, (ops.ConvLayer, ops.WinogradLayer, ops.DualConvLayer)):
                return o.to(device)
            if isinstance(o, (_Bottleneck, _BasicBlock)):
                for l in o.layers():
                    l.to(device)
                return o
            if isinstance(o, _BottleneckGN):
                return o.to(device)
            if isinstance(o, (list, tuple)):
                return [mv(x) for x in o]
            return o
        self._P = {k: mv(v) for k, v in P.items()}
        self._packed_device = torch.device(device)
        self._shared_dirty = None
        if self._trainer_alive() is not None:      # the train-mode layers follow the same weights
            self._PT = None

    def _head_math(self):
        """Arithmetic of the trainable layers' GEMMs: f32 MFMA while a Trainer rewrites their weights every step
        (``ops.gemm_math``), the build's default otherwise."""
        return 'f32' if self._trainer_alive() is not None else None

    def _pack_shared(self, sd) -> dict:
        with ops.gemm_math(self._head_math()):
            return self._pack_shared_impl(sd)

    def _pack_heads(self, sd) -> dict:
        with ops.gemm_math(self._head_math()):
            return self._pack_heads_impl(sd)

    def _pack_shared_impl(self, sd) -> dict:
        """The shared head for INFERENCE (BatchNorm in eval mode, folded into the conv epilogues) from torch-layout
        weights and running statistics ``sd`` (CPU state dict, or a Trainer's device-resident masters + buffers)."""
        cfg = self.cfg
        eps = cfg['backbone']['bn_eps']
        P = {}
        P['shared'] = [_Bottleneck(sd, f'roi_head.shared_head.{b}', 1, eps, winograd=self.use_winograd)
                       for b in range(cfg['roi_head']['shared_head']['num_blocks'])]
        # conv1 of the first shared_head block with its BN scale folded and no shift / ReLU: applied to the C4 map
        # (RoIAlign commutes with it); the shift is added after the pooling
        c1 = P['shared'][0].conv1
        P['sh0_lin'] = ops.ConvLayer(c1.w.clone(), None if c1.scale is None else c1.scale.clone(), None, c1.cin,
                                     c1.cout, c1.cout_pad, c1.kh, c1.kw, c1.stride, c1.pad, False,
                                     None if c1.w3 is None else c1.w3.clone(), None if c1.wh is None else c1.wh.clone()) \
            if self.use_roi_commute else None
        P['sh0_shift'] = c1.shift.clone() if (self.use_roi_commute and c1.shift is not None) else None
        return P

    def _pack_heads_impl(self, sd) -> dict:
        """The packed AG-RPN / relation / box / mask head layers from torch-layout weights ``sd`` (CPU tensors of the
        state dict, or the device-resident master weights of ``fgn_amd.train.Trainer``: packing is torch ops only)."""
        cfg = self.cfg
        P = {}
        P['rpn_conv'] = ops.pack_conv(sd['rpn_head.rpn_conv.weight'], bias=sd['rpn_head.rpn_conv.bias'], pad=1,
                                      relu=True)
        wr = sd['rpn_head.rpn_conv.weight']
        P['rpn_conv_wg'] = ops.pack_winograd(wr, bias=sd['rpn_head.rpn_conv.bias'], relu=True, m=self.use_winograd) \
            if self.use_winograd and wr.shape[1] % 32 == 0 and wr.shape[0] % 4 == 0 else None
        # objectness and delta 1x1 convs fused into one launch: channels [0,A) | [A,5A)
        # (zero rows pad the 5A = 75 channels to a multiple of 4: 16-byte epilogue stores and split-K become
        # available to the launch; rpn_merge reads the first 5A channels of each pixel)
        wh = torch.cat([sd['rpn_head.rpn_cls.weight'], sd['rpn_head.rpn_reg.weight']], 0)
        bh = torch.cat([sd['rpn_head.rpn_cls.bias'], sd['rpn_head.rpn_reg.bias']], 0)
        padc = (-wh.shape[0]) % 4
        if padc:
            wh = torch.cat([wh, wh.new_zeros((padc,) + tuple(wh.shape[1:]))], 0)
            bh = torch.cat([bh, bh.new_zeros(padc)], 0)
        P['rpn_head'] = ops.pack_conv(wh, bias=bh)
        # relation conv split along its input channels: [Wq | Ws] (fgn_roi_head.py:270)
        wrel = sd['roi_head.cls_reg_shared_conv.weight']
        c = wrel.shape[1] // 2
        P['rel_q'] = ops.pack_conv(wrel[:, :c].contiguous())
        P['rel_s'] = ops.pack_conv(wrel[:, c:].contiguous(), bias=sd['roi_head.cls_reg_shared_conv.bias'])
        P['gn_w'] = sd['roi_head.cls_reg_shared_conv_norm.weight'].clone()
        P['gn_b'] = sd['roi_head.cls_reg_shared_conv_norm.bias'].clone()
        P['fc_w'] = torch.cat([sd['roi_head.bbox_head.fc_cls.weight'], sd['roi_head.bbox_head.fc_reg.weight']], 0)
        P['fc_b'] = torch.cat([sd['roi_head.bbox_head.fc_cls.bias'], sd['roi_head.bbox_head.fc_reg.bias']], 0)
        mh = cfg['roi_head']['mask_head']
        P['mask_convs'] = [ops.pack_conv(sd[f'roi_head.mask_head.convs.{i}.conv.weight'],
                                         bias=sd[f'roi_head.mask_head.convs.{i}.conv.bias'], pad=1, relu=True)
                           for i in range(mh['num_convs'])]
        P['mask_convs_wg'] = [
            ops.pack_winograd(sd[f'roi_head.mask_head.convs.{i}.conv.weight'],
                              bias=sd[f'roi_head.mask_head.convs.{i}.conv.bias'], relu=True, m=self.use_winograd)
            if self.use_winograd and sd[f'roi_head.mask_head.convs.{i}.conv.weight'].shape[1] % 32 == 0 and
            sd[f'roi_head.mask_head.convs.{i}.conv.weight'].shape[0] % 4 == 0 else None
            for i in range(mh['num_convs'])]
        # ConvTranspose2d(k=2,s=2) [Cin,Cout,2,2] -> 1x1 conv with 4*Cout outputs, n=(dy*2+dx)*Cout+co
        wt = sd['roi_head.mask_head.upsample.weight']
        cin_u, cout_u = wt.shape[:2]
        w4 = wt.permute(2, 3, 1, 0).reshape(4 * cout_u, cin_u, 1, 1).contiguous()
        P['upsample'] = ops.pack_conv(w4, bias=sd['roi_head.mask_head.upsample.bias'].repeat(4), relu=True)
        P['logit_w'] = sd['roi_head.mask_head.conv_logits.weight'].reshape(-1).clone()
        # a float for host-resident weights; the device-resident master weights of a Trainer stay on the device (the
        # kernel reads the bias through a pointer: no host synchronisation at the end of every training step)
        lb = sd['roi_head.mask_head.conv_logits.bias']
        P['logit_b'] = lb.reshape(-1)[:1].float().contiguous().clone() if lb.is_cuda else float(lb[0])
        return P

    def _repack_heads_(self, sd) -> None:
        """``_pack_heads`` of updated DEVICE weights into the layers ``self._P`` already holds (same shapes): one strided
        copy per convolution, one kernel per Winograd layer, no allocation - the re-pack between two training steps
        (round 4: 2.5 ms of ~150 torch kernels -> a few dozen launches).  Same values as a fresh ``_pack_heads``."""
        P = self._P
        ops.repack_conv_(P['rpn_conv'], sd['rpn_head.rpn_conv.weight'], sd['rpn_head.rpn_conv.bias'])
        if P['rpn_conv_wg'] is not None:
            ops.repack_winograd_(P['rpn_conv_wg'], sd['rpn_head.rpn_conv.weight'], sd['rpn_head.rpn_conv.bias'])
        # fused objectness + delta 1x1 conv: channels [0, A) | [A, 5A) (+ zero rows up to a multiple of 4)
        h = P['rpn_head']
        a = sd['rpn_head.rpn_cls.weight'].shape[0]
        n = a + sd['rpn_head.rpn_reg.weight'].shape[0]
        h.w[:a, :h.cin].copy_(sd['rpn_head.rpn_cls.weight'].reshape(a, h.cin))
        h.w[a:n, :h.cin].copy_(sd['rpn_head.rpn_reg.weight'].reshape(n - a, h.cin))
        h.shift[:a].copy_(sd['rpn_head.rpn_cls.bias'])
        h.shift[a:n].copy_(sd['rpn_head.rpn_reg.bias'])
        wrel = sd['roi_head.cls_reg_shared_conv.weight']
        c = wrel.shape[1] // 2
        P['rel_q'].w[:wrel.shape[0], :c].copy_(wrel[:, :c, 0, 0])
        P['rel_s'].w[:wrel.shape[0], :c].copy_(wrel[:, c:, 0, 0])
        P['rel_s'].shift.copy_(sd['roi_head.cls_reg_shared_conv.bias'])
        P['gn_w'].copy_(sd['roi_head.cls_reg_shared_conv_norm.weight'])
        P['gn_b'].copy_(sd['roi_head.cls_reg_shared_conv_norm.bias'])
        nc = sd['roi_head.bbox_head.fc_cls.weight'].shape[0]
        P['fc_w'][:nc].copy_(sd['roi_head.bbox_head.fc_cls.weight'])
        P['fc_w'][nc:].copy_(sd['roi_head.bbox_head.fc_reg.weight'])
        P['fc_b'][:nc].copy_(sd['roi_head.bbox_head.fc_cls.bias'])
        P['fc_b'][nc:].copy_(sd['roi_head.bbox_head.fc_reg.bias'])
        for i, (layer, wg) in enumerate(zip(P['mask_convs'], P['mask_convs_wg'])):
            w, b = sd[f'roi_head.mask_head.convs.{i}.conv.weight'], sd[f'roi_head.mask_head.convs.{i}.conv.bias']
            ops.repack_conv_(layer, w, b)
            if wg is not None:
                ops.repack_winograd_(wg, w, b)
        # ConvTranspose2d(k=2,s=2) [Cin,Cout,2,2] -> 1x1 conv with 4*Cout outputs, n=(dy*2+dx)*Cout+co
        wt = sd['roi_head.mask_head.upsample.weight']
        cin_u, cout_u = wt.shape[:2]
        up = P['upsample']
        up.w[:4 * cout_u].view(2, 2, cout_u, up.w.shape[1])[..., :cin_u].copy_(wt.permute(2, 3, 1, 0))
        up.shift.view(4, cout_u).copy_(sd['roi_head.mask_head.upsample.bias'].expand(4, cout_u))
        P['logit_w'].copy_(sd['roi_head.mask_head.conv_logits.weight'].reshape(-1))
        P['logit_b'].copy_(sd['roi_head.mask_head.conv_logits.bias'].reshape(-1)[:1])

    # --- stages ---------------------------------------------------------------------------
    def extract_feat(self, img_nchw: torch.Tensor) -> torch.Tensor:
        """ResNet-50 stages 1-3 (fgn.py:67-77): NCHW fp32 in, NHWC [B,h,w,1024] out."""
        P = self._P
        bb = self.cfg['backbone']
        x = ops.nchw3_to_nhwc4(img_nchw.contiguous())
        for conv, gn in P['stem']:
            x = ops.conv2d(x, conv)
            if gn is not None:
                x = ops.group_norm(x, gn[0], gn[1], bb.get('gn_groups', 32), bb['bn_eps'], relu=True, inplace=True)
        x = ops.maxpool3x3s2(x)
        for stage in P['stages']:
            for blk in stage:
                x = blk(x)
        return x

    def extract_feat_pair(self, qry_nchw: torch.Tensor, spp_nchw: torch.Tensor, phase_counter=None):
        """Both backbone passes of an episode (fgn.py:212-215) through SHARED launches wherever a layer does not look at
        the spatial structure (``use_merged_backbone``; frozen-BN bottleneck backbones only)."""
        P = self._P
        outs = [ops.nchw3_to_nhwc4(img.contiguous()) for img in (qry_nchw, spp_nchw)]
        for conv, _ in P['stem']:
            y = [torch.empty((o.shape[0], (o.shape[1] + 2 * conv.pad - conv.kh) // conv.stride + 1,
                              (o.shape[2] + 2 * conv.pad - conv.kw) // conv.stride + 1, conv.cout), device=o.device,
                             dtype=torch.float32) for o in outs]
            _conv_pair(outs[0], outs[1], conv, y[0], y[1])
            outs = y
        hw = lambda t: ((t.shape[1] - 1) // 2 + 1, (t.shape[2] - 1) // 2 + 1)
        x = _Pair((outs[0].shape[0],) + hw(outs[0]), (outs[1].shape[0],) + hw(outs[1]), outs[0].shape[3], outs[0].device)
        ops.maxpool3x3s2(outs[0], out=x.q)
        ops.maxpool3x3s2(outs[1], out=x.s)
        self._phase_mark('layer1', phase_counter)    # (stem + max-pool done)
        for si, stage in enumerate(P['stages']):
            for blk in stage:
                x = _bottleneck_pair(blk, x)
            self._phase_mark('layer%d' % (si + 2), phase_counter)     # 'layer2' = layer1 done, ..., the last stage's mark equals 'rpn'
        return x.q, x.s

    def _shared_head(self, x, n_img_dev=None, y1=None):
        for bi, blk in enumerate(self._P['shared']):
            x = blk(x, n_img_dev, y1=y1 if bi == 0 else None)
        return x

    def _roi_feats(self, fmap, g_map, rois, n_dev):
        """RoIAlign of the C4 map + the shared_head (fgn_roi_head.py:331-336 / 366-369).  RoIAlign is linear per
        channel, so the first 1x1 conv of the shared_head is taken on the feature map (``g_map``, once per episode,
        50x84 pixels instead of 49 per RoI) and only its BN shift + ReLU follow the pooling."""
        P, rh = self._P, self.cfg['roi_head']
        PS, inv = rh['roi_out_size'], 1.0 / rh['featmap_stride']
        if g_map is not None:       # both maps at the same sampling points: one launch
            x, y1 = ops.roi_align2(fmap, g_map, rois, PS, inv, rh['roi_sampling_ratio'], True, n_dev,
                                   post_shift2=P['sh0_shift'], relu2=True)
        else:
            x, y1 = ops.roi_align(fmap, rois, PS, inv, rh['roi_sampling_ratio'], True, n_dev), None
        return x, self._shared_head(x, n_dev, y1=y1)

    def _mask_head(self, mf, vmask, n_dev=None, prob_out=None):
        """``_mask_forward`` after the shared_head (fgn_roi_head.py:379-380): support-vector guidance, FCNMaskHead
        (4 x conv3x3+ReLU, ConvTranspose 2x2/2 + ReLU as one 1x1 conv with 4*C' outputs, 1x1 logits), sigmoid.
        mf [D,7,7,C], vmask [D,C] -> logits, probabilities [D,14,14]."""
        P = self._P
        m = mf
        for li, (layer, wg) in enumerate(zip(P['mask_convs'], P['mask_convs_wg'])):
            scale = vmask if li == 0 else None         # support-vector guidance (fgn_roi_head.py:379) fused into conv 0
            if wg is not None and ops.winograd_pays(m.shape[0], m.shape[1], m.shape[2], wg.cin, wg.cout, wg.m):
                m = ops.conv3x3_winograd(m, wg, in_scale=scale, n_img_dev=n_dev)
            else:
                m = ops.conv2d(m, layer, in_scale=scale, n_img_dev=n_dev)
        up = ops.conv2d(m, P['upsample'], n_img_dev=n_dev)                         # [D,7,7,4*C']
        return ops.mask_logits(up, P['logit_w'], P['logit_b'], self.cfg['roi_head']['roi_out_size'], n_dev,
                               prob_out=prob_out)

    def forward(self, return_loss=True, **kwargs):
        if return_loss:
            return self.forward_train(**kwargs)
        return self.simple_test(**kwargs)

    @torch.no_grad()
    def forward_train(self, qry_img, qry_bboxes, qry_cat_ids, qry_isegmaps, qry_bboxes_ignore=None, proposals=None,
                      spp_imgs=None, spp_bboxes=None, spp_isegmaps=None, img_shape=None, **kwargs) -> Dict:
        """The loss dict of one training step (fgn.py:125-185) as forward values - see ``fgn_amd.train``.  Keys and
        container types are the reference's: ``loss_rpn_cls`` / ``loss_rpn_bbox`` (lists of one tensor),
        ``loss_cls``, ``ACC-Unbalanced``, ``ACC-Balanced``, ``loss_bbox``, ``loss_mask``."""
        from . import train
        return train.forward_train(self, qry_img, qry_bboxes, qry_cat_ids, qry_isegmaps,
                                   qry_bboxes_ignore=qry_bboxes_ignore, proposals=proposals, spp_imgs=spp_imgs,
                                   spp_bboxes=spp_bboxes, spp_isegmaps=spp_isegmaps, img_shape=img_shape, **kwargs)

    @torch.no_grad()
    def simple_test(self, qry_img, qry_bboxes=None, qry_cat_ids=None, qry_isegmaps=None, qry_bboxes_ignore=None,
                    spp_imgs=None, spp_bboxes=None, spp_isegmaps=None, qry_child_idx=None, img_shape=None,
                    rescale=False, cats_ids_to_sample_real=None, spp_insts_ids=None, idx=None,
                    support_code=None, **kwargs) -> List[Dict]:
        """Test without augmentation (fgn.py:187-303).  ``support_code`` (optional, from
        ``encode_supports``) replaces the three ``spp_*`` inputs."""
        dets = self.detect_device(qry_img, spp_imgs, spp_bboxes, spp_isegmaps, img_shape, support_code,
                                  qry_isegmaps=qry_isegmaps)
        return self.pack_results(dets, qry_img.shape[0], qry_bboxes=qry_bboxes, qry_cat_ids=qry_cat_ids,
                                 qry_isegmaps=qry_isegmaps, img_shape=img_shape, qry_child_idx=qry_child_idx,
                                 cats_ids_to_sample_real=cats_ids_to_sample_real, spp_insts_ids=spp_insts_ids,
                                 idx=idx)

    # --- support branch (fgn.py:212-215 backbone pass; fgn_ag_rpn_head.py:38-41; fgn_roi_head.py:419-449) ---
    def _support_front(self, spp_imgs, spp_bboxes, spp_isegmaps, B, dev, stream, defer_backbone: bool = False) -> dict:
        """modify_input for the supports (fgn.py:79-108: H2D, YXYX -> XYXY on private copies), their
        backbone pass and the AG-RPN class vectors."""
        N, K = self.n_ways, self.k_shots
        spp = spp_imgs.to(dev, torch.float32, non_blocking=True).reshape(B * N * K, *spp_imgs.shape[-3:])
        b = spp_bboxes.to(dev, torch.float32, non_blocking=True).reshape(B * N * K, 4)
        spp_xyxy = torch.stack((b[:, 1], b[:, 0], b[:, 3], b[:, 2]), 1)      # no host-built index tensor: graph-capturable
        m = spp_isegmaps.to(dev, non_blocking=True).reshape(B * N * K, *spp_isegmaps.shape[-2:])
        spp_masks = (m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)).contiguous()
        sc = dict(B=B, device=dev, spp_xyxy=spp_xyxy, spp_masks=spp_masks, spp=spp)
        if not defer_backbone:
            self._support_vectors(sc, self.extract_feat(sc.pop('spp')))         # [B*N*K,s,s,C]
        return sc

    def _support_vectors(self, sc: dict, spp_fmaps) -> None:
        sc['spp_fmaps'] = spp_fmaps
        sc['vec'] = ops.support_class_vectors(spp_fmaps, None, sc['B'] * self.n_ways, self.k_shots)   # [B*N,C]

    def _support_back(self, sc: dict, B, dev, shared=None) -> None:
        """count_spp (fgn_roi_head.py:419-449) and the support half of the relation conv.  ``shared``: the shared-head
        callable (forward_train passes the train-mode BatchNorm variant)."""
        P, rh = self._P, self.cfg['roi_head']
        N, K, PS = self.n_ways, self.k_shots, rh['roi_out_size']
        bidx = torch.arange(B * N * K, device=dev, dtype=torch.float32)[:, None]
        spp_rois = torch.cat([bidx, sc['spp_xyxy']], 1).contiguous()
        sc['masks7'] = ops.roi_align_mask(sc['spp_masks'], spp_rois, PS, 1.0, -1, False)
        # `spp_bboxes /= 16` then roi_align(scale=1) == roi_align(scale=1/16): /16 is exact in fp32
        sfeat = ops.roi_align(sc['spp_fmaps'], spp_rois, PS, 1.0 / rh['featmap_stride'], -1, False)
        sfeat = self._shared_head(sfeat) if shared is None else shared(sfeat)
        sc['cat_mean'] = ops.support_kmean(sfeat, B * N, K)                           # [B*N,7,7,C]
        sc['cat_mean_mp'] = ops.support_class_vectors(sfeat, sc['masks7'], B * N, K)  # [B*N,C]
        sc['S'] = ops.conv2d(sc['cat_mean'], P['rel_s'])                              # Ws*support + bias

    def _support_rois(self, sc: dict, B, dev, x_out, y1_out) -> None:
        """First half of count_spp for the MERGED shared head: the support masks and maps are pooled (into the first rows
        of the RoI batch the box head will run, ``x_out``), and - the first 1x1 conv of the shared head being taken on
        the feature map for the query's RoIs - the same conv on the pooled support features fills their rows of
        ``y1_out``."""
        P, rh = self._P, self.cfg['roi_head']
        N, K, PS = self.n_ways, self.k_shots, rh['roi_out_size']
        bidx = torch.arange(B * N * K, device=dev, dtype=torch.float32)[:, None]
        spp_rois = torch.cat([bidx, sc['spp_xyxy']], 1).contiguous()
        sc['masks7'] = ops.roi_align_mask(sc['spp_masks'], spp_rois, PS, 1.0, -1, False)
        ops.roi_align(sc['spp_fmaps'], spp_rois, PS, 1.0 / rh['featmap_stride'], -1, False, out=x_out)
        if y1_out is not None:
            ops.conv2d(x_out, P['shared'][0].conv1, out=y1_out)

    def _support_finish(self, sc: dict, sfeat, B) -> None:
        """Second half of count_spp (fgn_roi_head.py:439-447) + the support half of the relation conv, on the support
        rows of the shared head's output."""
        P, N, K = self._P, self.n_ways, self.k_shots
        sc['cat_mean'] = ops.support_kmean(sfeat, B * N, K)                           # [B*N,7,7,C]
        sc['cat_mean_mp'] = ops.support_class_vectors(sfeat, sc['masks7'], B * N, K)  # [B*N,C]
        sc['S'] = ops.conv2d(sc['cat_mean'], P['rel_s'])                              # Ws*support + bias

    @torch.no_grad()
    def encode_supports(self, spp_imgs, spp_bboxes, spp_isegmaps) -> dict:
        """Support-feature caching across queries: everything the path derives
        from the support set alone - backbone pass, AG-RPN class vectors, count_spp, the support
        half of the relation conv - computed once on the current stream.  The returned code is
        passed as ``support_code=`` to ``simple_test`` / ``detect_device`` for every query that
        shares the support set (the reference recomputes it per query, fgn.py:212-215; results
        are identical)."""
        if not torch.cuda.is_available():
            raise ops._lib.FgnHipError('FGN.encode_supports needs a GPU: the HIP path has no CPU fallback')
        dev = torch.device('cuda', torch.cuda.current_device())
        if self._packed_device != dev:
            self._pack(dev)
        self._sync_trained_shared()
        B = spp_imgs.shape[0] if spp_imgs.dim() == 5 else 1
        sc = self._support_front(spp_imgs, spp_bboxes, spp_isegmaps, B, dev, torch.cuda.current_stream())
        self._support_back(sc, B, dev)
        return sc

    def _sync_trained_shared(self) -> None:
        """A ``fgn_amd.train.Trainer`` updates the head layers in place after every step; the inference form of the
        shared head (eval-mode BatchNorm folded from the CURRENT weights and running statistics) is rebuilt here, on
        the first inference call after a training step, so ``simple_test`` always sees the trainer's weights."""
        src = getattr(self, '_shared_dirty', None)
        if src is not None:
            for k, v in self._pack_shared(src).items():
                self._P[k] = v
            self._shared_dirty = None
            self._graphs = {}

    def _stream_for(self, role: str, main) -> 'torch.cuda.Stream':
        """One auxiliary HIP stream per (role, caller stream), or what ``transfer_mode`` prescribes for the 'upload' and
        'copy' roles (``transfer_stream``).

        HIP maps streams onto a small pool of hardware queues (GPU_MAX_HW_QUEUES, 4 by default; the k-th stream created
        takes queue k mod 4 once the pool is full), work of streams that share a queue runs in order, and a pool larger
        than 4 is time-sliced by the firmware: measured r04 (profiles/r04_hw_queues.txt) 2 / 3 / 4 / 5 / 6 / 8 queues
        give 170 / 172 / 190 / 114 / 142 / 138 img/s, and transfer streams in the high-priority pool (= more active
        queues) 140.  So the budget is four queues, and which streams share one matters."""
        mode = self.transfer_mode
        if (mode in (2, 3) and role == 'copy') or (mode == 3 and role == 'upload'):
            return main                                   # the transfers of an episode ride on its caller stream
        if (mode == 1 and role in ('upload', 'copy')) or (mode == 2 and role == 'upload'):
            key = ('xfer', 0)                             # one stream for every caller
        else:
            key = (role, main.cuda_stream)
        st = self._streams.get(key)
        if st is None:
            st = self._streams[key] = torch.cuda.Stream()
        return st

    def transfer_stream(self, mode: int = 2) -> 'torch.cuda.Stream':
        """Select a transfer arrangement and create its shared stream NOW - a serving loop calls this BEFORE it creates
        its caller streams, so that the shared stream and each of up to three caller streams get a hardware queue of
        their own.  mode 2: the result copies of an episode ride on its caller stream (no copy stream: a stream that
        waits for an episode's results blocks whatever shares its queue), one upload stream for all callers; mode 1: one
        stream for uploads AND result copies (measured r04: serialises the episodes, 190 -> 159 img/s - an upload queued
        behind a copy that waits for the previous episode holds up the next); mode 0: one upload and one copy stream
        per caller stream (round 3)."""
        self.transfer_mode = int(mode)
        return self._stream_for('upload', torch.cuda.current_stream())

    def _upload(self, tensors: dict, gt_masks, dev, main, into: Optional[dict] = None):
        """``modify_input`` (fgn.py:79-108): host -> device copies of one batch, on an upload stream so that they
        overlap the previous batch's kernels (a pinned source makes them asynchronous); the compute streams wait
        on one event.  Tensors already on the device pass through.  The ground-truth masks (copied to the GPU by
        the reference too, fgn.py:95) are run-length encoded right there (``qry_isegmaps_rle``, fgn.py:298): two
        small kernels on the upload stream instead of ~4 ms of host work per 800x1333 episode.  ``into``: device
        tensors to copy INTO (the static input buffers of a captured graph, when the upload stream is the caller
        stream itself and so ordered behind the previous replay that reads them) instead of fresh allocations."""
        gts = None
        if gt_masks is not None:
            gts = [g if isinstance(g, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(g)) for g in gt_masks]
        on_host = [t for t in list(tensors.values()) + (gts or []) if t is not None and not t.is_cuda]
        if not on_host and gts is None:
            return tensors, None, None
        up = self._stream_for('upload', main)
        if not on_host:
            up.wait_stream(main)               # device-resident inputs may still be being produced on the caller's stream
        out, gt_out = {}, None
        with torch.cuda.stream(up):
            for k, t in tensors.items():
                if t is not None and into is not None and k in into and not t.is_cuda and into[k].dtype == t.dtype \
                        and into[k].shape == t.shape:
                    into[k].copy_(t, non_blocking=True)
                    out[k] = into[k]
                else:
                    out[k] = None if t is None else t.to(dev, non_blocking=True)
            if gts is not None:
                gt_out = []
                for g in gts:
                    g = g.to(dev, non_blocking=True)
                    g = g if g.dtype in (torch.bool, torch.uint8) else (g != 0)
                    gt_out.append(ops.dense_mask_rle(g.contiguous(), packed=True))
            ready = up.record_event()
        for k, t in out.items():
            if t is not None and not (into is not None and t is into.get(k)):
                t.record_stream(main)
        return out, gt_out, ready

    @torch.no_grad()
    def detect_device(self, qry_img, spp_imgs, spp_bboxes, spp_isegmaps, img_shape, support_code=None,
                      qry_isegmaps=None, phase_counter=None) -> list:
        """Everything up to the host wait: queues the host->device copies, the whole path and the device->host
        copies of the results.  Returns, per image, a dict of device tensors (det_bboxes [D,5], det_labels [D],
        n_dets [1], mask_prob, RLE bytes) plus the pinned host slot ``pack_results`` reads.
        With ``support_code`` (from ``encode_supports``) the support branch is skipped.  ``qry_isegmaps`` (list of
        [n_i,H,W] bool): the ground-truth masks, RLE-encoded on the device for ``qry_isegmaps_rle``.  With
        ``use_graphs`` the launch sequence of one input geometry is captured once into a hipGraph and replayed
        (same kernels, same results; ~0.2 ms of host time instead of ~2 ms).  ``phase_counter``: the one-element int32
        device counter THIS call's episode bumps at ``phase_point`` (see ``_phase_mark``); passed per call, so two
        caller threads never see each other's counter (the attribute of the same name is the fallback for callers that
        set it on the model)."""
        if not torch.cuda.is_available():
            raise ops._lib.FgnHipError('FGN.simple_test needs a GPU: the HIP path has no CPU fallback')
        dev = torch.device('cuda', torch.cuda.current_device())
        main = torch.cuda.current_stream()
        ins = {'qry_img': qry_img}
        if support_code is None:
            ins.update(spp_imgs=spp_imgs, spp_bboxes=spp_bboxes, spp_isegmaps=spp_isegmaps)
        graphed = self.use_graphs and self.debug_trace is None and ops.PROFILE is None
        if phase_counter is None:
            phase_counter = self.phase_counter
        # graph replay with the transfers on the caller stream: host tensors go straight into the graph's static input
        # buffers (the caller stream orders the copy behind the previous replay that reads them) - no device-to-device hop
        into = None
        if graphed and self.use_packed_transfers and self._stream_for('upload', main) is main:
            ge0 = self._graphs.get(self._graph_key(ins, img_shape, support_code, phase_counter, main, dev))
            into = ge0.static if ge0 is not None else None
        ins, gt_rle, uploaded = self._upload(ins, qry_isegmaps, dev, main, into=into)
        if uploaded is not None:
            main.wait_event(uploaded)
        if graphed:
            ge, outs = self._detect_graphed(ins, img_shape, support_code, phase_counter)
        else:
            outs = self._detect_eager(ins['qry_img'], ins.get('spp_imgs'), ins.get('spp_bboxes'),
                                      ins.get('spp_isegmaps'), img_shape, support_code, phase_counter=phase_counter)
        if gt_rle is not None:
            for d, g in zip(outs, gt_rle):
                d['gt_rle'] = g
        self._start_download(outs, main, uploaded if gt_rle is not None else None)
        if graphed:
            ge.last_download = outs[0]['host_ready']
        return outs

    def _graph_key(self, ins: dict, img_shape, support_code, phase_counter, main, dev) -> tuple:
        hw = tuple((int(s[0]), int(s[1])) for s in img_shape)
        # everything the captured launch sequence depends on besides the weights (those drop ``_graphs`` when they
        # change): the paste semantic is an argument of the captured RLE kernel, the transfer arrangement decides which
        # streams the capture forks, and the phase mark is a captured kernel with the counter's address in its arguments
        self._skip_empty()
        mark = None if (phase_counter is None or not self.phase_point) else (phase_counter.data_ptr(), self.phase_point)
        return (main.cuda_stream, dev.index, hw, support_code is not None, bool(self.use_merged_backbone),
                bool(self.use_merged_support_head), self.paste_semantics, int(self.transfer_mode), mark,
                bool(self.use_side_stream), bool(self.use_packed_transfers)) + \
            tuple((k, tuple(v.shape), v.dtype) for k, v in ins.items() if v is not None)

    def _detect_graphed(self, ins: dict, img_shape, support_code, phase_counter=None):
        main = torch.cuda.current_stream()
        dev = torch.device('cuda', torch.cuda.current_device())
        key = self._graph_key(ins, img_shape, support_code, phase_counter, main, dev)
        ge = self._graphs.get(key)
        if ge is None:
            ge = self._graphs[key] = _GraphedEpisode(self, ins, img_shape, support_code, dev, phase_counter)
        return ge, ge.run(self, ins, support_code, main)

    def _detect_eager(self, qry_img, spp_imgs, spp_bboxes, spp_isegmaps, img_shape, support_code=None,
                      phase_counter=None) -> list:
        dev = torch.device('cuda', torch.cuda.current_device())
        if self._packed_device != dev:
            self._pack(dev)
        self._sync_trained_shared()
        with ops.arena(dev):     # zero-initialised small outputs of this episode: one fill (caller's stream only)
            return self._detect_body(qry_img, spp_imgs, spp_bboxes, spp_isegmaps, img_shape, support_code, dev,
                                     phase_counter)

    # Phase mark of a pipelined serving loop (bench.py): when two caller streams replay episodes side by
    # side, their relative phase settles in one of several steady states that differ by ~4 % in throughput.  A caller may
    # hand over a one-element int32 device counter (``phase_counter``) and the point of the episode at which to bump it
    # (``phase_point``: 'layer1'..'layer3' inside the backbone, 'rpn' = backbone done, 'rpn_conv', 'proposals', 'mask'):
    # ``ops.phase_signal`` is captured into the episode's
    # graph like any kernel, and the OTHER caller stream runs ``ops.phase_wait`` on that counter before its next episode.
    # (An event would be the natural tool; a captured graph cannot record one that another stream waits for on this
    # stack - "External events are disallowed in rocm".)  None: no mark.  The counter travels as an ARGUMENT of
    # ``detect_device`` (and is part of the graph cache key: a replayed graph bumps the counter it was captured with);
    # ``phase_counter`` on the model is the fallback for callers that set it there.
    phase_counter = None
    phase_point = None

    def _phase_mark(self, name: str, counter) -> None:
        if counter is not None and name == self.phase_point:
            ops.phase_signal(counter)

    def _detect_body(self, qry_img, spp_imgs, spp_bboxes, spp_isegmaps, img_shape, support_code, dev,
                     phase_counter=None) -> list:
        P, cfg = self._P, self.cfg
        N, K = self.n_ways, self.k_shots
        tr = self.debug_trace
        B, _, H, W = qry_img.shape
        rh, rp, tc = cfg['roi_head'], cfg['rpn_head'], cfg['test_cfg']
        PS = rh['roi_out_size']
        inv_stride = 1.0 / rh['featmap_stride']

        qry = qry_img.to(dev, torch.float32, non_blocking=True)

        # Two HIP streams: the support branch (9 small crops: low-occupancy launches) runs beside
        # the query branch, and its RoI/shared-head/reduction tail runs beside the single-workgroup
        # proposal kernel.  Joined by events; no host synchronisation.
        main = torch.cuda.current_stream()
        cached = support_code is not None
        if cached:
            if support_code['B'] != B or support_code['device'] != dev:
                raise ValueError('support_code was encoded for another batch size or device')
            side = main
        elif self.use_side_stream:
            side = self._stream_for('side', main)              # one side stream per caller stream
        else:
            side = main
        merged = (not cached) and self.use_merged_backbone and self.cfg['backbone'].get('norm', 'BN') == 'BN' and \
            self.cfg['backbone'].get('block', 'bottleneck') == 'bottleneck'
        # merged shared head: rows [0, ns) of the box head's RoI batch are the support RoIs
        msh = (not cached) and self.use_merged_support_head
        ns = B * N * K
        x_all = y1_all = None
        if msh:
            c4 = rh['shared_head']['inplanes']
            x_all = torch.empty((ns + B * tc['rpn']['max_per_img'], PS, PS, c4), device=dev, dtype=torch.float32)
            if P['sh0_lin'] is not None:
                y1_all = torch.empty(tuple(x_all.shape[:3]) + (P['sh0_lin'].cout,), device=dev, dtype=torch.float32)
            if side is not main and not torch.cuda.is_current_stream_capturing():
                x_all.record_stream(side)
                if y1_all is not None:
                    y1_all.record_stream(side)

        def support_tail():       # on the side stream, right behind the support maps
            if msh:
                self._support_rois(sc, B, dev, x_all[:ns], None if y1_all is None else y1_all[:ns])
            else:
                self._support_back(sc, B, dev)
        qry_fmap = None
        if cached:
            sc = support_code
        elif merged:
            sc = self._support_front(spp_imgs, spp_bboxes, spp_isegmaps, B, dev, main, defer_backbone=True)
            qry_fmap, spp_fmaps = self.extract_feat_pair(qry, sc.pop('spp'), phase_counter)
            backbone_done = main.record_event()
            with torch.cuda.stream(side):
                side.wait_event(backbone_done)
                self._support_vectors(sc, spp_fmaps)
                vec_ready = side.record_event()
                support_tail()
            if side is not main and not torch.cuda.is_current_stream_capturing():
                spp_fmaps.record_stream(side)
        else:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                sc = self._support_front(spp_imgs, spp_bboxes, spp_isegmaps, B, dev, side)
                vec_ready = side.record_event()
                # count_spp (fgn_roi_head.py:419-449) follows at once: its ~30 tiny 9-RoI launches run beside the
                # second half of the query backbone.  (Released later, at the AG-RPN conv, they were starved by that
                # conv's persistent workgroups and the RoI head waited ~0.1 ms for them.)
                support_tail()
        vec = sc['vec']

        if qry_fmap is None:
            qry_fmap = self.extract_feat(qry)                   # [B,h,w,C]
        fh, fw, C = qry_fmap.shape[1:]
        if tr is not None:
            tr['qry_fmap'], tr['spp_fmaps'] = qry_fmap, sc.get('spp_fmaps')

        # ---- AG-RPN (fgn_ag_rpn_head.py:26-118) -------------------------------------------
        if not cached:
            main.wait_event(vec_ready)
        rpn_start = main.record_event()
        self._phase_mark('rpn', phase_counter)
        # guidance multiply (fgn_ag_rpn_head.py:44): never materialised - it rides in the Winograd input transform,
        # or in the A-operand staging of the direct kernel where the layer is too small for the Winograd form
        if P['rpn_conv_wg'] is not None and ops.winograd_fits(B * N, fh, fw, C, P['rpn_conv_wg'].cout,
                                                              P['rpn_conv_wg'].m):
            # Winograd F(2x2,3x3); the guidance multiply rides in its input transform
            x = ops.conv3x3_winograd(qry_fmap, P['rpn_conv_wg'], in_scale=vec, a_img_div=N)
        else:
            x = ops.conv2d(ops.scale_channels(qry_fmap, vec, N), P['rpn_conv'])
        self._phase_mark('rpn_conv', phase_counter)
        head = ops.conv2d(x, P['rpn_head'])                                     # [B*N,h,w,5A]
        A = P['anchors'].shape[0]
        logits, scores, deltas = ops.rpn_merge(head, B, N, A)

        # ---- shared_head conv1 on the query map (see _roi_feats): independent of the proposals, so it runs on the
        # side stream beside the AG-RPN transforms and the proposal kernels instead of after them
        g_map = None
        if not cached:
            with torch.cuda.stream(side):
                if P['sh0_lin'] is not None and side is not main:
                    side.wait_event(rpn_start)
                    g_map = ops.conv2d(qry_fmap, P['sh0_lin'])
                spp_ready = side.record_event()
            if not torch.cuda.is_current_stream_capturing():
                for key in ('spp_fmaps', 'vec', 'masks7') if msh else ('spp_fmaps', 'vec', 'S', 'cat_mean', 'cat_mean_mp', 'masks7'):
                    sc[key].record_stream(main)                                   # produced on side, consumed on main
                if g_map is not None:
                    g_map.record_stream(main)
                    qry_fmap.record_stream(side)
        ih, iw = int(img_shape[0][0]), int(img_shape[0][1])
        if any(int(s[0]) != ih or int(s[1]) != iw for s in img_shape):
            raise ValueError('all images of a batch must share img_shape (the dataset batches by size)')
        props, n_props, rois_all = ops.rpn_proposals(scores, deltas, P['anchors'], fh, fw, rp['anchor_stride'], ih, iw,
                                                     rp['target_means'], rp['target_stds'], tc['rpn']['nms_pre'],
                                                     tc['rpn']['min_bbox_size'], tc['rpn']['nms_iou_threshold'],
                                                     tc['rpn']['max_per_img'], with_rois=True)   # [B*R,5] = bbox2roi
        if not cached:
            main.wait_event(spp_ready)
        if tr is not None:
            tr.update(class_vec=vec, rpn_logits=logits, rpn_scores=scores, rpn_deltas=deltas, proposals=props,
                      n_props=n_props)

        self._phase_mark('proposals', phase_counter)
        # ---- box head on the proposals of all B images at once (fgn_roi_head.py:531-616); a RoI carries its
        # image index in column 0 (bbox2roi), which selects the feature map in RoIAlign and the support set
        # in the relation head.  With one image the device-side proposal count bounds every launch; with
        # several images the valid RoIs are not a prefix of the list, so the zero-box padding rows are
        # computed too and dropped per image by det_post.
        rel, bh = rh['relation'], rh['bbox_head']
        R, D = props.shape[1], tc['rcnn']['max_per_img']
        cnt_all = n_props[0:1] if B == 1 else None
        if g_map is None and P['sh0_lin'] is not None:
            g_map = ops.conv2d(qry_fmap, P['sh0_lin'])                                # [B,h,w,planes]
        if msh:
            # support RoIs (pooled on the side stream) and proposals through the shared head as ONE RoI batch
            inv = 1.0 / rh['featmap_stride']
            ops.roi_align(qry_fmap, rois_all, PS, inv, rh['roi_sampling_ratio'], True, cnt_all, out=x_all[ns:])
            if y1_all is not None:
                ops.roi_align(g_map, rois_all, PS, inv, rh['roi_sampling_ratio'], True, cnt_all,
                              post_shift=P['sh0_shift'], relu=True, out=y1_all[ns:])
            cnt_plus = None if cnt_all is None else cnt_all + ns
            feats_all = self._shared_head(x_all, cnt_plus, y1=y1_all)
            self._support_finish(sc, feats_all[:ns], B)
            feats = feats_all[ns:]
        else:
            _, feats = self._roi_feats(qry_fmap, g_map, rois_all, cnt_all)
        S, cat_mean, cat_mean_mp, masks7 = sc['S'], sc['cat_mean'], sc['cat_mean_mp'], sc['masks7']
        if tr is not None:
            tr.update(spp_masks7=masks7, spp_cat_mean=cat_mean, spp_cat_mean_mp=cat_mean_mp)
        Q = ops.conv2d(feats, P['rel_q'], n_img_dev=cnt_all)
        cls_raw, reg_raw = ops.relation_gn_head(Q, S, rois_all, P['gn_w'], P['gn_b'], P['fc_w'], P['fc_b'], N,
                                                rel['gn_groups'], rel['gn_eps'], cnt_all)
        # the batch's result record: detections, labels, counts, mask probabilities and RLE strings land in one buffer
        rec = _ResultRecord(B, D, 2 * PS, dev)
        rec.clear_head()
        packed = self.use_packed_transfers
        # one selection workgroup per image, one launch for the batch
        det_all, lab_all, n_det_all, mrois_all = ops.det_post(
            rois_all, cls_raw, reg_raw, N, ih, iw, bh['target_means'], bh['target_stds'], tc['rcnn']['score_thr'],
            tc['rcnn']['nms_iou_threshold'], D, n_props, img_index=0, batch=B,
            out=(rec.det.view(B * D, 5), rec.lab.view(B * D), rec.cnt.view(B)))      # mrois_all [B*D,5] = bbox2roi
        dets = [det_all[i * D:(i + 1) * D] for i in range(B)]
        labs = [lab_all[i * D:(i + 1) * D] for i in range(B)]
        n_dets = [n_det_all[i:i + 1] for i in range(B)]
        self._phase_mark('mask', phase_counter)
        # ---- mask head on the detections of all images at once (fgn_roi_head.py:704-718, 360-382)
        nd_all = n_dets[0] if B == 1 else None
        vmask = ops.gather_support_vectors(cat_mean_mp, lab_all, mrois_all, N, nd_all)
        _, mf = self._roi_feats(qry_fmap, g_map, mrois_all, nd_all)
        mlog, mprob = self._mask_head(mf, vmask, nd_all, prob_out=rec.prob.view(B * D, 2 * PS, 2 * PS))
        outs = []
        for i in range(B):
            det, lab, n_det = dets[i], labs[i], n_dets[i]
            mp_i = mprob[i * D:(i + 1) * D]
            # paste + threshold + COCO RLE fused on device: the D x H x W masks are never written
            skip_empty = self._skip_empty()
            rle_bytes, rle_len, rle_ovf = ops.mask_rle(mp_i, det, ih, iw, tc['rcnn']['mask_thr_binary'], n_det,
                                                       skip_empty=skip_empty,
                                                       out=(rec.rle[i], rec.rle_len[i], rec.rle_ovf[i]))
            if tr is not None:
                masks = ops.mask_paste(mp_i, det, ih, iw, tc['rcnn']['mask_thr_binary'], n_det, skip_empty=skip_empty)
                tr.setdefault('per_image', []).append(dict(
                    rois=rois_all[i * R:(i + 1) * R], roi_feats=feats[i * R:(i + 1) * R], Q=Q[i * R:(i + 1) * R],
                    cls_raw=cls_raw[i * R * N:(i + 1) * R * N], reg_raw=reg_raw[i * R * N:(i + 1) * R * N],
                    det=det, lab=lab, n_det=n_det, mask_logits=mlog[i * D:(i + 1) * D], mask_prob=mp_i,
                    masks=masks, mask_feats=mf[i * D:(i + 1) * D]))
            outs.append(dict(det_bboxes=det, det_labels=lab, n_dets=n_det, mask_prob=mp_i, rle_bytes=rle_bytes,
                             rle_len=rle_len, rle_overflow=rle_ovf, img_hw=(ih, iw), record=rec if packed else None))
        return outs

    MAX_SLOTS = 64

    def _pinned_slot(self, batch: int, max_det: int, n_gt: int, mask_size: int = 14) -> dict:
        """A free pinned host slot for one batch's results (pinned allocation is slow, so slots are kept per
        (batch, max_det, mask size) and reused): the host mirror of a ``_ResultRecord`` plus a byte buffer for the
        packed ground-truth RLE.  A slot is busy from ``detect_device`` until ``pack_results`` has read it; when every
        slot is busy (more batches in flight than ever before) another one is allocated."""
        key = (batch, max_det, mask_size, ops.RLE_BYTE_CAP)
        ring = self._pinned.setdefault(key, [])
        slot = next((s for s in ring if not s['busy']), None)
        if slot is None:
            if len(ring) >= self.MAX_SLOTS:
                raise ops._lib.FgnHipError(f'{self.MAX_SLOTS} batches are in flight without pack_results(); '
                                           'pack (or drop and call release_results on) earlier detect_device outputs')
            rec = _ResultRecord(batch, max_det, mask_size, pinned=True)
            slot = dict(key=key, busy=False, gt_cap=0, rec=rec, det=rec.det, lab=rec.lab, cnt=rec.cnt, rle_len=rec.rle_len,
                        rle_ovf=rec.rle_ovf, rle=rec.rle, prob=rec.prob)
            ring.append(slot)
        if n_gt > slot['gt_cap']:
            # generous from the start: growing a slot is a pinned allocation (~5 ms of host time) in the middle of a
            # pipelined run - with caps of 2 x the first count seen, slots kept growing for dozens of steps
            cap = max(64, 2 * n_gt)
            slot.update(gt_cap=cap, gt_buf=torch.empty(cap * (ops.RLE_BYTE_CAP + 8), dtype=torch.uint8, pin_memory=True))
        slot['busy'] = True
        return slot

    @staticmethod
    def release_results(dets: list) -> None:
        """Give back the pinned host slot of ``detect_device`` outputs that will not be packed."""
        if dets and 'host' in dets[0]:
            dets[0]['host']['busy'] = False

    def _start_download(self, outs: list, main, also_wait=None) -> None:
        """Queue the device->host copies of one batch on a copy stream behind the compute work;
        ``pack_results`` later waits on the event only, so the next batch's kernels are not
        serialised behind a host round trip."""
        cp = self._stream_for('copy', main)
        max_det = outs[0]['det_bboxes'].shape[0]
        n_gt = sum(d['gt_rle'][1].shape[0] for d in outs if 'gt_rle' in d)
        rec = outs[0].get('record')
        slot = self._pinned_slot(len(outs), max_det, n_gt, outs[0]['mask_prob'].shape[-1])
        cp.wait_stream(main)
        if also_wait is not None:
            cp.wait_event(also_wait)           # the ground-truth RLE kernels run on the upload stream
        with torch.cuda.stream(cp):
            if rec is not None and rec.key == slot['rec'].key:
                # ONE copy for the batch: detections, labels, counts, RLE strings and lengths, mask probabilities
                slot['rec'].buf.copy_(rec.buf, non_blocking=True)
                rec.buf.record_stream(cp)
            else:                  # outputs assembled by hand (tests): field by field
                for i, d in enumerate(outs):
                    for dst, src in (('det', 'det_bboxes'), ('lab', 'det_labels'), ('cnt', 'n_dets'), ('rle_len', 'rle_len'),
                                     ('rle_ovf', 'rle_overflow'), ('rle', 'rle_bytes'), ('prob', 'mask_prob')):
                        slot[dst][i].copy_(d[src].reshape(slot[dst][i].shape), non_blocking=True)
                        d[src].record_stream(cp)
            g0 = 0
            row = ops.RLE_BYTE_CAP + 8
            for d in outs:
                if 'gt_rle' in d:
                    n = d['gt_rle'][1].shape[0]
                    if n:
                        if len(d['gt_rle']) == 4:        # packed [lens | overflow | bytes]: one copy per image
                            slot['gt_buf'][g0 * row:(g0 + n) * row].copy_(d['gt_rle'][3], non_blocking=True)
                            d['gt_rle'][3].record_stream(cp)
                        else:
                            hb = slot['gt_buf'][g0 * row:(g0 + n) * row]
                            gb, gl, go = d['gt_rle'][:3]
                            hb[:n * 4].view(torch.int32).copy_(gl, non_blocking=True)
                            hb[n * 4:n * 8].view(torch.int32).copy_(go, non_blocking=True)
                            hb[n * 8:].view(n, ops.RLE_BYTE_CAP).copy_(gb, non_blocking=True)
                            for t in (gb, gl, go):
                                t.record_stream(cp)
                    d['gt_slice'] = (g0, n)
                    g0 += n
            ev = cp.record_event()
        for d in outs:
            d['host'] = slot
            d['host_ready'] = ev

    def pack_results(self, dets: list, batch: int, qry_bboxes=None, qry_cat_ids=None, qry_isegmaps=None,
                     img_shape=None, qry_child_idx=None, cats_ids_to_sample_real=None, spp_insts_ids=None,
                     idx=None) -> List[Dict]:
        """Device->host copy and result dicts (fgn.py:240-303).  Passthrough boxes stay YXYX
        (SERVER semantics); caller tensors are never mutated."""
        passthrough = {'idx': idx, 'qry_bboxes': qry_bboxes, 'qry_img_shape': img_shape,
                       'qry_cat_ids': qry_cat_ids, 'qry_child_idx': qry_child_idx,
                       'cats_ids_to_sample_real': cats_ids_to_sample_real, 'spp_insts_ids': spp_insts_ids}
        dets[0]['host_ready'].synchronize()        # the one host wait: the batch's D2H copies
        host = dets[0]['host']
        results = []
        for i in range(batch):
            di = dets[i]
            n = int(host['cnt'][i, 0])
            db = host['det'][i, :n].numpy()
            lens = host['rle_len'][i, :n].numpy()
            ovf = host['rle_ovf'][i, :n].numpy()
            ih, iw = di['img_hw']
            strings = host['rle'][i].numpy()
            rles = [{'size': [ih, iw], 'counts': strings[j, :lens[j]].tobytes()} for j in range(n)]
            if n and ovf.any():     # a device cap overflowed: dense paste + host RLE for those masks only
                thr = self.cfg['test_cfg']['rcnn']['mask_thr_binary']
                for j in np.flatnonzero(ovf):
                    # (from the HOST copies: a replayed graph may have overwritten the device tensors by now)
                    pdev = di['det_bboxes'].device
                    dense = ops.mask_paste(host['prob'][i, j:j + 1].to(pdev).contiguous(),
                                           host['det'][i, j:j + 1].to(pdev).contiguous(),
                                           ih, iw, thr, skip_empty=self._skip_empty())
                    rles[j] = rle.encode(dense[0].cpu().numpy())
            one = {'dt_scores': db[:, 4].reshape(-1).copy(),
                   'dt_bboxes': db[:, [1, 0, 3, 2]].reshape(-1, 4).copy(),
                   'dt_cat_ids': host['lab'][i, :n].numpy().copy().reshape(-1),
                   'dt_isegmaps_rle': rles}
            for key, val in passthrough.items():
                v = val[i] if val is not None else None
                one[key] = v.cpu().numpy() if isinstance(v, torch.Tensor) else v
            gt = qry_isegmaps[i] if qry_isegmaps is not None else None
            if 'gt_slice' in di:                   # ground-truth masks were encoded on the device (detect_device)
                g0, ng = di['gt_slice']
                row = ops.RLE_BYTE_CAP + 8
                hb = host['gt_buf'][g0 * row:(g0 + ng) * row]
                glen, govf = hb[:ng * 4].view(torch.int32).numpy(), hb[ng * 4:ng * 8].view(torch.int32).numpy()
                gstr = hb[ng * 8:].view(ng, ops.RLE_BYTE_CAP).numpy()
                one['qry_isegmaps_rle'] = [
                    {'size': [ih, iw], 'counts': gstr[j, :glen[j]].tobytes()} if not govf[j] else
                    rle.encode(np.asarray(gt[j].cpu() if isinstance(gt[j], torch.Tensor) else gt[j]))
                    for j in range(ng)]
            elif gt is not None:
                gt = gt.cpu().numpy() if isinstance(gt, torch.Tensor) else np.asarray(gt)
                one['qry_isegmaps_rle'] = rle.encode_many(gt)
            results.append(one)
        host['busy'] = False
        return results
