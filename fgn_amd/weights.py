"""Seeded synthetic weights in the reference's checkpoint layout.

The reference ships no checkpoints (README.md:27) and its DenseCL init is an
absolute external path (fgn_r50_c4_densecl.py:4-11), so benchmarks and parity
tests run on seeded random weights.  Parameter names follow the mmcv
``state_dict`` layout a real FGN checkpoint has, so a real
checkpoint loads through the same code path:

  backbone.{conv1,bn1,layer1..3.N.{conv1..3,bn1..3,downsample.{0,1}}}
  rpn_head.{rpn_conv,rpn_cls,rpn_reg}
  roi_head.shared_head.{0,1,2}.{conv1..3,bn1..3}          (fgn_roi_head.py:202-233)
  roi_head.cls_reg_shared_conv / cls_reg_shared_conv_norm   (fgn_roi_head.py:240-251)
  roi_head.bbox_head.{fc_cls,fc_reg}
  roi_head.mask_head.{convs.N.conv,upsample,conv_logits}

Initialisers follow the reference where it defines them (Kaiming-normal convs in
shared_head / relation conv, Xavier-normal RPN and Linear layers;
fgn_roi_head.py:224-231,247-251, fgn_r50_c4_densecl.py:61-63,103-106).  BN
statistics are randomised (not 0/1) so that the folded-BN epilogue is actually
exercised, and the last BN of every bottleneck is damped so activations stay O(1)
through 16 residual blocks.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import torch


def _kaiming(g, cout, cin, kh, kw):
    std = math.sqrt(2.0 / (cin * kh * kw))
    return torch.randn(cout, cin, kh, kw, generator=g) * std


def _xavier(g, *shape):
    if len(shape) == 4:
        fan_out = shape[0] * shape[2] * shape[3]
        fan_in = shape[1] * shape[2] * shape[3]
    else:
        fan_out, fan_in = shape
    std = math.sqrt(2.0 / (fan_in + fan_out))
    return torch.randn(*shape, generator=g) * std


def _bn(sd, g, name, c, gamma_scale=1.0):
    sd[name + '.weight'] = (0.75 + 0.5 * torch.rand(c, generator=g)) * gamma_scale
    sd[name + '.bias'] = 0.1 * torch.randn(c, generator=g)
    sd[name + '.running_mean'] = 0.1 * torch.randn(c, generator=g)
    sd[name + '.running_var'] = 0.75 + 0.5 * torch.rand(c, generator=g)
    sd[name + '.num_batches_tracked'] = torch.zeros((), dtype=torch.long)      # torch BatchNorm buffer (no RNG draw)


def _gn(sd, g, name, c, gamma_scale=1.0):
    sd[name + '.weight'] = (0.75 + 0.5 * torch.rand(c, generator=g)) * gamma_scale
    sd[name + '.bias'] = 0.1 * torch.randn(c, generator=g)


def _bottleneck_gn(sd, g, prefix, cin, planes, cout, downsample, pooled):
    """mmdet Bottleneck with norm_cfg GN: norm layers are named gn1..gn3; under avg_down the shortcut
    is [AvgPool2d (only when strided), conv1x1, norm] so the conv sits at index 1 (strided) or 0."""
    sd[prefix + '.conv1.weight'] = _kaiming(g, planes, cin, 1, 1)
    _gn(sd, g, prefix + '.gn1', planes)
    sd[prefix + '.conv2.weight'] = _kaiming(g, planes, planes, 3, 3)
    _gn(sd, g, prefix + '.gn2', planes)
    sd[prefix + '.conv3.weight'] = _kaiming(g, cout, planes, 1, 1)
    _gn(sd, g, prefix + '.gn3', cout, gamma_scale=0.35)
    if downsample:
        i = 1 if pooled else 0
        sd[f'{prefix}.downsample.{i}.weight'] = _kaiming(g, cout, cin, 1, 1)
        _gn(sd, g, f'{prefix}.downsample.{i + 1}', cout, gamma_scale=0.7)


def _bottleneck(sd, g, prefix, cin, planes, cout, downsample):
    sd[prefix + '.conv1.weight'] = _kaiming(g, planes, cin, 1, 1)
    _bn(sd, g, prefix + '.bn1', planes)
    sd[prefix + '.conv2.weight'] = _kaiming(g, planes, planes, 3, 3)
    _bn(sd, g, prefix + '.bn2', planes)
    sd[prefix + '.conv3.weight'] = _kaiming(g, cout, planes, 1, 1)
    _bn(sd, g, prefix + '.bn3', cout, gamma_scale=0.35)
    if downsample:
        sd[prefix + '.downsample.0.weight'] = _kaiming(g, cout, cin, 1, 1)
        _bn(sd, g, prefix + '.downsample.1', cout, gamma_scale=0.7)


def _basic_block(sd, g, prefix, cin, planes, downsample):
    """mmdet BasicBlock (ResNet-18/34): two 3x3 convs, expansion 1."""
    sd[prefix + '.conv1.weight'] = _kaiming(g, planes, cin, 3, 3)
    _bn(sd, g, prefix + '.bn1', planes)
    sd[prefix + '.conv2.weight'] = _kaiming(g, planes, planes, 3, 3)
    _bn(sd, g, prefix + '.bn2', planes, gamma_scale=0.35)
    if downsample:
        sd[prefix + '.downsample.0.weight'] = _kaiming(g, planes, cin, 1, 1)
        _bn(sd, g, prefix + '.downsample.1', planes, gamma_scale=0.7)


def init_state_dict(cfg: dict, seed: int = 0) -> 'OrderedDict[str, torch.Tensor]':
    """Build a seeded fp32 state_dict for the model described by ``cfg``."""
    g = torch.Generator().manual_seed(seed)
    sd: 'OrderedDict[str, torch.Tensor]' = OrderedDict()
    bb = cfg['backbone']
    stem = bb['stem_channels']
    scratch = bb.get('norm', 'BN') == 'GN'
    if bb.get('deep_stem'):       # mmdet ResNet deep stem: Sequential(conv, norm, relu) x 3 -> stem.{0,1,3,4,6,7}
        for i, (ci, co) in enumerate(((3, stem // 2), (stem // 2, stem // 2), (stem // 2, stem))):
            sd[f'backbone.stem.{3 * i}.weight'] = _kaiming(g, co, ci, 3, 3)
            (_gn if scratch else _bn)(sd, g, f'backbone.stem.{3 * i + 1}', co)
    else:
        sd['backbone.conv1.weight'] = _kaiming(g, stem, 3, 7, 7)
        (_gn if scratch else _bn)(sd, g, 'backbone.gn1' if scratch else 'backbone.bn1', stem)
    cin = stem
    basic = bb.get('block', 'bottleneck') == 'basic'
    for li, (nblk, planes, stride) in enumerate(zip(bb['stage_blocks'], bb['stage_planes'], bb['strides'])):
        cout = planes if basic else planes * 4
        for b in range(nblk):
            if basic:
                _basic_block(sd, g, f'backbone.layer{li + 1}.{b}', cin, planes,
                             downsample=(b == 0 and (stride != 1 or cin != planes)))
            elif scratch:
                _bottleneck_gn(sd, g, f'backbone.layer{li + 1}.{b}', cin, planes, cout, downsample=(b == 0),
                               pooled=bool(bb.get('avg_down')) and stride != 1)
            else:
                _bottleneck(sd, g, f'backbone.layer{li + 1}.{b}', cin, planes, cout,
                            downsample=(b == 0))
            cin = cout

    rp = cfg['rpn_head']
    c, f = rp['in_channels'], rp['feat_channels']
    na = len(rp['anchor_scales']) * len(rp['anchor_ratios'])
    sd['rpn_head.rpn_conv.weight'] = _xavier(g, f, c, 3, 3) * 0.25
    sd['rpn_head.rpn_conv.bias'] = 0.01 * torch.randn(f, generator=g)
    sd['rpn_head.rpn_cls.weight'] = _xavier(g, na, f, 1, 1) * 0.5
    sd['rpn_head.rpn_cls.bias'] = 0.01 * torch.randn(na, generator=g)
    sd['rpn_head.rpn_reg.weight'] = _xavier(g, na * 4, f, 1, 1) * 0.1
    sd['rpn_head.rpn_reg.bias'] = 0.01 * torch.randn(na * 4, generator=g)

    rh = cfg['roi_head']
    sh = rh['shared_head']
    for b in range(sh['num_blocks']):
        _bottleneck(sd, g, f'roi_head.shared_head.{b}', sh['inplanes'],
                    sh['planes'], sh['inplanes'], downsample=False)
    rel = rh['relation']
    sd['roi_head.cls_reg_shared_conv.weight'] = _kaiming(
        g, rel['out_channels'], rel['in_channels'], 1, 1)
    bound = 1.0 / math.sqrt(rel['in_channels'])
    sd['roi_head.cls_reg_shared_conv.bias'] = (
        torch.rand(rel['out_channels'], generator=g) * 2 - 1) * bound
    sd['roi_head.cls_reg_shared_conv_norm.weight'] = 0.75 + 0.5 * torch.rand(
        rel['out_channels'], generator=g)
    sd['roi_head.cls_reg_shared_conv_norm.bias'] = 0.1 * torch.randn(
        rel['out_channels'], generator=g)

    bh = rh['bbox_head']
    ncls = bh['num_classes'] + 1
    sd['roi_head.bbox_head.fc_cls.weight'] = _xavier(g, ncls, bh['in_channels']) * 8.0
    sd['roi_head.bbox_head.fc_cls.bias'] = 0.01 * torch.randn(ncls, generator=g)
    sd['roi_head.bbox_head.fc_reg.weight'] = _xavier(g, 4, bh['in_channels'])
    sd['roi_head.bbox_head.fc_reg.bias'] = 0.01 * torch.randn(4, generator=g)

    mh = rh['mask_head']
    cin = mh['in_channels']
    co = mh['conv_out_channels']
    for i in range(mh['num_convs']):
        # conv 0 sees roi_feat * support vector (a product of two O(2) maps)
        sd[f'roi_head.mask_head.convs.{i}.conv.weight'] = _kaiming(g, co, cin, 3, 3) * \
            (0.15 if i == 0 else 1.0)
        sd[f'roi_head.mask_head.convs.{i}.conv.bias'] = 0.01 * torch.randn(co, generator=g)
        cin = co
    # ConvTranspose2d weight layout is [in, out, kh, kw]
    sd['roi_head.mask_head.upsample.weight'] = torch.randn(
        co, co, 2, 2, generator=g) * math.sqrt(2.0 / co)
    sd['roi_head.mask_head.upsample.bias'] = 0.01 * torch.randn(co, generator=g)
    # zero-sum logits weights: post-ReLU features are all positive, a zero-sum
    # filter keeps the synthetic masks mixed instead of saturated
    wl = _xavier(g, mh['num_classes'], co, 1, 1)
    sd['roi_head.mask_head.conv_logits.weight'] = (wl - wl.mean(dim=1, keepdim=True)) * 2.0
    sd['roi_head.mask_head.conv_logits.bias'] = 0.01 * torch.randn(
        mh['num_classes'], generator=g)
    return sd


def load_checkpoint(path: str) -> 'OrderedDict[str, torch.Tensor]':
    """Load an mmcv-style checkpoint ({'state_dict', 'meta', ...}) or a bare
    state_dict (main.py:426-430 resumes from the former)."""
    ckpt = torch.load(path, map_location='cpu')
    sd = ckpt.get('state_dict', ckpt)
    return OrderedDict((k, v if k.endswith('num_batches_tracked') else v.float()) for k, v in sd.items())


def backbone_state_from_pretrained(ckpt, prefix: str = 'backbone.') -> 'OrderedDict[str, torch.Tensor]':
    """Keys of a backbone-only checkpoint -> this model's ``backbone.*`` names.

    The reference initialises its headline config from a DenseCL ResNet-50 file through mmcv's ``Pretrained``
    init_cfg on the BACKBONE module (fgn_r50_c4_densecl.py:4-11, 39-41; ``model.init_weights()``, main.py:431-434):
    ``load_checkpoint(backbone, path, strict=False)``.  Such a file is a torchvision-style ResNet state dict - keys
    ``conv1.weight``, ``bn1.*``, ``layer1.0.conv1.weight`` ... without any module prefix, possibly wrapped in
    ``{'state_dict': ...}``, possibly carrying a ``module.`` / ``backbone.`` / ``encoder_q.`` prefix from the
    self-supervised trainer, and with ``layer4.*`` / ``fc.*`` entries this C4 model does not have (main.py:403-405
    deletes layer4).  Returns the renamed tensors; what does not belong to the backbone is dropped by the caller
    (non-strict, like mmcv)."""
    if isinstance(ckpt, str):
        ckpt = torch.load(ckpt, map_location='cpu')
    sd = ckpt.get('state_dict', ckpt)
    out = OrderedDict()
    for k, v in sd.items():
        for pre in ('module.', 'backbone.', 'encoder_q.', 'encoder.'):      # mmcv strips 'module.'; the rest is lenient
            while k.startswith(pre):
                k = k[len(pre):]
        out[prefix + k] = v if k.endswith('num_batches_tracked') else v.float()
    return out
