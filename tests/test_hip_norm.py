"""GroupNorm / average-pool kernels and the from-scratch backbone variant (fgn_r50_c4_scratch.py)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('n,h,w,c,groups,res,relu', [
    (1, 61, 83, 32, 32, False, True),       # stem//2: one channel per group
    (2, 33, 47, 64, 32, False, True),
    (3, 17, 23, 256, 32, True, True),       # bottleneck tail: GN + identity + ReLU
    (9, 8, 8, 1024, 32, True, False),
    (1, 400, 667, 32, 32, False, True),     # cfg3 stem size: 128 chunks per image
    (2, 5, 3, 96, 8, False, False),         # C/4 does not divide the block
])
def test_group_norm_matches_torch(n, h, w, c, groups, res, relu):
    from fgn_amd import ops
    g = torch.Generator().manual_seed(h * w + c)
    x = torch.randn(n, h, w, c, generator=g) * 1.7 + 0.4
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.2
    r = torch.randn(n, h, w, c, generator=g) if res else None
    ref = F.group_norm(x.permute(0, 3, 1, 2), groups, gamma, beta, 1e-5)
    if res:
        ref = ref + r.permute(0, 3, 1, 2)
    if relu:
        ref = F.relu(ref)
    got = ops.group_norm(x.cuda(), gamma.cuda(), beta.cuda(), groups, 1e-5, relu=relu,
                         residual=None if r is None else r.cuda())
    d = (got.cpu().permute(0, 3, 1, 2) - ref).abs().max().item()
    assert d <= 1e-4 * max(ref.abs().max().item(), 1.0), d     # tolerance: fp32 statistics, 1e-4 relative
    # in place gives the same bytes
    xin = x.cuda()
    again = ops.group_norm(xin, gamma.cuda(), beta.cuda(), groups, 1e-5, relu=relu,
                           residual=None if r is None else r.cuda(), inplace=True)
    assert again.data_ptr() == xin.data_ptr() and torch.equal(again, got)


@pytest.mark.parametrize('n,h,w,c', [(1, 24, 33, 64), (2, 17, 17, 256), (1, 1, 5, 32), (3, 100, 167, 512)])
def test_avgpool2x2_ceil_matches_torch(n, h, w, c):
    from fgn_amd import ops
    x = torch.randn(n, h, w, c, generator=torch.Generator().manual_seed(h))
    ref = F.avg_pool2d(x.permute(0, 3, 1, 2), 2, 2, ceil_mode=True, count_include_pad=False)
    got = ops.avgpool2x2(x.cuda()).cpu().permute(0, 3, 1, 2)
    assert got.shape == ref.shape
    assert (got - ref).abs().max().item() <= 1e-6


def test_group_norm_rejects_bad_shapes():
    from fgn_amd import ops
    from fgn_amd.lib import FgnHipError
    x = torch.zeros(1, 4, 4, 30, device='cuda')
    with pytest.raises(FgnHipError):
        ops.group_norm(x, torch.ones(30, device='cuda'), torch.zeros(30, device='cuda'), 3)
    with pytest.raises(FgnHipError):
        ops.group_norm(torch.zeros(1, 4, 4, 32, device='cuda'), torch.ones(32, device='cuda'),
                       torch.zeros(32, device='cuda'), 5)


def test_e2e_scratch_backbone_variant():
    """Deep stem + avg_down + GN(32) backbone through the whole path vs the oracle; odd sizes so the
    ceil-mode average pool sees partial windows."""
    from fgn_amd.config import tiny_config
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import make_batch
    from fgn_amd.fsiseg_eval import evaluate_results
    from fgn_amd.weights import init_state_dict
    from oracle import fgn_ref_cpu as O
    cfg = tiny_config(3, 2, width_div=2, scratch=True)
    sd = init_state_dict(cfg, 1)      # a seed whose random box head fires (seed 0 scores everything background)
    batch = make_batch(0, 2, 3, 2, 150, 214, 64)
    tr_ref = {}
    ref = O.simple_test(sd, cfg, **batch, trace=tr_ref)
    model = FGN(3, 2, backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'],
                test_cfg=cfg['test_cfg'], state_dict=sd)
    model.debug_trace = {}
    got = model.simple_test(**batch, rescale=True)
    for name in ('qry_fmap', 'spp_fmaps'):
        r = tr_ref[name]
        d = (model.debug_trace[name].permute(0, 3, 1, 2).cpu() - r).abs().max().item()
        assert d <= 1e-4 * r.abs().max().item(), (name, d)
    assert sum(len(r['dt_scores']) for r in ref) > 0
    for r, g_ in zip(ref, got):
        assert abs(len(r['dt_scores']) - len(g_['dt_scores'])) <= max(2, len(r['dt_scores']) // 20)
    ap_ref, ap_got = evaluate_results(ref, 3), evaluate_results(got, 3)
    for k in ap_ref:
        assert abs(ap_ref[k] - ap_got[k]) <= 0.1, (k, ap_ref[k], ap_got[k])
    # the mmcv config dict of the reference maps onto the same model
    from fgn_amd.detector import normalise_config
    n = normalise_config(3, 3, backbone=dict(type='ResNet', depth=50, num_stages=3, strides=(1, 2, 2),
                                              out_indices=(2,), deep_stem=True, avg_down=True,
                                              norm_cfg=dict(type='GN', requires_grad=True, num_groups=32)))
    assert n['backbone']['deep_stem'] and n['backbone']['avg_down'] and n['backbone']['norm'] == 'GN'
