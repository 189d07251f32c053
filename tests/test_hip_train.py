"""``forward_train``: the HIP kernels of the training path against the training oracle.
Selection (assignment, sampling, proposal ranking / NMS) is bit-exact; losses carry an fp32 tolerance stated per test."""
import copy
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _boxes(g, n, span=300.0):
    ctr = torch.rand(n, 2, generator=g) * span
    wh = torch.rand(n, 2, generator=g) * 80 + 4
    return torch.cat([ctr - wh / 2, ctr + wh / 2], 1)


# ---------------------------------------------------------------- assigner: bit-exact
@pytest.mark.parametrize('k,n,thr', [(5, 4000, (0.5, 0.3, 0.3)), (7, 2000, (0.5, 0.5, 0.5)), (1, 300, (0.7, 0.3, 0.3)),
                                     (0, 100, (0.5, 0.3, 0.3)), (3, 500, (0.5, 0.5, 0.0)), (40, 3000, (0.5, 0.5, 0.5))])
def test_box_assign_matches_oracle_bitwise(k, n, thr):
    from fgn_amd import ops
    from oracle import fgn_train_cpu as T
    g = torch.Generator().manual_seed(k * 31 + n)
    boxes = _boxes(g, n)
    gts = boxes[torch.randperm(n, generator=g)[:k]] + torch.randn(k, 4, generator=g) * 4 if k else boxes[:0]
    boxes[10:30] = boxes[40:60]                       # duplicated boxes: tied IoUs
    if k >= 2:
        gts[1] = gts[0]                                 # duplicated GT: arg-max takes the first, low-quality the last
    inside = (torch.rand(n, generator=g) > 0.2)
    ref_all, mo_all = T.max_iou_assign(T.bbox_overlaps(gts, boxes), *thr, True)
    got, mo = ops.box_assign(boxes.cuda(), gts.cuda(), *thr, True, with_overlaps=True)
    assert np.array_equal(got.cpu().numpy(), ref_all.numpy().astype(np.int32))
    if k:
        assert np.array_equal(mo.cpu().numpy(), mo_all.numpy())        # IoU itself bit for bit
    # with inside flags: the reference assigns on the compacted list (my_anchor_head.py:239-243)
    ref_in, _ = T.max_iou_assign(T.bbox_overlaps(gts, boxes[inside]), *thr, True)
    got = ops.box_assign(boxes.cuda(), gts.cuda(), *thr, True, inside=inside.to(torch.uint8).cuda()).cpu()
    assert np.array_equal(got[inside].numpy(), ref_in.numpy().astype(np.int32))
    assert bool((got[~inside] == -2).all())
    # [n,5] proposals (score column ignored)
    b5 = torch.cat([boxes, torch.rand(n, 1, generator=g)], 1)
    got = ops.box_assign(b5.cuda(), gts.cuda(), *thr, True).cpu()
    assert np.array_equal(got.numpy(), ref_all.numpy().astype(np.int32))


def test_box_assign_on_the_anchor_grid():
    """63 000 anchors of the cfg3 grid against GT boxes with integer corners (ties between anchors are common)."""
    from fgn_amd import ops
    from oracle import fgn_ref_cpu as O, fgn_train_cpu as T
    base = O.base_anchors((2, 4, 8, 16, 32), (0.5, 1.0, 2.0), 16)
    anchors = torch.from_numpy(O.grid_anchors(base, 50, 84, 16))
    gts = torch.tensor([[100., 120., 420., 380.], [16., 16., 80., 48.], [600., 300., 1300., 790.],
                        [601., 301., 1301., 791.], [0., 0., 32., 32.]])
    valid = T.valid_flags(50, 84, 800, 1333, 16, 15)
    inside = T.anchor_inside_flags(anchors, valid, 800., 1333., 0)
    ref, _ = T.max_iou_assign(T.bbox_overlaps(gts, anchors[inside]), 0.5, 0.3, 0.3, True)
    got = ops.box_assign(anchors.cuda(), gts.cuda(), 0.5, 0.3, 0.3, True, inside=inside.to(torch.uint8).cuda()).cpu()
    assert np.array_equal(got[inside].numpy(), ref.numpy().astype(np.int32))
    assert int((ref > 0).sum()) > 0


def test_bbox2delta_matches_oracle():
    from fgn_amd import ops
    from oracle import fgn_train_cpu as T
    g = torch.Generator().manual_seed(3)
    p, q = _boxes(g, 500), _boxes(g, 500)
    for means, stds in (((0., 0., 0., 0.), (1., 1., 1., 1.)), ((0., 0., 0., 0.), (.1, .1, .2, .2))):
        ref = T.bbox2delta(p, q, means, stds)
        got = ops.bbox2delta(p.cuda(), q.cuda(), means, stds).cpu()
        # torch.log (vectorised, <= 1 ulp) vs the correctly rounded log here, then / std
        assert float((got - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


# ---------------------------------------------------------------- loss reductions
def test_loss_sums_match_torch():
    from fgn_amd import ops
    from oracle import fgn_train_cpu as T
    g = torch.Generator().manual_seed(4)
    n = 5000
    x, w = torch.randn(n, generator=g) * 4, torch.rand(n, generator=g)
    y = (torch.rand(n, generator=g) > 0.7).float()
    ref = (F.binary_cross_entropy_with_logits(x, y, reduction='none') * w).sum() / 77.0
    got = ops.bce_logits_sum(x.cuda(), y.cuda(), w.cuda(), 77.0).cpu()
    assert float(got) == pytest.approx(float(ref), rel=2e-6)
    soft = torch.rand(n, generator=g)                                   # mask-target form: binarised at 0.5, mean
    ref = F.binary_cross_entropy_with_logits(x, (soft >= 0.5).float(), reduction='mean')
    got = ops.bce_logits_sum(x.cuda(), soft.cuda(), None, float(n), y_threshold=0.5).cpu()
    assert float(got) == pytest.approx(float(ref), rel=2e-6)
    p, t, w4 = torch.randn(n, 4, generator=g) * 2, torch.randn(n, 4, generator=g), torch.rand(n, 4, generator=g)
    ref = T.smooth_l1_weighted(p, t, w4, 128.0)
    got = ops.smooth_l1_sum(p.cuda(), t.cuda(), w4.cuda(), 128.0).cpu()
    assert float(got) == pytest.approx(float(ref), rel=2e-6)
    logits = torch.randn(700, 4, generator=g) * 3
    labels = torch.randint(0, 4, (700,), generator=g)
    lw = torch.rand(700, generator=g)
    ref = T.softmax_ce_weighted(logits, labels, lw, 300.0)
    got = ops.softmax_ce_sum(logits.cuda(), labels.cuda(), lw.cuda(), 300.0).cpu()
    assert float(got) == pytest.approx(float(ref), rel=2e-6)
    assert float(ops.bce_logits_sum(x[:0].cuda(), y[:0].cuda(), None, 1.0).cpu()) == 0.0


# ---------------------------------------------------------------- BatchNorm, training mode
@pytest.mark.parametrize('P,C', [(128 * 49, 512), (9 * 49, 1024), (3, 64), (6272, 128)])
def test_bn_train_matches_torch(P, C):
    from fgn_amd import ops
    g = torch.Generator().manual_seed(P + C)
    x = torch.randn(P, C, generator=g) * 2 + torch.randn(C, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    rm, rv = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    res = torch.randn(P, C, generator=g)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    ref = F.batch_norm(x.t()[None].contiguous(), rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)[0].t()
    ref = F.relu(ref + res)
    rm_d, rv_d = rm.cuda(), rv.cuda()
    y, mean, var = ops.bn_train(x.cuda(), gamma.cuda(), beta.cuda(), 1e-5, 0.1, rm_d, rv_d, residual=res.cuda(),
                                relu=True, inplace=False)
    assert float((y.cpu() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert float((mean.cpu() - x.mean(0)).abs().max()) <= 1e-6
    assert float((var.cpu() - x.var(0, unbiased=False)).abs().max()) <= 1e-5 * float(x.var(0).max())
    assert float((rm_d.cpu() - rm_ref).abs().max()) <= 1e-6
    assert float((rv_d.cpu() - rv_ref).abs().max()) <= 1e-5 * float(rv_ref.max())


# ---------------------------------------------------------------- proposals at training sizes: bit-exact
def _train_proposal_case(seed, fh, fw, nms_pre, max_out, ties, img):
    from fgn_amd import ops
    from fgn_amd.config import fgn_r50_c4_config
    from oracle import fgn_ref_cpu as O, fgn_train_cpu as T
    cfg = fgn_r50_c4_config(3, 3)
    pc = dict(nms_pre=nms_pre, max_per_img=max_out, nms_iou_threshold=0.7, min_bbox_size=0)
    g = torch.Generator().manual_seed(seed)
    cls = torch.randn(15, fh, fw, generator=g) * 3
    if ties:
        cls = (cls * 2).round() / 2
        cls[:, : fh // 2] += 20
    reg = torch.randn(60, fh, fw, generator=g) * 0.5
    ref = T.rpn_get_bboxes_cfg(cls.numpy(), reg.numpy(), np.array([img[0], img[1], 3]), cfg, pc)
    logits = cls.permute(1, 2, 0).reshape(1, -1).contiguous()
    scores = torch.from_numpy(O.sigmoid32(logits.numpy()))
    deltas = reg.permute(1, 2, 0).reshape(1, -1, 4).contiguous()
    rp = cfg['rpn_head']
    anchors = torch.from_numpy(ops.base_anchors(rp['anchor_scales'], rp['anchor_ratios'], rp['anchor_stride']))
    # two images in one call: the second is the first with its scores reversed in sign
    sc2 = torch.cat([scores, torch.from_numpy(O.sigmoid32(-logits.numpy()))]).cuda()
    props, n, rois = ops.rpn_proposals(sc2, torch.cat([deltas, deltas]).cuda(), anchors.cuda(), fh, fw, 16, img[0], img[1],
                                       rp['target_means'], rp['target_stds'], nms_pre, 0, 0.7, max_out, with_rois=True)
    n0 = int(n[0].item())
    assert n0 == len(ref), (n0, len(ref))
    assert np.array_equal(props[0, :n0].cpu().numpy(), ref)              # boxes AND scores bitwise
    assert float(props[0, n0:].abs().sum()) == 0.0
    ref1 = T.rpn_get_bboxes_cfg(-cls.numpy(), reg.numpy(), np.array([img[0], img[1], 3]), cfg, pc)
    n1 = int(n[1].item())
    assert n1 == len(ref1) and np.array_equal(props[1, :n1].cpu().numpy(), ref1)
    assert np.array_equal(rois[max_out:max_out + n1, 1:].cpu().numpy(), ref1[:, :4])
    assert bool((rois[max_out:, 0] == 1).all())


@pytest.mark.parametrize('ties', [False, True])
def test_train_proposals_bit_exact_cfg3_size(ties):
    _train_proposal_case(11, 50, 84, 12000, 2000, ties, (800, 1333))     # 63 000 anchors -> 12 000 -> 2000


def test_train_proposals_bit_exact_small_maps():
    _train_proposal_case(12, 10, 14, 12000, 2000, False, (160, 224))     # 2100 anchors: fewer than nms_pre
    _train_proposal_case(13, 24, 30, 9000, 1500, True, (380, 470))       # 10 800 anchors -> 9000


# ---------------------------------------------------------------- forward_train end to end
def _models(cfg, seed=0):
    from fgn_amd.detector import FGN
    from fgn_amd.weights import init_state_dict
    sd = init_state_dict(cfg, seed)
    # non-trivial BatchNorm parameters for the shared head so that the train-mode statistics matter
    g = torch.Generator().manual_seed(99)
    for k in sd:
        if k.startswith('roi_head.shared_head') and k.endswith('.weight') and sd[k].dim() == 1:
            sd[k] = 0.75 + 0.5 * torch.rand(sd[k].shape, generator=g)
        if k.startswith('roi_head.shared_head') and k.endswith('.bias') and sd[k].dim() == 1:
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=g)
    m = FGN(cfg['n_ways'], cfg['k_shots'], backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'],
            test_cfg=cfg['test_cfg'], state_dict=sd)
    return m, sd


def _f(v):
    v = v[0] if isinstance(v, list) else v
    return float(v.detach().reshape(-1)[0]) if isinstance(v, torch.Tensor) else float(v)


def _compare_losses(got, ref, rel):
    for k in ('loss_rpn_cls', 'loss_rpn_bbox'):
        assert isinstance(got[k], list) and len(got[k]) == 1
    for k in ('loss_rpn_cls', 'loss_rpn_bbox', 'loss_cls', 'loss_bbox', 'loss_mask'):
        assert _f(got[k]) == pytest.approx(_f(ref[k]), rel=rel, abs=1e-7), k
    for k in ('ACC-Unbalanced', 'ACC-Balanced'):
        assert _f(got[k]) == pytest.approx(_f(ref[k]), abs=1e-6), k
    assert set(got) == set(ref)


@pytest.mark.parametrize('n_ways,k_shots,batch', [(3, 3, 2), (1, 1, 1), (5, 2, 1)])
def test_forward_train_matches_oracle_small(n_ways, k_shots, batch):
    """Half-width model, 160x224 queries: every stage compared - sampled sets and labels bit-exact, losses to 1e-4."""
    from fgn_amd.config import tiny_config
    from fgn_amd.episodes import make_batch
    from fgn_amd import train as TR
    from oracle import fgn_train_cpu as T
    cfg = tiny_config(n_ways, k_shots, width_div=2)
    m, sd = _models(cfg)
    b = make_batch(0, batch, n_ways, k_shots, 160, 224, 64)
    sd_ref = copy.deepcopy(sd)
    tr_ref = {}
    torch.manual_seed(5)
    ref = T.forward_train(sd_ref, cfg, trace=tr_ref, **b)
    m.debug_trace = {}
    torch.manual_seed(5)
    got = m.forward(return_loss=True, **b)
    tr = m.debug_trace
    # the same anchors sampled for every guided pass
    for (pos, neg), t in zip(tr['rpn_sets'], tr_ref['rpn_targets']):
        full = torch.nonzero(t['inside']).view(-1)
        assert np.array_equal(pos.cpu().numpy(), full[t['pos_inds']].numpy())
        assert np.array_equal(neg.cpu().numpy(), full[t['neg_inds']].numpy())
    assert tr['rpn_num_total_samples'] == tr_ref['rpn_num_total_samples']
    # the same proposals kept (count) and the same RoIs sampled
    for p, q in zip(tr['proposals'], tr_ref['proposals']):
        assert p.shape[0] == len(q)
        assert float((p.cpu() - torch.from_numpy(q)).abs().max()) < 1e-2
    for s, r in zip(tr['samples'], tr_ref['samples']):
        assert np.array_equal(s['pos_inds'].cpu().numpy(), r['pos_inds'].numpy())
        assert np.array_equal(s['neg_inds'].cpu().numpy(), r['neg_inds'].numpy())
        assert np.array_equal(s['pos_gt_labels'].cpu().numpy(), r['pos_gt_labels'].numpy())
    assert np.array_equal(tr['labels'].cpu().numpy(), tr_ref['bbox_targets'][0].numpy())
    # train-mode shared head output and box head outputs
    bf = tr['bbox_feats'].permute(0, 3, 1, 2).cpu()
    assert float((bf - tr_ref['bbox_feats']).abs().max()) <= 2e-4 * float(tr_ref['bbox_feats'].abs().max())
    assert float((tr['cls_score'].cpu() - tr_ref['cls_score']).abs().max()) <= 1e-4 * \
        max(1.0, float(tr_ref['cls_score'].abs().max()))
    # mask targets: identical except pixels whose pooled value sits on the 0.5 threshold
    soft = tr['mask_targets_soft'].cpu()
    flips = ((soft >= 0.5).float() != tr_ref['mask_targets'])
    assert bool(((soft - 0.5).abs()[flips] < 1e-5).all())
    assert int(flips.sum()) <= 2
    _compare_losses(got, ref, 1e-4)
    # running BatchNorm statistics of the shared head were updated like the module's buffers
    for k, v in TR.bn_buffers(m).items():
        assert float((v - sd_ref[k]).abs().max()) <= 1e-4 * max(1.0, float(sd_ref[k].abs().max())), k
        assert float((v - sd[k]).abs().max()) > 0


def test_forward_train_full_width_given_proposals():
    """R50 widths at 256x320: the oracle's proposals are handed to both sides so that the RoI stage is compared at
    full width without depending on near-tie NMS decisions of random weights."""
    from fgn_amd.config import fgn_r50_c4_config
    from fgn_amd.episodes import make_batch
    from oracle import fgn_train_cpu as T
    cfg = fgn_r50_c4_config(3, 2)
    m, sd = _models(cfg)
    b = make_batch(3, 1, 3, 2, 256, 320, 128)
    tr_ref = {}
    torch.manual_seed(8)
    ref = T.forward_train(copy.deepcopy(sd), cfg, trace=tr_ref, **b)
    torch.manual_seed(8)
    got = m.forward_train(proposals=tr_ref['proposals'], **b)
    _compare_losses(got, ref, 2e-4)
    # gradients at full width (1024-channel relation head with 32-channel GroupNorm groups, 512 / 1024-channel
    # BatchNorm, 1024 -> 256 mask conv): same bounds as the half-width end-to-end test
    from fgn_amd.train import Trainer
    sd_g = {k: v.clone() for k, v in sd.items()}
    names = [k for k, v in sd_g.items() if k.startswith(T.TRAINABLE_PREFIXES) and v.is_floating_point()
             and 'running_' not in k]
    for k in names:
        sd_g[k].requires_grad_(True)
    torch.manual_seed(8)
    T.total_loss(T.forward_train(sd_g, cfg, grad=True, proposals=tr_ref['proposals'], **b)).backward()
    m2, _ = _models(cfg)
    trn = Trainer(m2)
    torch.manual_seed(8)
    trn.forward_backward(dict(b, proposals=tr_ref['proposals']))
    rep = _grad_report(trn.grads, {k: sd_g[k].grad for k in names})
    print('full width, largest gradient errors (max/max, L2/L2):', sorted(rep.items(), key=lambda kv: -kv[1][1])[:4])
    bad = {k: v for k, v in rep.items() if v[1] > 1e-2 or v[0] > 6e-2}
    assert not bad, bad


def test_forward_train_cfg3_size_runs_and_is_reproducible():
    """800x1333 queries, 63 000 anchors, 12 000 -> 2000 proposals: finite losses, bit-identical under the same seed."""
    from fgn_amd.config import fgn_r50_c4_config
    from fgn_amd.episodes import CONFIGS, make_batch
    cfg = fgn_r50_c4_config(3, 3)
    m, _ = _models(cfg)
    b = make_batch(0, 1, **CONFIGS['cfg3'])
    torch.manual_seed(1)
    a = m.forward_train(**b)
    m._PT = None                                    # fresh running statistics: same inputs, same outputs
    torch.manual_seed(1)
    c = m.forward_train(**b)
    for k in a:
        va = float(a[k][0]) if isinstance(a[k], list) else float(a[k])
        vc = float(c[k][0]) if isinstance(c[k], list) else float(c[k])
        assert np.isfinite(va) and va == vc, k
    assert float(a['loss_rpn_cls'][0]) > 0 and float(a['loss_cls']) > 0 and float(a['loss_mask']) > 0
    # ... and equal to the oracle's at full size: 63 000 anchors assigned and sampled, 12 000 -> 2000 proposals,
    # 128 RoIs through the full-width heads (same sampled sets; losses to 2e-4)
    from oracle import fgn_train_cpu as T
    _, sd = _models(cfg)
    tr_ref = {}
    torch.manual_seed(1)
    ref = T.forward_train(copy.deepcopy(sd), cfg, trace=tr_ref, **b)
    m._PT = None
    m.debug_trace = {}
    torch.manual_seed(1)
    got = m.forward_train(**b)
    # the AG-RPN half does not depend on the proposals: same sampled anchors, same two losses
    for (pos, neg), t in zip(m.debug_trace['rpn_sets'], tr_ref['rpn_targets']):
        full = torch.nonzero(t['inside']).view(-1)
        assert np.array_equal(pos.numpy(), full[t['pos_inds']].numpy()) and \
            np.array_equal(neg.numpy(), full[t['neg_inds']].numpy())
    for k in ('loss_rpn_cls', 'loss_rpn_bbox'):
        assert _f(got[k]) == pytest.approx(_f(ref[k]), rel=2e-4), k
    # the 2000 proposals: the same boxes up to near-tie rank swaps of the 12 000 -> NMS list (scores of the two paths
    # differ by ~1e-6 and random weights put many anchors within that of each other): compared as sets
    p, q = m.debug_trace['proposals'][0].cpu(), torch.from_numpy(tr_ref['proposals'][0])
    assert p.shape == q.shape
    d = (p[:, None, :4] - q[None, :, :4]).abs().amax(-1)
    unmatched = int((d.min(1).values > 1e-2).sum())
    print('proposals of the HIP path without a counterpart in the oracle list:', unmatched, 'of', p.shape[0])
    assert unmatched <= 0.02 * p.shape[0]
    # RoI stage at full size on identical proposals: same sampled RoIs, all seven losses
    m._PT = None
    m.debug_trace = {}
    torch.manual_seed(1)
    got = m.forward_train(proposals=tr_ref['proposals'], **b)
    s, r = m.debug_trace['samples'][0], tr_ref['samples'][0]
    assert np.array_equal(s['pos_inds'].numpy(), r['pos_inds'].numpy())
    assert np.array_equal(s['neg_inds'].numpy(), r['neg_inds'].numpy())
    _compare_losses(got, ref, 2e-4)


# ---------------------------------------------------------------- backward: gradients vs torch.autograd on the oracle
def _oracle_grads(sd, cfg, b, seed):
    from oracle import fgn_train_cpu as T
    sd = {k: v.clone() for k, v in sd.items()}
    names = [k for k, v in sd.items() if k.startswith(T.TRAINABLE_PREFIXES) and v.is_floating_point()
             and 'running_' not in k]
    for k in names:
        sd[k].requires_grad_(True)
    torch.manual_seed(seed)
    losses = T.forward_train(sd, cfg, grad=True, **b)
    T.total_loss(losses).backward()
    return losses, {k: sd[k].grad for k in names}


def _grad_report(got, ref):
    """-> {name: (max abs error / max abs reference, L2 error / L2 reference)}"""
    rep = {}
    for k, r in ref.items():
        assert k in got, f'no gradient for {k}'
        g = got[k].cpu()
        assert tuple(g.shape) == tuple(r.shape), (k, g.shape, r.shape)
        rep[k] = (float((g - r).abs().max()) / (float(r.abs().max()) + 1e-12),
                  float((g - r).norm()) / (float(r.norm()) + 1e-12))
    return rep


@pytest.mark.parametrize('n_ways,k_shots,batch,l2_bound,max_bound', [(3, 2, 2, 3e-3, 3e-2), (1, 2, 2, 1e-2, 6e-2)])
def test_backward_matches_autograd_of_the_oracle(n_ways, k_shots, batch, l2_bound, max_bound):
    """Every trainable tensor of the heads (AG-RPN, shared head incl. train-mode BN, relation conv + GN, fc, mask
    head) end to end against torch.autograd on the oracle.  The pieces are exact on identical inputs (the *_alone
    tests, 2e-4 / 2e-5); end to end the two forward passes differ by ~1e-5 relative, and a ReLU whose pre-activation
    changes sign between them switches one element's gradient on or off - behind a train-mode BatchNorm that moves a
    channel's sums by ~1 % (more with the 196-row support batch of the 1-way case).  Hence two bounds per tensor: L2
    error / L2 norm (a missing or mis-scaled term shows here) and worst element / largest element."""
    from fgn_amd.config import tiny_config
    from fgn_amd.episodes import make_batch
    from fgn_amd.train import Trainer
    cfg = tiny_config(n_ways, k_shots, width_div=2)
    m, sd = _models(cfg)
    b = make_batch(0, batch, n_ways, k_shots, 160, 224, 64)
    ref_losses, ref = _oracle_grads(sd, cfg, b, 5)
    tr = Trainer(m)
    torch.manual_seed(5)
    got_losses = tr.forward_backward(b)
    _compare_losses(got_losses, ref_losses, 1e-4)
    rep = _grad_report(tr.grads, ref)
    print('largest gradient errors (max/max, L2/L2):', sorted(rep.items(), key=lambda kv: -kv[1][1])[:6])
    bad = {k: v for k, v in rep.items() if v[1] > l2_bound or v[0] > max_bound}
    assert not bad, bad
    # everything outside the shared head sits behind at most one ReLU flip: tight
    for k, v in rep.items():
        if 'shared_head' not in k:
            assert v[1] <= 3e-3, (k, v)
    assert set(tr.grads) == set(ref)


def test_adagrad_step_matches_torch_and_training_reduces_the_loss():
    from fgn_amd import ops
    from fgn_amd.config import tiny_config
    from fgn_amd.episodes import make_batch
    from fgn_amd.train import Trainer
    g = torch.Generator().manual_seed(1)
    p0, gr = torch.randn(1000, generator=g), torch.randn(1000, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adagrad([p_ref], lr=0.005, weight_decay=1e-5)
    p_dev, st = p0.cuda(), torch.zeros(1000, device='cuda')
    for _ in range(3):
        p_ref.grad = gr.clone()
        opt.step()
        ops.adagrad_step(p_dev, gr.cuda(), st, 0.005, 1e-5)
    assert float((p_dev.cpu() - p_ref.detach()).abs().max()) <= 1e-6
    # a few steps on one episode with the same sampled sets: the summed loss goes down
    cfg = tiny_config(3, 2, width_div=2)
    m, _ = _models(cfg)
    b = make_batch(1, 1, 3, 2, 160, 224, 64)
    tr = Trainer(m, lr=0.01)
    tot = []
    for _ in range(8):
        torch.manual_seed(3)
        L = tr.step(b)
        tot.append(sum(float(v[0]) if isinstance(v, list) else float(v) for k, v in L.items() if 'loss' in k))
    print('summed loss over 8 steps:', [round(t, 4) for t in tot])
    assert tot[-1] < 0.8 * tot[0]
    sd = tr.state_dict()
    assert float((sd['rpn_head.rpn_cls.weight'] - m._sd['rpn_head.rpn_cls.weight']).abs().max()) > 0
    # the trained heads run through the inference path
    m.load_state_dict(sd)
    out = m.simple_test(**b, rescale=True)
    assert len(out) == 1 and 'dt_scores' in out[0]


# ---------------------------------------------------------------- backward pieces in isolation (identical inputs)
def test_shared_head_backward_alone_matches_autograd():
    """The three Bottlenecks with train-mode BatchNorm on identical inputs: gradients of every weight to 2e-4.  The
    reference side is torch.autograd over conv2d / batch_norm(training) with the ReLU masks of the HIP forward pass
    (out of ~5e5 activations per layer one or two have a pre-activation within 1e-5 of zero and take the other
    branch under MKL rounding; one such flip moves a channel's BatchNorm sums by ~1 %, tools/bwd_probe3.py)."""
    from fgn_amd.config import tiny_config
    from fgn_amd import train as TR
    cfg = tiny_config(3, 2, width_div=2)
    m, sd = _models(cfg)
    g = torch.Generator().manual_seed(21)
    C = cfg['roi_head']['shared_head']['inplanes']
    x = torch.randn(40, C, 7, 7, generator=g).abs()
    dout = torch.randn(40, C, 7, 7, generator=g)
    tr = TR.Trainer(m)
    tape, grads = [], {}
    got = TR.shared_head_train(m, x.permute(0, 2, 3, 1).contiguous().cuda(), 0.1, tape)
    TR._shared_backward(m, tr.W, tape, dout.permute(0, 2, 3, 1).contiguous().cuda(), grads)
    nchw = lambda t: t.permute(0, 3, 1, 2).cpu()
    ref_sd = {k: v.clone() for k, v in sd.items()}
    names = [k for k in ref_sd if k.startswith('roi_head.shared_head') and 'running_' not in k and 'num_batches' not in k]
    for k in names:
        ref_sd[k].requires_grad_(True)
    xb, flips = x, 0
    for b in range(3):
        p = f'roi_head.shared_head.{b}'
        bn = lambda t, i: F.batch_norm(t, None, None, ref_sd[f'{p}.bn{i}.weight'], ref_sd[f'{p}.bn{i}.bias'], True, 0.1, 1e-5)

        def relu_as(pre, key):
            nonlocal flips
            mask = nchw(tape[b][key]) > 0
            flips += int((mask != (pre.detach() > 0)).sum())
            return pre * mask
        y1 = relu_as(bn(F.conv2d(xb, ref_sd[p + '.conv1.weight']), 1), 'y1')
        y2 = relu_as(bn(F.conv2d(y1, ref_sd[p + '.conv2.weight'], padding=1), 2), 'y2')
        xb = relu_as(bn(F.conv2d(y2, ref_sd[p + '.conv3.weight']), 3) + xb, 'out')
    assert float((nchw(got) - xb.detach()).abs().max()) <= 2e-5 * float(xb.abs().max())
    assert flips <= 20
    (xb * dout).sum().backward()
    rep = _grad_report(grads, {k: ref_sd[k].grad for k in names})
    print(f'shared head alone ({flips} ReLU sign flips between the forward passes):',
          sorted(rep.items(), key=lambda kv: -kv[1][0])[:3])
    assert max(v[0] for v in rep.values()) <= 2e-4, rep


def test_relation_head_backward_alone_matches_autograd():
    from fgn_amd import ops
    from oracle import fgn_ref_cpu as O
    g = torch.Generator().manual_seed(22)
    R, N, C, B = 9, 3, 256, 2
    Q = torch.randn(R, C, 7, 7, generator=g, requires_grad=True)
    S = torch.randn(B * N, C, 7, 7, generator=g, requires_grad=True)
    gw = (torch.rand(C, generator=g) + 0.5).requires_grad_(True)
    gb = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    fcw = (torch.randn(6, C, generator=g) * 0.1).requires_grad_(True)
    img = torch.tensor([0, 0, 1, 1, 1, 0, 1, 0, 0])
    d6 = torch.randn(R * N, 6, generator=g)
    z = (Q[:, None] + S.view(B, N, C, 7, 7)[img]).reshape(R * N, C, 7, 7)
    y = torch.relu(torch.nn.functional.group_norm(z, 8, gw, gb, 1e-5))
    pooled = y.mean(dim=(2, 3))
    out6 = pooled @ fcw.t()
    (out6 * d6).sum().backward()
    nhwc = lambda t: t.detach().permute(0, 2, 3, 1).contiguous().cuda()
    rois = torch.cat([img.float()[:, None], torch.zeros(R, 4)], 1).cuda()
    dQ, dZ, pl, dgw, dgb = ops.relation_gn_head_backward(nhwc(Q), nhwc(S), rois, gw.detach().cuda(), gb.detach().cuda(),
                                                         fcw.detach().cuda(), d6.cuda(), N, 8, 1e-5)
    rel = lambda a, b: float((a - b).abs().max()) / float(b.abs().max())
    assert rel(dQ.permute(0, 3, 1, 2).cpu(), Q.grad) <= 2e-5
    assert rel(pl.cpu(), pooled.detach()) <= 2e-5
    assert rel(dgw.cpu(), gw.grad) <= 2e-5 and rel(dgb.cpu(), gb.grad) <= 2e-5
    dS = torch.zeros(B * N, 7, 7, C)
    dZc = dZ.cpu().view(R, N, 7, 7, C)
    for r in range(R):
        dS[int(img[r]) * N:(int(img[r]) + 1) * N] += dZc[r]
    assert rel(dS.permute(0, 3, 1, 2), S.grad) <= 2e-5
    assert rel((d6.t() @ pl.cpu()), fcw.grad) <= 2e-5


def test_forward_train_edge_cases_match_oracle():
    """An image without any ground truth beside one whose objects all belong to class 0 (two guided passes of every
    image see no GT at all: every inside anchor is a negative), and proposals handed in with a score column."""
    from fgn_amd.config import tiny_config
    from fgn_amd.episodes import make_batch
    from oracle import fgn_train_cpu as T
    cfg = tiny_config(3, 2, width_div=2)
    m, sd = _models(cfg)
    b = make_batch(4, 2, 3, 2, 160, 224, 64)
    b['qry_cat_ids'][0] = torch.zeros_like(b['qry_cat_ids'][0])
    b['qry_bboxes'][1] = b['qry_bboxes'][1][:0]
    b['qry_cat_ids'][1] = b['qry_cat_ids'][1][:0]
    b['qry_isegmaps'][1] = b['qry_isegmaps'][1][:0]
    tr_ref = {}
    torch.manual_seed(9)
    ref = T.forward_train(copy.deepcopy(sd), cfg, trace=tr_ref, **b)
    m.debug_trace = {}
    torch.manual_seed(9)
    got = m.forward_train(**b)
    s1 = m.debug_trace['samples'][1]
    assert s1['pos_inds'].numel() == 0 and s1['neg_inds'].numel() == cfg['train_cfg']['rcnn']['num']
    for s, r in zip(m.debug_trace['samples'], tr_ref['samples']):
        assert np.array_equal(s['pos_inds'].cpu().numpy(), r['pos_inds'].numpy())
        assert np.array_equal(s['neg_inds'].cpu().numpy(), r['neg_inds'].numpy())
    _compare_losses(got, ref, 1e-4)
    # no ground truth anywhere: zero box / mask losses, like `bbox_pred[pos_inds].sum()` / `mask_pred.sum()` of nothing
    b['qry_bboxes'][0], b['qry_cat_ids'][0] = b['qry_bboxes'][0][:0], b['qry_cat_ids'][0][:0]
    b['qry_isegmaps'][0] = b['qry_isegmaps'][0][:0]
    torch.manual_seed(9)
    ref = T.forward_train(copy.deepcopy(sd), cfg, **b)
    m._PT = None
    torch.manual_seed(9)
    got = m.forward_train(**b)
    assert float(got['loss_bbox']) == 0.0 and float(got['loss_mask']) == 0.0 and float(got['loss_rpn_bbox'][0]) == 0.0
    _compare_losses(got, ref, 1e-4)


def test_training_produces_a_working_detector():
    """Functional end-to-end check of forward_train + backward + Adagrad + re-pack + the inference path: the heads
    (full-width R50-C4 model, frozen randomly initialised backbone) are overfitted on two cluttered-character
    episodes for 250 steps; simple_test on those episodes then finds the objects (AP50 0 before, 1.0 measured after;
    tools/overfit_probe.py prints the trajectory)."""
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import collate
    from fgn_amd.fewshot_ds import ClutteredCharsFewShotISEG
    from fgn_amd.fsiseg_eval import evaluate_results
    from fgn_amd.train import Trainer
    ds = ClutteredCharsFewShotISEG('OMNIISEG', 3, 1, n_imgs=8, img_size=256, batch=2)
    batch = collate([ds[i] for i in range(2)])
    m = FGN(3, 1)
    before = evaluate_results(m.simple_test(**batch, rescale=True), 3)
    tr = Trainer(m, lr=0.005)
    first = last = None
    for it in range(250):
        torch.manual_seed(it)
        L = tr.step(batch)
        tot = sum(_f(v) for k, v in L.items() if 'loss' in k)
        first = tot if first is None else first
        last = tot
    live = evaluate_results(m.simple_test(**batch, rescale=True), 3)      # straight after the last step: the trainer's weights
    m.load_state_dict(tr.state_dict())
    after = evaluate_results(m.simple_test(**batch, rescale=True), 3)       # and through a checkpoint round trip
    assert live == after
    print('summed loss', round(first, 3), '->', round(last, 3), '| AP50 before', before, 'after', after)
    assert last < 0.5 * first
    assert before['bbox_mAP50'] < 0.2
    assert after['bbox_mAP50'] >= 0.6 and after['segm_mAP50'] >= 0.6


def test_checkpoint_resume_continues_the_run():
    """Two steps, checkpoint (weights + Adagrad sums + running statistics), a fresh Trainer resumed from it: the third
    step gives the same losses and the same weights as the uninterrupted run."""
    from fgn_amd.config import tiny_config
    from fgn_amd.episodes import make_batch
    from fgn_amd.train import Trainer
    cfg = tiny_config(3, 2, width_div=2)
    b = make_batch(2, 1, 3, 2, 160, 224, 64)
    m1, _ = _models(cfg)
    t1 = Trainer(m1)
    for it in range(2):
        torch.manual_seed(it)
        t1.step(b)
    ck = t1.checkpoint(meta={'iter': 2})
    torch.manual_seed(2)
    l1 = t1.step(b)
    m2, _ = _models(cfg)
    t2 = Trainer(m2)
    t2.resume(ck)
    torch.manual_seed(2)
    l2 = t2.step(b)
    # same losses and weights (the rocBLAS weight-gradient GEMMs may pick split-K kernels that accumulate with atomics,
    # so "same" is to the last few bits, not necessarily bitwise)
    for k in l1:
        assert _f(l1[k]) == pytest.approx(_f(l2[k]), rel=1e-5, abs=1e-7), k
    for k in t1.W:
        d = float((t1.W[k] - t2.W[k]).abs().max())
        assert d <= 1e-5 * max(1.0, float(t1.W[k].abs().max())), (k, d)
    assert set(ck) == {'state_dict', 'optimizer', 'meta'} and ck['meta']['iter'] == 2


def test_checkpoint_at_a_decayed_lr_keeps_the_schedule_base():
    """mmcv's LrUpdaterHook keeps the schedule's base in every param group as ``initial_lr`` and derives each
    later lr from it (``group.setdefault('initial_lr', group['lr'])`` on resume).  A checkpoint taken at a decayed lr
    must therefore carry both: ``lr`` = the scheduled value, ``initial_lr`` = base x lr_mult; a resumed Trainer gets
    both back, and a checkpoint without the key (written before any schedule hook ran) reads lr as the base."""
    from fgn_amd.config import tiny_config
    from fgn_amd.train import Trainer, step_lr
    cfg = tiny_config(3, 2, width_div=2)
    m, _ = _models(cfg)
    t = Trainer(m, lr=0.005)
    t.set_lr(step_lr(t.base_lr, it=500, epoch=3))               # after the step at epoch 3: 0.1 x base
    assert t.lr == pytest.approx(0.0005) and t.base_lr == 0.005
    opt = t.optimizer_state_dict()
    names = opt['param_names']
    g_rpn = opt['param_groups'][names.index('rpn_head.rpn_conv.weight')]
    g_roi = opt['param_groups'][names.index('roi_head.bbox_head.fc_cls.weight')]
    g_bb = opt['param_groups'][0]
    assert g_rpn['lr'] == pytest.approx(0.0005) and g_rpn['initial_lr'] == pytest.approx(0.005)
    assert g_roi['lr'] == pytest.approx(0.00005) and g_roi['initial_lr'] == pytest.approx(0.0005)      # lr_mult 0.1
    assert g_bb['lr'] == pytest.approx(0.0005) and g_bb['initial_lr'] == pytest.approx(0.005)
    # mmcv's resume on these groups: the regular lr of epoch 3 is derived from initial_lr - once
    assert step_lr(g_rpn['initial_lr'], it=500, epoch=3) == pytest.approx(g_rpn['lr'])
    m2, _ = _models(cfg)
    t2 = Trainer(m2, lr=0.123)
    t2.resume(t.checkpoint())
    assert t2.lr == pytest.approx(0.0005) and t2.base_lr == pytest.approx(0.005)
    for g in opt['param_groups']:
        del g['initial_lr']
    t2.resume({'state_dict': t.state_dict(), 'optimizer': opt})
    assert t2.lr == pytest.approx(0.0005) and t2.base_lr == pytest.approx(0.0005)


def test_adagrad_multi_more_tensors_than_one_table_with_empty_ones():
    """``fgn_adagrad_multi_f32`` packs up to 64 tensors per launch and skips empty ones without taking a slot;
    with more than 64 list entries and empty tensors among them every tensor must still be updated exactly ONCE."""
    from fgn_amd import ops
    g = torch.Generator().manual_seed(0)
    sizes = [0 if i % 7 == 3 else 1 + (i * 37) % 5000 for i in range(150)]
    P = [torch.randn(n, generator=g).cuda() for n in sizes]
    G = [torch.randn(n, generator=g).cuda() for n in sizes]
    S = [torch.rand(n, generator=g).cuda() for n in sizes]
    lrs = [0.005 * (0.1 if i % 2 else 1.0) for i in range(len(sizes))]
    want_p, want_s = [], []
    for p, gr, s_, lr in zip(P, G, S, lrs):
        gv = gr + 1e-5 * p
        st = s_ + gv * gv
        want_s.append(st)
        want_p.append(p - lr * gv / (st.sqrt() + 1e-10))
    ops.adagrad_multi(P, G, S, lrs, 1e-5, 1e-10)
    torch.cuda.synchronize()
    for i, (p, s_, wp, ws) in enumerate(zip(P, S, want_p, want_s)):
        assert torch.allclose(s_, ws, rtol=1e-6, atol=1e-9), i
        assert torch.allclose(p, wp, rtol=1e-5, atol=1e-8), i


def test_step_without_positives_keeps_every_gradient_and_torch_optimizer_layout():
    """(1) A batch without any ground truth samples no positive RoI: the mask head (and the box regression) get ZERO
    gradients - not missing ones - so the data-parallel bucket has the same keys on every rank and Adagrad still
    applies its weight decay (torch: g = 0 + wd * p), as the reference's ``mask_pred.sum() * 0`` losses do.
    (2) The checkpoint's optimizer entry is ``torch.optim.Adagrad.state_dict()`` layout: torch loads it, and
    ``resume`` accepts torch's own.  (3) A re-pack after training (device / Winograd switch / ``_packed_device`` reset)
    starts from the trainer's weights, not from the initial state dict."""
    from fgn_amd.config import tiny_config
    from fgn_amd.episodes import make_batch
    from fgn_amd.train import Trainer
    cfg = tiny_config(3, 2, width_div=2)
    b = make_batch(4, 2, 3, 2, 160, 224, 64)
    empty = copy.deepcopy(b)
    for i in range(2):
        empty['qry_bboxes'][i], empty['qry_cat_ids'][i] = b['qry_bboxes'][i][:0], b['qry_cat_ids'][i][:0]
        empty['qry_isegmaps'][i] = b['qry_isegmaps'][i][:0]
    m, _ = _models(cfg)
    t = Trainer(m)
    torch.manual_seed(0)
    t.step(b)
    name = 'roi_head.mask_head.convs.1.conv.weight'
    w0, s0 = t.W[name].clone(), t.state[name].clone()
    torch.manual_seed(1)
    t.step(empty)
    assert set(t.grads) == set(t.W)
    assert float(t.grads[name].abs().max()) == 0.0 and float(t.grads['rpn_head.rpn_conv.weight'].abs().max()) > 0.0
    # torch.optim.Adagrad on a zero gradient with weight decay: sum += (wd p)^2, p -= lr * wd p / (sqrt(sum) + eps)
    p = torch.nn.Parameter(w0.clone())
    opt = torch.optim.Adagrad([p], lr=t.lr * t.mult, weight_decay=t.wd, eps=t.eps)
    opt.state[p]['sum'].copy_(s0)
    p.grad = torch.zeros_like(p)
    opt.step()
    assert torch.allclose(t.W[name], p.detach(), rtol=1e-6, atol=1e-9) and not torch.equal(t.W[name], w0)
    assert torch.allclose(t.state[name], opt.state[p]['sum'], rtol=1e-6, atol=0)
    # ---- torch layout of the optimizer entry: the index space is the REFERENCE model's named_parameters() - frozen
    # backbone and the unused layer4 included (mmcv lists frozen parameters under paramwise_cfg, Adagrad creates state
    # for every one) - so a real torch.optim.Adagrad built over ALL parameters loads it, as the reference's resume does
    from fgn_amd.train import reference_param_order
    ck = t.checkpoint()
    order = reference_param_order(m._sd, cfg['backbone'])
    names = ck['optimizer']['param_names']
    assert names == [k for k, _ in order] and len(names) > len(t.W) and ck['meta']['iter'] == 2
    assert names[0] == 'backbone.conv1.weight' and any(k.startswith('backbone.layer4.') for k in names)
    params = [torch.nn.Parameter(ck['state_dict'][k].clone() if k in ck['state_dict'] else torch.zeros(shape),
                                 requires_grad=k in t.W) for k, shape in order]
    topt = torch.optim.Adagrad([{'params': [q]} for q in params], lr=t.lr, weight_decay=t.wd, eps=t.eps)
    topt.load_state_dict({k: v for k, v in ck['optimizer'].items() if k != 'param_names'})
    for q, k in zip(params, names):
        if k in t.W:
            assert torch.equal(topt.state[q]['sum'], t.state[k].cpu()) and float(topt.state[q]['step']) == 2.0
        else:                                              # torch's own initial state of a parameter without gradient
            assert float(topt.state[q]['sum'].abs().max()) == 0.0 and float(topt.state[q]['step']) == 0.0
            assert topt.state[q]['sum'].shape == q.shape
    assert topt.param_groups[names.index(name)]['lr'] == pytest.approx(t.lr * t.mult)
    assert topt.param_groups[names.index('rpn_head.rpn_conv.weight')]['lr'] == pytest.approx(t.lr)
    assert topt.param_groups[0]['lr'] == pytest.approx(t.lr)
    bn_key = 'roi_head.shared_head.0.bn1.num_batches_tracked'
    assert int(ck['state_dict'][bn_key]) == int(m._sd[bn_key]) + 4          # 2 steps x (RoI batch + support batch)
    m2, _ = _models(cfg)
    t2 = Trainer(m2)
    t2.resume({'state_dict': ck['state_dict'], 'optimizer': topt.state_dict(), 'meta': ck['meta']})   # torch's own dict
    assert t2.n_steps == 2
    for k in t.W:
        assert torch.equal(t2.state[k], t.state[k]) and torch.equal(t2.W[k], t.W[k])
    assert int(t2.state_dict()[bn_key]) == int(ck['state_dict'][bn_key])   # the loaded counter is the base, no offset
    # a torch optimizer over the trainable heads alone (no frozen entries) maps by state-dict order, as before
    hp = [torch.nn.Parameter(ck['state_dict'][k].clone()) for k in t.W]
    hopt = torch.optim.Adagrad([{'params': [q]} for q in hp], lr=t.lr, weight_decay=t.wd, eps=t.eps)
    for q, k in zip(hp, t.W):
        hopt.state[q]['sum'].copy_(t.state[k].cpu())
        hopt.state[q]['step'] = torch.tensor(2.0)
    m3, _ = _models(cfg)
    t3 = Trainer(m3)
    t3.resume({'state_dict': ck['state_dict'], 'optimizer': hopt.state_dict()})
    assert t3.n_steps == 2 and all(torch.equal(t3.state[k], t.state[k]) for k in t.W)
    # an index space that is neither is refused, not guessed
    bad = hopt.state_dict()
    bad['param_groups'] = bad['param_groups'][:-1]
    with pytest.raises(ValueError):
        t3.resume({'state_dict': ck['state_dict'], 'optimizer': bad})
    del m3, t3
    # ---- re-pack after training
    want = m.simple_test(**b, rescale=True)
    m._packed_device = None                          # what a device change / use_winograd switch does
    got = m.simple_test(**b, rescale=True)
    for w, g in zip(want, got):
        assert np.array_equal(w['dt_scores'], g['dt_scores']) and np.array_equal(w['dt_bboxes'], g['dt_bboxes'])
    torch.manual_seed(2)
    l_a = t.step(b)                                   # forward_train after the re-pack: trainer's weights, too
    torch.manual_seed(2)
    l_b = t2.step(b)
    for k in l_a:
        assert _f(l_a[k]) == pytest.approx(_f(l_b[k]), rel=1e-5, abs=1e-7), k
    fresh = m.state_dict()
    assert torch.equal(fresh[name], t.W[name].cpu())   # the model's state dict is the trainer's


def test_small_gemm_matches_fp64():
    """``fgn_gemm_small_f32``: the three products of a training step whose shapes the MFMA kernels do not take (6-row fc
    weight gradient, data gradient through the fc, the 75-channel AG-RPN head with a row-strided operand)."""
    from fgn_amd import ops
    g = torch.Generator().manual_seed(4)
    for (m, k, n, trans, stride_pad) in ((6, 384, 1024, True, 0), (384, 6, 1024, False, 0), (190, 75, 1024, False, 1),
                                         (1, 1, 5, False, 0), (7, 33, 3, True, 0)):
        a_full = torch.randn((k, m + stride_pad) if trans else (m, k + stride_pad), generator=g).cuda()
        a = a_full[:, :m] if trans else a_full[:, :k]
        b = torch.randn(k, n, generator=g).cuda()
        got = ops.gemm_small(a, b, trans_a=trans)
        ref = ((a.t() if trans else a).double() @ b.double())
        assert got.shape == (m, n)
        assert (got.double() - ref).abs().max().item() <= 2e-6 * max(ref.abs().max().item(), 1.0) * max(k, 1) ** 0.5
        assert torch.equal(got, ops.gemm_small(a, b, trans_a=trans))


@pytest.mark.parametrize('R,M,N', [(6272, 1024, 512), (441, 512, 4608), (200, 1024, 9216), (1000, 76, 1024), (37, 8, 12),
                                   (300, 1024, 4), (5000, 4, 256), (33, 64, 64)])
def test_weight_gradient_gemm_matches_fp64(R, M, N):
    """fgn_gemm_tn_f32 (dW = dY^T X on fp32 MFMA, row slabs reduced in a fixed order): 1e-5 of the result range vs fp64,
    bit-identical run to run."""
    from fgn_amd import ops
    g = torch.Generator().manual_seed(R + M + N)
    a, b = torch.randn(R, M, generator=g), torch.randn(R, N, generator=g)
    ref = a.double().t() @ b.double()
    got = ops.gemm_tn(a.cuda(), b.cuda())
    assert float((got.cpu().double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    assert torch.equal(got, ops.gemm_tn(a.cuda(), b.cuda()))


def test_in_place_repack_equals_a_fresh_pack():
    """Round 4: between two steps ``Trainer.refresh`` writes the updated weights INTO the packed layers (one strided copy
    per convolution, ``fgn_winograd_pack_weights_f32`` per Winograd layer) instead of re-building them with torch ops.
    (1) the device transform == the host transform of ``pack_winograd`` (fp64 arithmetic, one rounding: at most the
    last bit of an element differs, by the summation order inside fp64); every other packed tensor is bitwise the fresh
    pack's.  (2) three steps with the in-place re-pack end on the weights of three steps with fresh packs."""
    from fgn_amd import ops
    from fgn_amd.config import tiny_config
    from fgn_amd.episodes import make_batch
    from fgn_amd.train import Trainer
    g = torch.Generator().manual_seed(0)
    for m, (cout, cin) in ((4, (96, 64)), (2, (64, 128)), (4, (256, 32))):
        w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).cuda()
        b = torch.randn(cout, generator=g).cuda()
        fresh = ops.pack_winograd(w, bias=b, relu=True, m=m)
        w2, b2 = (w * 1.7 + 0.01).contiguous(), b + 1.0
        want = ops.pack_winograd(w2, bias=b2, relu=True, m=m)
        ops.repack_winograd_(fresh, w2, b2)
        assert torch.equal(fresh.shift, want.shift)
        d = (fresh.u - want.u).abs().max().item()
        assert d <= 2.5e-7 * want.u.abs().max().item(), d
        assert float((fresh.u != want.u).float().mean()) < 0.01
        assert float(fresh.u[:, cout:].abs().max()) == 0.0 if fresh.cout_pad > cout else True
    cfg = tiny_config(3, 2, width_div=2)
    bt = make_batch(2, 1, 3, 2, 160, 224, 64)
    res = {}
    for mode in ('1', '0'):
        os.environ['FGN_TRAIN_REPACK_IN_PLACE'] = mode
        try:
            m1, _ = _models(cfg)
            t = Trainer(m1)
            for it in range(3):
                torch.manual_seed(it)
                t.step(bt)
            res[mode] = ({k: v.clone() for k, v in t.W.items()},
                         {k: (v.w.clone() if hasattr(v, 'w') else v.clone()) for k, v in m1._P.items()
                          if k in ('rpn_conv', 'rpn_head', 'rel_q', 'rel_s', 'upsample', 'fc_w', 'fc_b', 'gn_w', 'logit_w')})
        finally:
            os.environ.pop('FGN_TRAIN_REPACK_IN_PLACE', None)
    # Adagrad's first steps move a weight by ~lr whatever the size of its gradient, so a last-bit difference of a Winograd
    # weight that flips one ReLU changes single elements by up to a step (5e-4 under roi_head) in each of the steps that
    # follow the first re-pack (two of the three); all but a handful agree to 1e-5, none is further apart than those two
    # steps.  A flipped ReLU of an output channel that is alive at a few RoI positions only changes that channel's whole
    # weight ROW (its gradient is a sum over those positions): which rows, and how many, depends on the last bits of the
    # frozen backbone's kernels (round 4: one row, <= 0.2 % of a tensor; round 5's fused shortcuts: ten rows, 2 %)
    for k in res['1'][0]:
        d = (res['1'][0][k] - res['0'][0][k]).abs()
        rows = (d.reshape(d.shape[0], -1) > 1e-5).any(1).float().mean().item() if d.dim() > 1 else 0.0
        assert d.max().item() <= 1.05e-3 and (d > 1e-5).float().mean().item() <= 5e-2, (k, d.max().item(), rows)
    for k in res['1'][1]:
        a, b_ = res['1'][1][k], res['0'][1][k]
        assert a.shape == b_.shape and (a - b_).abs().max().item() <= 1.05e-3, k


@pytest.mark.gpu
def test_adagrad_of_all_tensors_in_one_launch_equals_one_launch_per_tensor():
    from fgn_amd import ops
    g = torch.Generator().manual_seed(5)
    sizes = [1, 3, 4095, 4096, 4097, 75 * 1024, 9437184 // 8, 6, 0, 512]
    mk = lambda: [torch.randn(n, generator=g).cuda() for n in sizes]
    P1, G1, S1 = mk(), mk(), [t.abs() for t in mk()]
    P2, S2 = [t.clone() for t in P1], [t.clone() for t in S1]
    lrs = [0.01 * (1 + i % 3) for i in range(len(sizes))]
    for p, gr, s, lr in zip(P1, G1, S1, lrs):
        ops.adagrad_step(p, gr, s, lr, 1e-5, 1e-10)
    ops.adagrad_multi(P2, G1, S2, lrs, 1e-5, 1e-10)
    for a, b in zip(P1 + S1, P2 + S2):
        assert torch.equal(a, b)
    # more tensors than one table holds
    many = 150
    P3 = [torch.randn(17 + i, generator=g).cuda() for i in range(many)]
    G3 = [torch.randn(17 + i, generator=g).cuda() for i in range(many)]
    S3 = [torch.zeros(17 + i).cuda() for i in range(many)]
    P4, S4 = [t.clone() for t in P3], [t.clone() for t in S3]
    for p, gr, s in zip(P3, G3, S3):
        ops.adagrad_step(p, gr, s, 0.005, 1e-5, 1e-10)
    ops.adagrad_multi(P4, G3, S4, [0.005] * many, 1e-5, 1e-10)
    for a, b in zip(P3 + S3, P4 + S4):
        assert torch.equal(a, b)


@pytest.mark.gpu
def test_pinned_staging_of_the_small_host_arrays():
    from fgn_amd import train as TR
    st = TR._Staging(nbytes=1 << 12)
    dev = torch.device('cuda')
    rng = np.random.default_rng(0)
    for call in range(3):
        st.next_call()
        arrs = [rng.integers(0, 1 << 40, 37).astype(np.int64), rng.random((5, 4)).astype(np.float32), np.zeros(0, np.int64),
                rng.integers(0, 255, 300).astype(np.uint8), rng.random(2000)]            # the last one does not fit: pageable path
        outs = [st.h2d(a, dev) for a in arrs]
        torch.cuda.synchronize()
        for a, t in zip(arrs, outs):
            assert t.dtype == torch.from_numpy(a[:0]).dtype and tuple(t.shape) == a.shape
            assert np.array_equal(t.cpu().numpy(), a)
