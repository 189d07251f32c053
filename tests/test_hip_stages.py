"""Stage-wise parity of the HIP kernels against the oracle on identical inputs.
Selection stages (top-k, decode, NMS, labels) are BIT-EXACT; feature stages carry an fp32
accumulation-order tolerance written in each test."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def _nchw(t):
    return t.permute(0, 3, 1, 2).contiguous()


def _close(got, ref, rel):
    d = (got - ref).abs().max().item()
    assert d <= rel * (ref.abs().max().item() + 1e-12), d


# ---------------------------------------------------------------- spatial
def test_nhwc4_and_maxpool():
    from fgn_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 64, 33, 47, generator=g)
    got = _nchw(ops.maxpool3x3s2(_nhwc(x).cuda()).cpu())
    assert torch.equal(got, F.max_pool2d(x, 3, 2, 1))


@pytest.mark.parametrize('aligned,sr', [(True, 0), (False, -1), (True, 2)])
def test_roi_align_matches_oracle(aligned, sr):
    from fgn_amd import ops
    from oracle import fgn_ref_cpu as O
    g = torch.Generator().manual_seed(1)
    fmap = torch.randn(2, 64, 20, 31, generator=g)
    rois = torch.tensor([[0, 8.3, 5.1, 200.7, 150.2], [1, 0, 0, 496, 320], [0, 100, 100, 101, 100.5],
                         [1, -20, -30, 40, 50], [0, 300, 200, 600, 400], [1, 17.5, 33.25, 18.0, 300.0],
                         [0, 50, 60, 50, 60]], dtype=torch.float32)
    ref = O.roi_align(fmap, rois.numpy(), 7, 1 / 16, sr, aligned)
    got = _nchw(ops.roi_align(_nhwc(fmap).cuda(), rois.cuda(), 7, 1 / 16, sr, aligned).cpu())
    _close(got, ref, 2e-6)
    # device count: rois past the count are not written
    cnt = torch.tensor([3], dtype=torch.int32, device='cuda')
    out = ops.roi_align(_nhwc(fmap).cuda(), rois.cuda(), 7, 1 / 16, sr, aligned, cnt)
    _close(_nchw(out[:3].cpu()), ref[:3], 2e-6)


def test_two_map_roi_align_is_identical_to_two_launches():
    """``ops.roi_align2``: the C4 map and the map of the shared head's commuted first conv pooled at the same RoIs by
    one launch (+ shift and ReLU on the second): the bytes of two single-map launches, incl. a device-side RoI count."""
    from fgn_amd import ops
    g = torch.Generator().manual_seed(8)
    f1 = torch.randn(2, 13, 17, 1024, generator=g).cuda()
    f2 = torch.randn(2, 13, 17, 512, generator=g).cuda()
    sh = torch.randn(512, generator=g).cuda()
    rois = torch.tensor([[0, 10.3, 20.7, 150.2, 140.9], [1, -30.0, -20.0, 90.0, 100.0], [0, 200.0, 150.0, 290.0, 230.0],
                         [1, 40.0, 40.0, 41.0, 42.0], [0, 0.0, 0.0, 272.0, 208.0], [1, 5.5, 3.25, 300.0, 260.0]]).cuda()
    for cnt in (None, torch.tensor([4], dtype=torch.int32, device='cuda')):
        a, b = ops.roi_align2(f1, f2, rois, 7, 1 / 16, 0, True, cnt, post_shift2=sh, relu2=True)
        n = 6 if cnt is None else 4
        assert torch.equal(a[:n], ops.roi_align(f1, rois, 7, 1 / 16, 0, True, cnt)[:n])
        assert torch.equal(b[:n], ops.roi_align(f2, rois, 7, 1 / 16, 0, True, cnt, post_shift=sh, relu=True)[:n])
    c, d = ops.roi_align2(f1[..., :64].contiguous(), f2[..., :32].contiguous(), rois, 7, 1 / 16, 2, False)
    assert torch.equal(c, ops.roi_align(f1[..., :64].contiguous(), rois, 7, 1 / 16, 2, False))
    assert torch.equal(d, ops.roi_align(f2[..., :32].contiguous(), rois, 7, 1 / 16, 2, False))


def test_roi_align_mask_matches_oracle():
    from fgn_amd import ops
    from oracle import fgn_ref_cpu as O
    g = torch.Generator().manual_seed(2)
    m = torch.rand(3, 1, 64, 64, generator=g) > 0.4
    rois = torch.tensor([[0, 6.5, 7.25, 57.0, 50.5], [1, 0, 0, 64, 64], [2, 20, 20, 20.5, 21]], dtype=torch.float32)
    ref = O.roi_align(m.float(), rois.numpy(), 7, 1.0, -1, False)[:, 0]
    got = ops.roi_align_mask(m[:, 0].to(torch.uint8).cuda().contiguous(), rois.cuda(), 7, 1.0, -1, False).cpu()
    _close(got, ref, 2e-6)


def test_support_reductions_match_golden_count_spp(golden_dir):
    """count_spp's reductions against the reference's own output (tests/golden/count_spp.npz)."""
    from fgn_amd import ops
    z = np.load(os.path.join(golden_dir, 'count_spp.npz'))
    n, k = int(z['n_ways']), int(z['k_shots'])
    fm = torch.from_numpy(z['spp_fmaps'])
    nk = fm.shape[0]
    rois = torch.cat([torch.arange(nk, dtype=torch.float32)[:, None], torch.from_numpy(z['spp_bboxes_xyxy'])], 1)
    masks7 = ops.roi_align_mask(torch.from_numpy(z['spp_isegmaps'])[:, 0].to(torch.uint8).cuda().contiguous(),
                                rois.cuda(), 7, 1.0, -1, False)
    feats = ops.roi_align(_nhwc(fm).cuda(), rois.cuda(), 7, 1 / 16, -1, False)
    cat_mean = ops.support_kmean(feats, nk // k, k)
    cat_mean_mp = ops.support_class_vectors(feats, masks7, nk // k, k)
    ref_mean = torch.from_numpy(z['cat_mean']).reshape(-1, *z['cat_mean'].shape[2:])
    _close(_nchw(cat_mean.cpu()), ref_mean, 5e-6)
    _close(cat_mean_mp.cpu().reshape(-1), torch.from_numpy(z['cat_mean_mp']).reshape(-1), 5e-6)


def test_class_vectors_match_oracle():
    from fgn_amd import ops
    from oracle import fgn_ref_cpu as O
    g = torch.Generator().manual_seed(3)
    spp = torch.randn(2 * 3 * 2, 96, 5, 6, generator=g)
    ref = O.ag_rpn_class_vectors(spp, 2, 3, 2).reshape(6, 96)
    got = ops.support_class_vectors(_nhwc(spp).cuda(), None, 6, 2).cpu()
    _close(got, ref, 5e-6)


# ---------------------------------------------------------------- AG-RPN merge vs reference golden
def test_ag_rpn_merge_matches_reference_golden(golden_dir):
    from fgn_amd import ops
    for name in ('ag_rpn.npz', 'ag_rpn_n1.npz'):
        z = np.load(os.path.join(golden_dir, name))
        n, k = int(z['n_ways']), int(z['k_shots'])
        qry, spp = torch.from_numpy(z['qry']), torch.from_numpy(z['spp'])
        b, c = qry.shape[:2]
        # pad channels 32 -> 32 (already a multiple of 32); run the HIP AG-RPN
        vec = ops.support_class_vectors(_nhwc(spp).cuda(), None, b * n, k)
        w = {kk.replace('__', '.'): torch.from_numpy(z[kk]) for kk in z.files if kk.startswith('rpn_head')}
        conv = ops.pack_conv(w['rpn_head.rpn_conv.weight'], bias=w['rpn_head.rpn_conv.bias'], pad=1, relu=True).to('cuda')
        head = ops.pack_conv(torch.cat([w['rpn_head.rpn_cls.weight'], w['rpn_head.rpn_reg.weight']], 0),
                             bias=torch.cat([w['rpn_head.rpn_cls.bias'], w['rpn_head.rpn_reg.bias']], 0)).to('cuda')
        ref_cls = torch.from_numpy(z['cls']).permute(0, 2, 3, 1).reshape(b, -1)          # (y,x,a)
        ref_reg = torch.from_numpy(z['reg']).permute(0, 2, 3, 1).reshape(b, -1, 4)
        # three forms of the guided 3x3 conv: the guidance multiply fused into the A-operand staging of the direct
        # kernel; the DEFAULT path of the detector (F(4x4,3x3) Winograd, guidance in wg4_input_kernel, grouped GEMM,
        # detector.py::_detect_body); F(2x2,3x3)
        guided = {'direct in_scale': lambda: ops.conv2d(_nhwc(qry).cuda(), conv, in_scale=vec, a_img_div=n)}
        for m in (4, 2):
            wg = ops.pack_winograd(w['rpn_head.rpn_conv.weight'], bias=w['rpn_head.rpn_conv.bias'], relu=True,
                                   m=m).to('cuda')
            guided[f'winograd F({m}x{m}) guided'] = \
                lambda wg=wg: ops.conv3x3_winograd(_nhwc(qry).cuda(), wg, in_scale=vec, a_img_div=n)
        for form, run in guided.items():
            y = ops.conv2d(run(), head)
            logits, scores, deltas = ops.rpn_merge(y, b, n, 15)
            dl = (logits.cpu() - ref_cls).abs().max().item() / ref_cls.abs().max().item()
            dd = (deltas.cpu() - ref_reg).abs().max().item() / ref_reg.abs().max().item()
            print(f'[{name} {form}] rel err logits {dl:.1e}, deltas {dd:.1e}')
            _close(logits.cpu(), ref_cls, 2e-5)
            # the arg-max choice is discrete: compare deltas where the winner is unambiguous
            _close(deltas.cpu(), ref_reg, 1e-4)


# ---------------------------------------------------------------- relation head
@pytest.mark.parametrize('gw,n', [(32, 3), (16, 3), (8, 3), (32, 1), (32, 2), (32, 5), (16, 8)])
def test_relation_head_matches_oracle(gw, n):
    """gw = channels per GroupNorm group: 32 is the reference's GN(32, 1024); 16 / 8 are narrower heads (the
    ResNet-18 variant of cfg2: GN(32, 256)).  n = ways: the kernel is unrolled per class count (1..8; the next
    class's support slab is loaded while the current one is normalised)."""
    from fgn_amd import ops
    from oracle import fgn_ref_cpu as O
    g = torch.Generator().manual_seed(5)
    c, r, b = 128, 9, 2
    feats = torch.randn(r, c, 7, 7, generator=g).abs()
    cat_mean = torch.randn(b, n, c, 7, 7, generator=g).abs()
    rois = torch.zeros(r, 5)
    rois[:, 0] = torch.tensor([0, 1, 0, 1, 1, 0, 0, 1, 0])
    sd = {'roi_head.cls_reg_shared_conv.weight': torch.randn(c, 2 * c, 1, 1, generator=g) / (2 * c) ** 0.5,
          'roi_head.cls_reg_shared_conv.bias': torch.randn(c, generator=g) * 0.1,
          'roi_head.cls_reg_shared_conv_norm.weight': torch.rand(c, generator=g) + 0.5,
          'roi_head.cls_reg_shared_conv_norm.bias': torch.randn(c, generator=g) * 0.1,
          'roi_head.bbox_head.fc_cls.weight': torch.randn(2, c, generator=g) * 0.1,
          'roi_head.bbox_head.fc_cls.bias': torch.randn(2, generator=g) * 0.1,
          'roi_head.bbox_head.fc_reg.weight': torch.randn(4, c, generator=g) * 0.1,
          'roi_head.bbox_head.fc_reg.bias': torch.randn(4, generator=g) * 0.1}
    cfg = {'roi_head': {'relation': {'gn_groups': c // gw, 'gn_eps': 1e-5}}}
    rel = O.relation(feats, rois.numpy(), cat_mean, sd, cfg, n)
    ref_cls, ref_reg = O.bbox_head_forward(rel, sd)
    wq = ops.pack_conv(sd['roi_head.cls_reg_shared_conv.weight'][:, :c].contiguous()).to('cuda')
    ws = ops.pack_conv(sd['roi_head.cls_reg_shared_conv.weight'][:, c:].contiguous(),
                       bias=sd['roi_head.cls_reg_shared_conv.bias']).to('cuda')
    Q = ops.conv2d(_nhwc(feats).cuda(), wq)
    S = ops.conv2d(_nhwc(cat_mean.view(b * n, c, 7, 7)).cuda(), ws)
    fc_w = torch.cat([sd['roi_head.bbox_head.fc_cls.weight'], sd['roi_head.bbox_head.fc_reg.weight']]).cuda()
    fc_b = torch.cat([sd['roi_head.bbox_head.fc_cls.bias'], sd['roi_head.bbox_head.fc_reg.bias']]).cuda()
    cls, reg = ops.relation_gn_head(Q, S, rois.cuda(), sd['roi_head.cls_reg_shared_conv_norm.weight'].cuda(),
                                    sd['roi_head.cls_reg_shared_conv_norm.bias'].cuda(), fc_w, fc_b, n, c // gw, 1e-5)
    _close(cls.cpu(), ref_cls, 2e-5)
    _close(reg.cpu(), ref_reg, 2e-5)


# ---------------------------------------------------------------- proposals: bit-exact
def _proposal_case(seed, fh, fw, nms_pre, max_out, ties, img=None, min_size=0):
    from fgn_amd import ops
    from fgn_amd.config import fgn_r50_c4_config, with_caps
    from oracle import fgn_ref_cpu as O
    cfg = with_caps(fgn_r50_c4_config(3, 3), nms_pre=nms_pre, rpn_max=max_out)
    cfg['test_cfg']['rpn']['min_bbox_size'] = min_size
    g = torch.Generator().manual_seed(seed)
    cls = torch.randn(15, fh, fw, generator=g) * 3
    if ties:
        cls = (cls * 2).round() / 2            # many exactly equal scores, incl. saturated sigmoid
        cls[:, : fh // 2] += 20
    reg = torch.randn(60, fh, fw, generator=g) * 0.5
    ih, iw = img if img else (fh * 16 - 3, fw * 16 - 5)
    ref = O.rpn_get_bboxes(cls.numpy(), reg.numpy(), np.array([ih, iw, 3]), cfg)
    logits = cls.permute(1, 2, 0).reshape(1, -1).contiguous()
    scores = torch.from_numpy(O.sigmoid32(logits.numpy()))
    deltas = reg.permute(1, 2, 0).reshape(1, -1, 4).contiguous()
    rp = cfg['rpn_head']
    anchors = torch.from_numpy(ops.base_anchors(rp['anchor_scales'], rp['anchor_ratios'], rp['anchor_stride']))
    assert np.array_equal(anchors.numpy(), O.base_anchors(rp['anchor_scales'], rp['anchor_ratios'], 16))
    props, n = ops.rpn_proposals(scores.cuda(), deltas.cuda(), anchors.cuda(), fh, fw, 16, ih, iw,
                                 rp['target_means'], rp['target_stds'], nms_pre, min_size, 0.7, max_out)
    n = int(n.item())
    got = props[0, :n].cpu().numpy()
    assert n == len(ref), (n, len(ref))
    assert np.array_equal(got, ref)            # boxes AND scores bitwise
    assert float(props[0, n:].abs().sum()) == 0.0


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_proposals_bit_exact_small(seed):
    _proposal_case(seed, 8, 10, 6000, 300, ties=False)       # 1200 anchors < nms_pre: no top-k cut


@pytest.mark.parametrize('ties', [False, True])
def test_proposals_bit_exact_cfg3_size(ties):
    _proposal_case(7, 50, 84, 6000, 300, ties, img=(800, 1333))    # 63 000 anchors -> 6000 -> 300


def test_proposals_bit_exact_with_boxes_under_the_minimum_size():
    """Candidates that fail the min-size test sit between the ranked ones: never kept, never suppressing (the
    multi-workgroup IoU matrix carries them as invalid rows / columns)."""
    _proposal_case(11, 50, 84, 6000, 300, ties=False, img=(800, 1333), min_size=40)
    _proposal_case(12, 50, 84, 6000, 300, ties=True, img=(800, 1333), min_size=90)


@pytest.mark.parametrize('shift', [10.0, 3.0])
def test_proposals_bit_exact_on_suppression_chains(shift):
    """Worst case of the wavefront NMS: the ranked boxes form staircases (box k overlaps only its neighbours, or with
    the smaller shift its next few), so inside a 64-candidate chunk every decision hangs on the one before - as many
    passes as lanes -, and the chains run across chunk boundaries.  Expected: the oracle's sequential greedy NMS."""
    from fgn_amd import ops
    from fgn_amd.config import fgn_r50_c4_config
    from oracle import fgn_ref_cpu as O
    from oracle import fgn_train_cpu as OT
    cfg = fgn_r50_c4_config(3, 3)
    rp = cfg['rpn_head']
    fh, fw, ih, iw, A = 50, 84, 800, 1333, 15
    base = O.base_anchors(rp['anchor_scales'], rp['anchor_ratios'], rp['anchor_stride'])
    anchors = O.grid_anchors(base, fh, fw, rp['anchor_stride'])                 # [fh*fw*A, 4], index = (y*fw + x)*A + a
    n_chain, per_row = 1200, 120
    k = np.arange(n_chain)
    bx = 20.0 + shift * (k % per_row)
    by = 20.0 + 72.0 * (k // per_row)
    target = np.stack([bx, by, bx + 64.0, by + 64.0], 1).astype(np.float32)
    # one anchor per chain box: the anchor whose centre is nearest, a different one for every box
    cx, cy = (target[:, 0] + target[:, 2]) / 2, (target[:, 1] + target[:, 3]) / 2
    px = np.clip(np.round(cx / 16).astype(int), 0, fw - 1)
    py = np.clip(np.round(cy / 16).astype(int), 0, fh - 1)
    used, idx = set(), []
    for i in range(n_chain):
        for a in range(A):
            j = (py[i] * fw + px[i]) * A + a
            if j not in used:
                used.add(j); idx.append(j)
                break
        else:
            raise AssertionError('ran out of anchors at a pixel')
    idx = np.array(idx)
    deltas = np.zeros((fh * fw * A, 4), np.float32)
    deltas[idx] = OT.bbox2delta(torch.from_numpy(anchors[idx]), torch.from_numpy(target), rp['target_means'],
                                rp['target_stds']).numpy()
    # (background anchors: distinct low scores - a block of equal keys would send the stage to its fallback path)
    logits = (np.random.RandomState(3).randn(fh * fw * A) - 12.0).astype(np.float32)
    logits[idx] = np.linspace(9.0, 3.0, n_chain).astype(np.float32)            # rank = chain order
    cls = torch.from_numpy(logits.reshape(fh, fw, A)).permute(2, 0, 1).contiguous()
    reg = torch.from_numpy(deltas.reshape(fh, fw, A * 4)).permute(2, 0, 1).contiguous()
    ref = O.rpn_get_bboxes(cls.numpy(), reg.numpy(), np.array([ih, iw, 3]), cfg)
    scores = torch.from_numpy(O.sigmoid32(logits[None]))
    props, n = ops.rpn_proposals(scores.cuda(), torch.from_numpy(deltas[None]).cuda(), torch.from_numpy(base).cuda(), fh, fw,
                                 16, ih, iw, rp['target_means'], rp['target_stds'], 6000, 0, 0.7, 300)
    n = int(n.item())
    assert n == len(ref) == 300
    assert np.array_equal(props[0, :n].cpu().numpy(), ref)
    # the chain really alternates: with the 10 px shift every second box of a row survives
    if shift == 10.0:
        kept_x = np.sort(ref[np.abs(ref[:, 1] - 20.0) < 1.0][:, 0])
        assert len(kept_x) == per_row // 2 and np.allclose(np.diff(kept_x), 20.0, atol=0.05)


def test_proposals_batch_of_two_images_equals_single_images():
    from fgn_amd import ops
    from fgn_amd.config import fgn_r50_c4_config
    rp = fgn_r50_c4_config(3, 3)['rpn_head']
    g = torch.Generator().manual_seed(21)
    fh, fw = 50, 84
    scores = torch.sigmoid(torch.randn(2, fh * fw * 15, generator=g) * 3).cuda()
    deltas = (torch.randn(2, fh * fw * 15, 4, generator=g) * 0.5).cuda()
    anchors = torch.from_numpy(ops.base_anchors(rp['anchor_scales'], rp['anchor_ratios'], 16)).cuda()
    run = lambda s, d: ops.rpn_proposals(s, d, anchors, fh, fw, 16, 800, 1333, rp['target_means'], rp['target_stds'],
                                         6000, 0, 0.7, 300, with_rois=True)
    props, n, rois = run(scores, deltas)
    for i in range(2):
        p1, n1, r1 = run(scores[i:i + 1].contiguous(), deltas[i:i + 1].contiguous())
        assert int(n1[0]) == int(n[i]) and torch.equal(p1[0], props[i])
        assert torch.equal(r1[:, 1:], rois[i * 300:(i + 1) * 300, 1:]) and float(rois[i * 300, 0]) == i


def test_proposals_bit_exact_odd_caps():
    _proposal_case(3, 20, 31, 1000, 77, ties=True)
    _proposal_case(4, 64, 64, 6000, 1000, ties=False)            # cfg5-like: 1000 proposals


def test_sigmoid_merge_is_correctly_rounded():
    from fgn_amd import ops
    from oracle import fgn_ref_cpu as O
    g = torch.Generator().manual_seed(9)
    head = torch.randn(3, 4, 5, 75, generator=g) * 6
    logits, scores, deltas = ops.rpn_merge(head.cuda(), 1, 3, 15)
    cls = head[..., :15].reshape(3, -1)
    best = cls.max(0).values
    assert torch.equal(logits.cpu()[0], best)
    assert np.array_equal(scores.cpu().numpy()[0], O.sigmoid32(best.numpy()))
    am = cls.argmax(0)
    ref_d = head[..., 15:].reshape(3, -1, 4)[am, torch.arange(am.numel())]
    assert torch.equal(deltas.cpu()[0], ref_d)


# ---------------------------------------------------------------- detections: bit-exact
@pytest.mark.parametrize('n_ways,r,seed', [(3, 300, 0), (3, 37, 1), (1, 120, 2), (5, 1000, 3)])
def test_det_post_bit_exact(n_ways, r, seed):
    from fgn_amd import ops
    from fgn_amd.config import fgn_r50_c4_config
    from oracle import fgn_ref_cpu as O
    cfg = fgn_r50_c4_config(n_ways, 1)
    g = torch.Generator().manual_seed(seed)
    ih, iw = 800, 1333
    x1 = torch.rand(r, generator=g) * iw * 0.8
    y1 = torch.rand(r, generator=g) * ih * 0.8
    rois = torch.stack([torch.zeros(r), x1, y1, x1 + torch.rand(r, generator=g) * 300 + 1,
                        y1 + torch.rand(r, generator=g) * 200 + 1], 1)
    rois[: r // 3, 1:] = rois[r // 3: 2 * (r // 3), 1:] + 3.0      # overlapping boxes -> NMS does work
    cls_raw = torch.randn(r * n_ways, 2, generator=g) * 2
    reg_raw = torch.randn(r * n_ways, 4, generator=g)
    cls_score, bbox_pred = O.count_modified_cls_bbox(r, cls_raw, reg_raw, n_ways)
    ref_b, ref_l = O.bbox_get_bboxes(rois.numpy(), cls_score.numpy(), bbox_pred.numpy(), np.array([ih, iw, 3]), cfg)
    bh = cfg['roi_head']['bbox_head']
    det, lab, n, dbg = ops.det_post(rois.cuda(), cls_raw.cuda(), reg_raw.cuda(), n_ways, ih, iw, bh['target_means'],
                                    bh['target_stds'], 0.05, 0.5, 100, debug_scores=True)
    n = int(n.item())
    assert np.array_equal(dbg.cpu().numpy(), O.softmax32(cls_score.numpy()))       # correctly rounded softmax
    assert n == len(ref_b)
    assert np.array_equal(det[:n].cpu().numpy(), ref_b)                               # boxes + scores bitwise
    assert np.array_equal(lab[:n].cpu().numpy(), ref_l)
    # device-side RoI count
    cnt = torch.tensor([r // 2], dtype=torch.int32, device='cuda')
    det2, lab2, n2 = ops.det_post(rois.cuda(), cls_raw.cuda(), reg_raw.cuda(), n_ways, ih, iw, bh['target_means'],
                                  bh['target_stds'], 0.05, 0.5, 100, n_rois_dev=cnt)
    h = r // 2
    cs2, bp2 = O.count_modified_cls_bbox(h, cls_raw[:h * n_ways], reg_raw[:h * n_ways], n_ways)
    rb2, rl2 = O.bbox_get_bboxes(rois[:h].numpy(), cs2.numpy(), bp2.numpy(), np.array([ih, iw, 3]), cfg)
    assert np.array_equal(det2[:int(n2.item())].cpu().numpy(), rb2)


def test_det_post_without_a_candidate_over_the_score_threshold():
    from fgn_amd import ops
    r, n_ways = 50, 3
    rois = torch.tensor([[0, 10.0 + i, 20.0, 200.0 + i, 180.0] for i in range(r)]).cuda()
    cls_raw = torch.zeros(r * n_ways, 2)
    cls_raw[:, 0] = 12.0                               # background logit: every class score ~ 6e-6 < 0.05
    det, lab, n = ops.det_post(rois, cls_raw.cuda(), torch.zeros(r * n_ways, 4).cuda(), n_ways, 400, 400, (0, 0, 0, 0),
                               (.1, .1, .2, .2), 0.05, 0.5, 100)
    assert int(n[0]) == 0 and float(det.abs().sum()) == 0.0 and int(lab.abs().sum()) == 0


def test_det_post_batch_of_images_in_one_launch_equals_one_launch_per_image():
    from fgn_amd import ops
    g = torch.Generator().manual_seed(5)
    B, r, n_ways, ih, iw = 3, 200, 3, 800, 1333
    x1 = torch.rand(B * r, generator=g) * iw * 0.8
    y1 = torch.rand(B * r, generator=g) * ih * 0.8
    rois = torch.stack([torch.arange(B * r) // r, x1, y1, x1 + torch.rand(B * r, generator=g) * 300 + 1,
                        y1 + torch.rand(B * r, generator=g) * 200 + 1], 1).float().cuda()
    cls_raw = (torch.randn(B * r * n_ways, 2, generator=g) * 2).cuda()
    reg_raw = torch.randn(B * r * n_ways, 4, generator=g).cuda()
    cnt = torch.tensor([r, r // 2, 7], dtype=torch.int32, device='cuda')
    args = (n_ways, ih, iw, (0, 0, 0, 0), (.1, .1, .2, .2), 0.05, 0.5, 100)
    det, lab, n, mr = ops.det_post(rois, cls_raw, reg_raw, *args, n_rois_dev=cnt, img_index=0, batch=B)
    for i in range(B):
        d1, l1, n1, m1 = ops.det_post(rois[i * r:(i + 1) * r], cls_raw[i * r * n_ways:(i + 1) * r * n_ways],
                                      reg_raw[i * r * n_ways:(i + 1) * r * n_ways], *args, n_rois_dev=cnt[i:i + 1],
                                      img_index=i)
        assert int(n1[0]) == int(n[i]) and int(n1[0]) > 0
        assert torch.equal(d1, det[i * 100:(i + 1) * 100]) and torch.equal(l1, lab[i * 100:(i + 1) * 100])
        assert torch.equal(m1, mr[i * 100:(i + 1) * 100])


def test_det_post_matches_reference_golden_cls_mod(golden_dir):
    """count_modified_cls_bbox column logic vs the reference's own output."""
    from fgn_amd import ops
    from oracle import fgn_ref_cpu as O
    z = np.load(os.path.join(golden_dir, 'cls_bbox.npz'))
    rois = torch.tensor([[0, 10. * i, 5. * i, 10. * i + 50, 5. * i + 40] for i in range(7)])
    det, lab, n, dbg = ops.det_post(rois.cuda(), torch.from_numpy(z['cls_raw']).cuda(),
                                    torch.from_numpy(z['reg_raw']).cuda(), 3, 400, 400, (0, 0, 0, 0),
                                    (.1, .1, .2, .2), 0.05, 0.5, 100, debug_scores=True)
    assert np.array_equal(dbg.cpu().numpy(), O.softmax32(z['cls_n3']))


@pytest.mark.gpu
def test_phase_signal_and_wait_between_streams():
    """The counter marks of the pipelined serving loop: a wait passes once the counter has reached its target, gives up
    after its timeout when it never does, and a signal captured into a hipGraph bumps the counter on every replay."""
    import time
    from fgn_amd import ops
    c = torch.zeros(1, dtype=torch.int32, device='cuda')
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(b):
        ops.phase_wait(c, 2, timeout_us=500000)              # queued first: sleeps until two signals have arrived
        done = torch.ones(1, device='cuda')
    with torch.cuda.stream(a):
        ops.phase_signal(c)
        ops.phase_signal(c)
    torch.cuda.synchronize()
    assert int(c) == 2 and float(done) == 1.0
    t = time.perf_counter()
    ops.phase_wait(c, 5, timeout_us=3000)                    # never reached: returns after ~3 ms
    torch.cuda.synchronize()
    assert 0.002 < time.perf_counter() - t < 0.2
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(a):
        with torch.cuda.graph(g, stream=a):
            ops.phase_signal(c)
        for _ in range(3):
            g.replay()
    torch.cuda.synchronize()
    assert int(c) == 5
