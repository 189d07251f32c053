"""End-to-end: FGN.simple_test on the HIP path vs the oracle on the same seeded episodes."""
import numpy as np
import pytest
import torch

from fgn_amd.agreement import episode_maxima, match_detections

pytestmark = pytest.mark.gpu

TOL = 1e-4      # BASELINE.json north_star: "masks/scores within 1e-4 fp32"


# Boxes: north_star asks for bit-exact boxes / labels.  Stage-wise they are (tests/test_hip_stages.py: identical head
# outputs in -> identical boxes, scores, labels out, bit for bit); END TO END the two paths accumulate in different orders
# (MFMA tiles vs MKL), so a decoded box moves by ~1e-3 px (measured maximum over 6400 detections 1.3e-3 px,
# profiles/r04_accuracy_64.json).  The asserted bound is 3x that maximum; the matching radius of fgn_amd.agreement stays
# 1e-2 px, so a pair that drifted past the bound fails as a box difference, not as a "flip".
BOX_TOL = 4e-3
CFG5_PROP_FLIPS = 8         # rows of the proposal symmetric difference tolerated at cfg5 (each with its printed reason)
# final binary masks (decoded dt_isegmaps_rle) of matched pairs: IoU >= 0.999, or - a mask of a few hundred pixels loses
# more than that to ONE pixel whose probability sits on the 0.5 threshold - at most MASK_DIFF_PX differing pixels
# (tests/test_hip_mask.py checks pixel by pixel that a differing pixel has the oracle's own sample within 1e-5 of the
# threshold).  Measured r05 over cfg1-cfg5: 98-100 of 100 RLE strings byte-identical per image, worst IoU 0.99899 (1 pixel).
MASK_IOU_MIN = 0.999
MASK_DIFF_PX = 3


def _check_tolerance(ref, got, tr_ref, tr, name, max_flips=0, cfg=None, max_prop_flips=0):
    """Per image: (1) the PROPOSAL sets of the two paths (fgn.py:229-235) compared as sets - symmetric difference
    printed with the reason each row can differ (fgn_amd.agreement.proposal_set_difference), ``max_prop_flips`` rows
    tolerated (0 unless the caller says why), none of them unexplained; (2) HIP detections matched to the oracle's
    (same label, box within 1e-2 px): |d score| <= 1e-4, max |d mask probability| <= 1e-4 and |d box| <= 4e-3 px on
    EVERY matched pair, detections without a partner counted as selection flips - ZERO asserted (every seeded episode
    of this file produced zero in rounds 1-4; a test that can genuinely flip passes its own ``max_flips`` with a
    comment saying why); (3) the FINAL binary masks of every matched pair (decoded ``dt_isegmaps_rle``,
    fgn_roi_head.py:668-671): IoU >= 0.999."""
    from fgn_amd.agreement import mask_iou_of_pairs, match_detections, proposal_set_difference
    start = 0
    out = []
    for i in range(len(ref)):
        if cfg is not None and 'proposals' in tr_ref and 'proposals' in tr:
            rp = cfg['test_cfg']['rpn']
            n_hip = int(tr['n_props'][i])
            ps = proposal_set_difference(tr_ref['proposals'][i], tr['proposals'][i, :n_hip].cpu().numpy(),
                                         rp['nms_iou_threshold'], rp['max_per_img'])
            print(f'[parity {name} img {i}] proposals ref/hip {ps["n_ref"]}/{ps["n_got"]}, matched {ps["matched"]}, '
                  f'only ref/hip {len(ps["only_ref"])}/{len(ps["only_got"])}; on matched: max|d box| {ps["max_dbox"]:.2e} px, '
                  f'max|d score| {ps["max_dscore"]:.2e}')
            for side in ('only_ref', 'only_got'):
                for row in ps[side]:
                    print(f'    {side} row {row["row"]} score {row["score"]:.7f} box {np.round(row["box"], 3).tolist()} '
                          f'reason {row["reason"]} iou margin {row["iou_margin"]}')
            assert ps['unexplained'] == 0, (name, i, ps)
            assert len(ps['only_ref']) <= max_prop_flips and len(ps['only_got']) <= max_prop_flips, (name, i, ps)
            assert ps['max_dscore'] <= TOL and ps['max_dbox'] <= BOX_TOL, (name, i, ps['max_dscore'], ps['max_dbox'])
        n_ref = len(ref[i]['dt_scores'])
        pi = tr['per_image'][i]
        n_got = int(pi['n_det'][0])
        assert n_got == len(got[i]['dt_scores'])
        if n_ref == 0 or n_got == 0:
            assert n_ref + n_got <= max_flips, (name, i, n_ref, n_got)
            start += n_ref
            continue
        m = episode_maxima(ref[i], got[i], tr_ref['mask_prob'][start:start + n_ref].numpy(),
                           pi['mask_prob'][:n_got].cpu().numpy(),
                           tr_ref['mask_logits'][start:start + n_ref].numpy(), pi['mask_logits'][:n_got].cpu().numpy())
        start += n_ref
        pairs, _, _ = match_detections(ref[i]['dt_bboxes'], ref[i]['dt_cat_ids'], got[i]['dt_bboxes'], got[i]['dt_cat_ids'])
        ious, ndiff = mask_iou_of_pairs(ref[i]['dt_isegmaps_rle'], got[i]['dt_isegmaps_rle'], pairs)
        m['min_mask_iou'] = float(ious.min()) if len(ious) else 1.0
        m['max_mask_diff_px'] = int(ndiff.max()) if len(ndiff) else 0
        m['masks_outside_bound'] = int(((ious < MASK_IOU_MIN) & (ndiff > MASK_DIFF_PX)).sum())
        m['masks_identical'] = int(sum(ref[i]['dt_isegmaps_rle'][a] == got[i]['dt_isegmaps_rle'][b] for a, b in pairs))
        print(f'[parity {name} img {i}] detections ref/hip {m["n_ref"]}/{m["n_got"]}, matched {m["matched"]}, '
              f'selection flips ref/hip {m["flips_ref"]}/{m["flips_got"]}; on matched pairs: max|d score| '
              f'{m["max_dscore"]:.2e}, max|d mask prob| {m["max_dprob"]:.2e}, max|d box| {m["max_dbox"]:.2e} px, '
              f'max|d mask logit| {m["max_dlogit"]:.2e} (|logit| <= {m.get("max_abs_logit", 0):.1f}); final masks: '
              f'min IoU {m["min_mask_iou"]:.6f}, at most {m["max_mask_diff_px"]} differing pixels, '
              f'{m["masks_identical"]}/{m["matched"]} RLE strings byte-identical')
        assert m['matched'] > 0
        assert m['max_dscore'] <= TOL, (name, i, m)
        assert m['max_dprob'] <= TOL, (name, i, m)
        assert m['max_dbox'] <= BOX_TOL, (name, i, m)
        assert m['masks_outside_bound'] == 0, (name, i, m)
        assert m['flips_ref'] <= max_flips and m['flips_got'] <= max_flips, (name, i, m)
        out.append(m)
    return out


def _iou(a, b):
    x1 = np.maximum(a[:, None, 0], b[None, :, 0]); y1 = np.maximum(a[:, None, 1], b[None, :, 1])
    x2 = np.minimum(a[:, None, 2], b[None, :, 2]); y2 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1]); ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / (aa[:, None] + ab[None, :] - inter + 1e-12)


def _run(cfg, batch):
    from fgn_amd.detector import FGN
    from fgn_amd.weights import init_state_dict
    from oracle import fgn_ref_cpu as O
    sd = init_state_dict(cfg, 0)
    tr_ref = {}
    ref = O.simple_test(sd, cfg, **batch, trace=tr_ref)
    model = FGN(cfg['n_ways'], cfg['k_shots'], backbone=cfg['backbone'], rpn_head=cfg['rpn_head'],
                roi_head=cfg['roi_head'], test_cfg=cfg['test_cfg'], state_dict=sd)
    model.debug_trace = {}
    got = model.simple_test(**batch, rescale=True)
    return ref, tr_ref, got, model.debug_trace


def _nchw(t):
    return t.permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize('n_ways,k_shots,hw', [(3, 2, (160, 224)), (1, 1, (128, 128))])
def test_e2e_half_width(n_ways, k_shots, hw):
    from fgn_amd.config import tiny_config
    from fgn_amd.episodes import make_batch
    cfg = tiny_config(n_ways, k_shots, width_div=2)
    batch = make_batch(0, 2, n_ways, k_shots, hw[0], hw[1], 64)
    ref, tr_ref, got, tr = _run(cfg, batch)
    # feature maps: fp32 accumulation-order tolerance
    for name in ('qry_fmap', 'spp_fmaps'):
        r = tr_ref[name]
        d = (_nchw(tr[name]) - r).abs().max().item()
        assert d <= 1e-4 * r.abs().max().item(), (name, d)
    d = (tr['spp_cat_mean_mp'].cpu().reshape(-1) - tr_ref['spp_cat_mean_mp'].reshape(-1)).abs().max().item()
    assert d <= 1e-4 * tr_ref['spp_cat_mean_mp'].abs().max().item()
    # detections: every matched pair within north_star's tolerance; flips counted
    _check_tolerance(ref, got, tr_ref, tr, f'half-width N{n_ways}K{k_shots}', cfg=cfg)
    for i in range(2):
        rb, gb = ref[i]['dt_bboxes'], got[i]['dt_bboxes']
        assert got[i]['dt_bboxes'].dtype == np.float32 and got[i]['dt_cat_ids'].dtype == np.int64
        if len(rb) == 0:
            continue
        # passthrough keys
        for key in ('idx', 'qry_bboxes', 'qry_cat_ids', 'qry_img_shape', 'spp_insts_ids'):
            assert np.array_equal(np.asarray(ref[i][key]), np.asarray(got[i][key])), key
        assert len(got[i]['dt_isegmaps_rle']) == len(gb)
        assert got[i]['qry_isegmaps_rle'] == ref[i]['qry_isegmaps_rle']
    # accuracy criterion of BASELINE.json: box / mask AP50 (FSISEGEval protocol) within 0.1 of the CPU path
    from fgn_amd.fsiseg_eval import evaluate_results
    ap_ref, ap_got = evaluate_results(ref, n_ways), evaluate_results(got, n_ways)
    for k in ap_ref:
        assert abs(ap_ref[k] - ap_got[k]) <= 0.1, (k, ap_ref[k], ap_got[k])


def test_empty_detections_and_empty_proposals():
    """Edge cases of the reference (fgn_roi_head.py:558-566, 630-632): no proposal / no detection
    in an image -> empty results, same as the oracle."""
    import copy
    from fgn_amd.config import tiny_config
    from fgn_amd.episodes import make_batch
    base = tiny_config(3, 1, width_div=2)
    batch = make_batch(5, 1, 3, 1, 128, 160, 64)
    cfg = copy.deepcopy(base)
    cfg['test_cfg']['rcnn']['score_thr'] = 2.0                 # softmax scores never exceed 1
    ref, _, got, _ = _run(cfg, batch)
    assert len(ref[0]['dt_scores']) == 0 and len(got[0]['dt_scores']) == 0
    assert got[0]['dt_bboxes'].shape == (0, 4) and got[0]['dt_isegmaps_rle'] == []
    cfg = copy.deepcopy(base)
    cfg['test_cfg']['rpn']['min_bbox_size'] = 1e6              # every proposal is filtered out
    ref, tr_ref, got, tr = _run(cfg, batch)
    assert len(tr_ref['proposals'][0]) == 0 and int(tr['n_props'][0]) == 0
    assert len(ref[0]['dt_scores']) == 0 and len(got[0]['dt_scores']) == 0


def test_e2e_five_way_build_extension():
    """N=5 (cfg5 of BASELINE.json) has no reference semantics (the reference asserts N in {1,3},
    fgn_roi_head.py:303-308); oracle and HIP path share the natural generalisation (columns 1::2)."""
    from fgn_amd.config import tiny_config, with_caps
    from fgn_amd.episodes import make_batch
    cfg = with_caps(tiny_config(5, 2, width_div=2), rpn_max=1000)
    batch = make_batch(3, 1, 5, 2, 160, 160, 64)
    ref, tr_ref, got, tr = _run(cfg, batch)
    _check_tolerance(ref, got, tr_ref, tr, 'five-way half-width R<=1000', cfg=cfg)
    assert set(np.unique(got[0]['dt_cat_ids'])) <= set(range(5))


@pytest.mark.parametrize('name', ['cfg1', 'cfg2'])
def test_e2e_full_width_reference_configs(name):
    """cfg1 (MNISTISEG 1-way 1-shot 128x128) and cfg2 (OMNIISEG 3-way 1-shot 256x256) of BASELINE.json
    with the full ResNet-50-C4 widths."""
    from fgn_amd.config import fgn_r50_c4_config
    from fgn_amd.episodes import CONFIGS, make_batch
    from fgn_amd.fsiseg_eval import evaluate_results
    shape = CONFIGS[name]
    cfg = fgn_r50_c4_config(shape['n_ways'], shape['k_shots'])
    batch = make_batch(11, 1, **shape)
    ref, tr_ref, got, tr = _run(cfg, batch)
    r = tr_ref['qry_fmap']
    assert (_nchw(tr['qry_fmap']) - r).abs().max().item() <= 1e-4 * r.abs().max().item()
    _check_tolerance(ref, got, tr_ref, tr, name, cfg=cfg)
    # HIP detections scored against the CPU path's detections as ground truth
    as_gt = dict(got[0])
    as_gt['qry_bboxes'], as_gt['qry_cat_ids'] = ref[0]['dt_bboxes'], ref[0]['dt_cat_ids']
    as_gt['qry_isegmaps_rle'] = ref[0]['dt_isegmaps_rle']
    agree = evaluate_results([as_gt], shape['n_ways'])
    assert agree['bbox_mAP50'] >= 0.9 and agree['segm_mAP50'] >= 0.9, agree


def test_e2e_resnet18_extension_of_cfg2():
    """cfg2 as BASELINE.json words it: OMNIISEG 3-way 1-shot, 256x256, ResNet-18 backbone.  A build extension (the
    reference has no R18 config and hard-codes 1024-channel heads): BasicBlock stages, 256-channel
    C4, heads at 256 channels, GroupNorm(32, 256) = 8 channels per group in the relation head; oracle and HIP path
    share the definition (config.fgn_r18_c4_config), parity at north_star's tolerance."""
    from fgn_amd.config import fgn_r18_c4_config
    from fgn_amd.episodes import CONFIGS, make_batch
    cfg = fgn_r18_c4_config(3, 1)
    batch = make_batch(11, 2, **CONFIGS['cfg2'])
    ref, tr_ref, got, tr = _run(cfg, batch)
    r = tr_ref['qry_fmap']
    assert tuple(r.shape[1:]) == (256, 16, 16)
    assert (_nchw(tr['qry_fmap']) - r).abs().max().item() <= 1e-4 * r.abs().max().item()
    _check_tolerance(ref, got, tr_ref, tr, 'cfg2 ResNet-18', cfg=cfg)


def test_rccl_gather_single_rank():
    """The collective of the multi-GPU path (one all_gather_into_tensor of padded detection records) on
    the RCCL backend; one rank is all a 1-GPU box can host, the N>1 layout is covered by the gloo test."""
    import os
    import torch.distributed as dist
    from fgn_amd import dist as fd
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29600 + os.getpid() % 300))
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        dets = [dict(det_bboxes=torch.rand(100, 5, device='cuda'), det_labels=torch.randint(0, 3, (100,), device='cuda'),
                     n_dets=torch.tensor([37], dtype=torch.int32, device='cuda'))]
        recs, cnts = fd.pack_detections(dets, 100)
        # world size 1 short-circuits in gather_detections; call the collective directly as it does
        msg = torch.cat([recs.reshape(1, -1), cnts.to(recs.dtype)[:, None]], 1).contiguous()
        out = torch.empty_like(msg)
        dist.all_gather_into_tensor(out, msg)
        torch.cuda.synchronize()
        assert torch.equal(out, msg)
        g_recs, g_cnts = fd.gather_detections(recs, cnts)
        assert g_recs.shape == (1, 1, 100, 6) and int(g_cnts[0, 0]) == 37
    finally:
        dist.destroy_process_group()


def test_graph_capture_with_an_rccl_collective_in_flight():
    """bench.py at N > 1: every caller stream captures its hipGraph during set-up while the all-gather of the previous
    step may still be polled by the process group's watchdog thread.  One rank is all this box hosts: the collective
    is issued on its own stream right before the second capture; results must equal the eager ones."""
    import os
    import torch.distributed as dist
    from fgn_amd.config import tiny_config
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import make_batch
    from fgn_amd.weights import init_state_dict
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', str(29300 + os.getpid() % 300))
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        cfg = tiny_config(3, 2, width_div=2)
        model = FGN(3, 2, backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'],
                    test_cfg=cfg['test_cfg'], state_dict=init_state_dict(cfg, 0))
        batch = make_batch(4, 1, 3, 2, 160, 224, 64)
        ref = model.simple_test(**batch, rescale=True)
        model.use_graphs = True
        msg = torch.rand(1 << 20, device='cuda')
        out = torch.empty_like(msg)
        comm = torch.cuda.Stream()
        got = []
        for st in (torch.cuda.Stream(), torch.cuda.Stream()):          # one capture per caller stream
            with torch.cuda.stream(comm):
                for _ in range(8):
                    dist.all_gather_into_tensor(out, msg)               # in flight / polled while the capture runs
            with torch.cuda.stream(st):
                got.append(model.simple_test(**batch, rescale=True))
        torch.cuda.synchronize()
        assert torch.equal(out, msg) and len(model._graphs) == 2
        for g in got:
            for x, y in zip(g, ref):
                assert len(y['dt_scores']) > 0
                for key in ('dt_scores', 'dt_bboxes', 'dt_cat_ids'):
                    assert np.array_equal(x[key], y[key]), key
                assert x['dt_isegmaps_rle'] == y['dt_isegmaps_rle']
    finally:
        dist.destroy_process_group()


def test_cfg3_full_size_parity_and_invariants():
    """The headline configuration itself (3-way 3-shot, 800x1333, full ResNet-50-C4, R<=300, D<=100):
    parity against the oracle, plus size-independent properties of the output."""
    from fgn_amd import rle
    from fgn_amd.config import fgn_r50_c4_config
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import CONFIGS, make_batch
    from fgn_amd.fsiseg_eval import evaluate_results
    from fgn_amd.weights import init_state_dict
    from oracle import fgn_ref_cpu as O
    shape = CONFIGS['cfg3']
    cfg = fgn_r50_c4_config(3, 3)
    sd = init_state_dict(cfg, 0)
    batch = make_batch(21, 1, **shape)
    model = FGN(3, 3, state_dict=sd)
    model.debug_trace = {}
    got = model.simple_test(**batch, rescale=True)
    tr = model.debug_trace
    model.debug_trace = None
    again = model.simple_test(**batch, rescale=True)
    g = got[0]
    # --- determinism: bit-identical results run to run (split-K sums are order-fixed)
    for key in ('dt_scores', 'dt_bboxes', 'dt_cat_ids'):
        assert np.array_equal(g[key], again[0][key]), key
    assert g['dt_isegmaps_rle'] == again[0]['dt_isegmaps_rle']
    # --- invariants of proposals (device tensors of the trace)
    n_props = int(tr['n_props'][0])
    props = tr['proposals'][0, :n_props].cpu().numpy()
    assert 0 < n_props <= 300 and np.all(np.diff(props[:, 4]) <= 0)                 # score-sorted
    assert props[:, [0, 2]].min() >= 0 and props[:, [0, 2]].max() <= 1333 and props[:, [1, 3]].max() <= 800
    assert np.all(props[:, 2] > props[:, 0]) and np.all(props[:, 3] > props[:, 1])
    iou = _iou(props[:, :4], props[:, :4])
    np.fill_diagonal(iou, 0)
    assert iou.max() <= 0.7 + 1e-6                                                  # NMS post-condition
    # --- invariants of detections
    d = len(g['dt_scores'])
    assert 0 < d <= 100 and np.all(np.diff(g['dt_scores']) <= 0) and g['dt_scores'].min() > 0.05
    assert set(np.unique(g['dt_cat_ids'])) <= {0, 1, 2}
    b = g['dt_bboxes'][:, [1, 0, 3, 2]]
    for c in range(3):                                                              # class-aware NMS at 0.5
        bc = b[g['dt_cat_ids'] == c]
        if len(bc) > 1:
            i2 = _iou(bc, bc)
            np.fill_diagonal(i2, 0)
            assert i2.max() <= 0.5 + 1e-6
    for j in (0, d // 2, d - 1):                                                    # masks live inside their box (+1 px)
        m = rle.decode(g['dt_isegmaps_rle'][j]).astype(bool)
        assert m.shape == (800, 1333)
        ys, xs = np.nonzero(m)
        if len(ys):
            assert xs.min() >= np.floor(b[j, 0]) - 1 and xs.max() <= np.ceil(b[j, 2]) + 1
            assert ys.min() >= np.floor(b[j, 1]) - 1 and ys.max() <= np.ceil(b[j, 3]) + 1
    # --- parity against the oracle on the same episode
    tr_ref = {}
    ref = O.simple_test(sd, cfg, **batch, trace=tr_ref)
    r = tr_ref['qry_fmap']
    assert (_nchw(tr['qry_fmap']) - r).abs().max().item() <= 1e-4 * r.abs().max().item()
    _check_tolerance(ref, got, tr_ref, tr, 'cfg3', cfg=cfg)
    as_gt = dict(g)
    as_gt['qry_bboxes'], as_gt['qry_cat_ids'] = ref[0]['dt_bboxes'], ref[0]['dt_cat_ids']
    as_gt['qry_isegmaps_rle'] = ref[0]['dt_isegmaps_rle']
    agree = evaluate_results([as_gt], 3)
    assert agree['bbox_mAP50'] >= 0.9 and agree['segm_mAP50'] >= 0.9, agree


def test_support_code_cache_is_identical():
    """queries that share a support set reuse ``encode_supports``; the results
    are the same bytes as the per-query recomputation the reference does (fgn.py:212-215) when that recomputation
    runs the two backbone passes as separate launches; against the default (query and support maps through shared
    backbone launches: other split-K plans) every detection agrees within north_star's tolerance."""
    from fgn_amd.config import tiny_config
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import make_batch
    from fgn_amd.weights import init_state_dict
    cfg = tiny_config(3, 2, width_div=2)
    model = FGN(3, 2, backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'],
                test_cfg=cfg['test_cfg'], state_dict=init_state_dict(cfg, 0))
    first = make_batch(0, 1, 3, 2, 160, 224, 64)
    code = model.encode_supports(first['spp_imgs'], first['spp_bboxes'], first['spp_isegmaps'])
    for q in range(3):
        other = make_batch(10 + q, 1, 3, 2, 160, 224, 64)
        batch = dict(other, spp_imgs=first['spp_imgs'], spp_bboxes=first['spp_bboxes'],
                     spp_isegmaps=first['spp_isegmaps'])
        no_spp = {k: v for k, v in batch.items() if not k.startswith('spp_i') and k != 'spp_bboxes'}
        cached = model.simple_test(**no_spp, support_code=code, rescale=True)
        model.use_merged_backbone = model.use_merged_support_head = False      # the separate form: the cache's own launches
        plain = model.simple_test(**batch, rescale=True)
        model.use_merged_backbone = model.use_merged_support_head = True       # the default
        merged = model.simple_test(**batch, rescale=True)
        for a, b, c in zip(plain, cached, merged):
            assert len(a['dt_scores']) > 0
            for key in ('dt_scores', 'dt_bboxes', 'dt_cat_ids'):
                assert np.array_equal(a[key], b[key]), key
            assert a['dt_isegmaps_rle'] == b['dt_isegmaps_rle']
            pairs, only_a, only_c = match_detections(a['dt_bboxes'], a['dt_cat_ids'], c['dt_bboxes'], c['dt_cat_ids'])
            ia, ic = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
            assert len(only_a) <= 2 and len(only_c) <= 2 and np.abs(a['dt_scores'][ia] - c['dt_scores'][ic]).max() <= TOL
    with pytest.raises(ValueError):
        two = make_batch(0, 2, 3, 2, 160, 224, 64)
        model.simple_test(**{k: v for k, v in two.items() if not k.startswith('spp_i') and k != 'spp_bboxes'},
                          support_code=code)


def test_hip_graph_replay_is_identical():
    """``use_graphs``: the captured hipGraph replays the same kernels - identical bytes to the eager
    launch sequence, across different inputs, batch 2, and with a cached support code."""
    from fgn_amd.config import tiny_config
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import make_batch
    from fgn_amd.weights import init_state_dict
    cfg = tiny_config(3, 2, width_div=2)
    model = FGN(3, 2, backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'],
                test_cfg=cfg['test_cfg'], state_dict=init_state_dict(cfg, 0))

    def same(a, b):
        for x, y in zip(a, b):
            assert len(x['dt_scores']) > 0
            for key in ('dt_scores', 'dt_bboxes', 'dt_cat_ids'):
                assert np.array_equal(x[key], y[key]), key
            assert x['dt_isegmaps_rle'] == y['dt_isegmaps_rle']

    batches = [make_batch(4 * q, 2, 3, 2, 160, 224, 64) for q in range(3)]
    eager = [model.simple_test(**b, rescale=True) for b in batches]
    model.use_graphs = True
    for rep in range(2):                       # second round replays an existing graph
        for b, e in zip(batches, eager):
            same(e, model.simple_test(**b, rescale=True))
    assert len(model._graphs) == 1
    # pipelined use: queue two replays before reading the first result (static outputs are reused)
    d0 = model.detect_device(batches[0]['qry_img'], batches[0]['spp_imgs'], batches[0]['spp_bboxes'],
                             batches[0]['spp_isegmaps'], batches[0]['img_shape'])
    d1 = model.detect_device(batches[1]['qry_img'], batches[1]['spp_imgs'], batches[1]['spp_bboxes'],
                             batches[1]['spp_isegmaps'], batches[1]['img_shape'])
    strip = lambda rs: [{k: r[k] for k in ('dt_scores', 'dt_bboxes', 'dt_cat_ids', 'dt_isegmaps_rle')} for r in rs]
    same(eager[0], strip(model.pack_results(d0, 2)))
    same(eager[1], strip(model.pack_results(d1, 2)))
    # another geometry -> another graph; cached support code (byte-identical to the per-query recomputation in its
    # separate-launch form, see test_support_code_cache_is_identical)
    one = make_batch(7, 1, 3, 2, 128, 160, 64)
    model.use_graphs = False
    model.use_merged_backbone = model.use_merged_support_head = False
    code = model.encode_supports(one['spp_imgs'], one['spp_bboxes'], one['spp_isegmaps'])
    ref = model.simple_test(**one, rescale=True)
    model.use_graphs = True
    q_only = {k: v for k, v in one.items() if not k.startswith('spp_i') and k != 'spp_bboxes'}
    same(ref, model.simple_test(**q_only, support_code=code, rescale=True))
    same(ref, model.simple_test(**q_only, support_code=code, rescale=True))
    same(ref, model.simple_test(**one, rescale=True))
    assert len(model._graphs) == 3


@pytest.mark.parametrize('in_flight,xfer_mode,lock', [(2, 0, ''), (3, 0, ''), (5, 0, ''), (3, 3, ''), (5, 3, ''), (3, 2, ''), (3, 1, ''),
                                                    (3, 3, 'rpn'), (3, 3, 'layer2'), (2, 0, 'proposals')])
def test_two_caller_streams_with_two_episodes_in_flight_are_identical_to_serial_eager(in_flight, xfer_mode, lock):
    """bench.py's execution mode (round 3): hipGraph replay, steps alternating between two caller streams, two to five
    episodes queued before the first is packed (three is bench.py's default: a caller stream then holds two replays of
    its graph, the second waiting on the GPU for the download of the first) - one captured graph, one set of static
    buffers, one side / upload / copy stream and one ring of pinned result slots per caller stream.  Twelve steps over
    five distinct episodes must give, step by step, the bytes of a serial eager run (boxes, scores, labels, detection
    RLE, ground-truth RLE).  ``xfer_mode``: the transfer arrangements of ``FGN.transfer_stream`` - 3 (bench.py's choice
    at one episode per step, round 4): uploads and result copies ride on the caller stream itself, no upload / copy
    streams; 2: copies on the caller stream, one upload stream for all; 1: one stream for both.  ``lock``: the phase
    lock of INTEGRATION.md's serving loop (a counter mark inside every captured episode, the other stream waits for
    it before its next episode): same bytes, and every episode has sent exactly one mark."""
    from fgn_amd.config import tiny_config
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import make_batch
    from fgn_amd.weights import init_state_dict
    cfg = tiny_config(3, 2, width_div=2)
    model = FGN(3, 2, backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'],
                test_cfg=cfg['test_cfg'], state_dict=init_state_dict(cfg, 0))
    eps = [make_batch(11 * q, 1, 3, 2, 160, 224, 64) for q in range(5)]
    eps = [{k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in e.items()} for e in eps]
    want = [model.simple_test(**e, rescale=True) for e in eps]
    assert all(len(w[0]['dt_scores']) > 0 for w in want)
    model.use_graphs = True
    if xfer_mode:
        model.transfer_stream(xfer_mode)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    pending, got = [], []
    from fgn_amd import ops
    marks = torch.zeros(2, dtype=torch.int32, device='cuda')
    sent = [0, 0]
    if lock:
        model.phase_point = lock

    def finish(item):
        e, dets = item
        return model.pack_results(dets, 1, qry_isegmaps=e['qry_isegmaps'], img_shape=e['img_shape'])
    for i in range(12):
        e = eps[i % 5]
        k = i % 2
        with torch.cuda.stream(streams[k]):
            if lock:
                if sent[1 - k]:
                    ops.phase_wait(marks[1 - k:2 - k], sent[1 - k])
                sent[k] += 1
            dets = model.detect_device(e['qry_img'], e['spp_imgs'], e['spp_bboxes'], e['spp_isegmaps'], e['img_shape'],
                                       qry_isegmaps=e['qry_isegmaps'], phase_counter=marks[k:k + 1] if lock else None)
        pending.append((e, dets))
        if len(pending) > in_flight:
            got.append(finish(pending.pop(0)))
    while pending:
        got.append(finish(pending.pop(0)))
    assert len(model._graphs) == 2                                 # one graph per caller stream
    torch.cuda.synchronize()
    assert marks.tolist() == (sent if lock else [0, 0])             # one mark per episode, captured graphs included
    for i, g in enumerate(got):
        w = want[i % 5]
        for key in ('dt_scores', 'dt_bboxes', 'dt_cat_ids'):
            assert np.array_equal(w[0][key], g[0][key]), (i, key)
        assert w[0]['dt_isegmaps_rle'] == g[0]['dt_isegmaps_rle'] and w[0]['qry_isegmaps_rle'] == g[0]['qry_isegmaps_rle'], i
    assert all(not s_['busy'] for ring in model._pinned.values() for s_ in ring)


def test_merged_support_head_option_matches_the_default():
    """``use_merged_support_head`` (support RoIs in the box head's RoI batch): same detections as the default within
    the tolerance of the metric, with and without the RoIAlign-commuted first conv, batch 1 (device-side proposal count
    shifted by the support rows) and batch 2."""
    from fgn_amd.config import tiny_config
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import make_batch
    from fgn_amd.weights import init_state_dict
    cfg = tiny_config(3, 2, width_div=2)
    sd = init_state_dict(cfg, 0)
    for nb, commute in ((1, True), (2, True), (1, False)):
        batch = make_batch(5, nb, 3, 2, 160, 224, 64)
        res = []
        for msh in (False, True):
            model = FGN(3, 2, backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'],
                        test_cfg=cfg['test_cfg'], state_dict=sd)
            model.use_roi_commute = commute
            model.use_merged_support_head = msh
            res.append(model.simple_test(**batch, rescale=True))
        for a, c in zip(*res):
            assert len(a['dt_scores']) > 0
            pairs, only_a, only_c = match_detections(a['dt_bboxes'], a['dt_cat_ids'], c['dt_bboxes'], c['dt_cat_ids'])
            ia, ic = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
            assert len(only_a) <= 2 and len(only_c) <= 2 and np.abs(a['dt_scores'][ia] - c['dt_scores'][ic]).max() <= TOL


def test_winograd_and_direct_paths_agree():
    """`use_winograd=False`, `use_roi_commute=False` (direct form for every 3x3, shared_head conv1 on the RoIs: the
    reference's formulation op for op) and the default path give the same detections within the tolerance of
    the metric."""
    from fgn_amd.config import tiny_config
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import make_batch
    from fgn_amd.weights import init_state_dict
    cfg = tiny_config(3, 2, width_div=2)
    sd = init_state_dict(cfg, 0)
    batch = make_batch(3, 2, 3, 2, 160, 224, 64)
    out = {}
    for wg in (True, 2, False):
        model = FGN(3, 2, backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'],
                    test_cfg=cfg['test_cfg'], state_dict=sd)
        model.use_winograd = wg
        model.use_roi_commute = bool(wg)    # False: shared_head conv1 on every RoI, as the reference formulates it
        model.debug_trace = {}
        out[wg] = (model.simple_test(**batch, rescale=True), model.debug_trace)
        assert (model._P['rpn_conv_wg'] is not None) == bool(wg) and (model._P['sh0_lin'] is not None) == bool(wg)
        if wg:
            assert model._P['rpn_conv_wg'].m == (4 if wg is True else 2)
    (b, tb) = out[False]
    ref = tb['rpn_logits']
    for wg in (True, 2):
      a, ta = out[wg]
      assert (ta['rpn_logits'] - ref).abs().max().item() <= 1e-4 * max(ref.abs().max().item(), 1.0)
      for x, y in zip(a, b):
        pairs, mx, my = match_detections(x['dt_bboxes'], x['dt_cat_ids'], y['dt_bboxes'], y['dt_cat_ids'])
        assert len(pairs) > 0 and len(mx) <= 2 and len(my) <= 2
        ia, ib = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
        assert np.abs(x['dt_scores'][ia] - y['dt_scores'][ib]).max() <= TOL


def test_cfg4_batched_and_cfg5_full_size_invariants():
    """The two multi-GPU configurations of BASELINE.json at their full sizes (per-GPU share):
    cfg4 = 3-way 3-shot episodes batched at 800x1328 (the reference's x16 rounding, base_fst.py:693-694): a
    batch of 8 (the per-GPU share) gives each episode the result it gets alone; cfg5 = 5-way 5-shot, 1024x1024, 1000 proposals:
    size-independent properties of the output (no oracle run: 2.9 TFLOP per episode on the CPU)."""
    from fgn_amd import rle
    from fgn_amd.config import fgn_r50_c4_config, with_caps
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import CONFIGS, RPN_MAX_PER_IMG, make_batch
    from fgn_amd.weights import init_state_dict
    # ---- cfg4: ONE batch of 8 (the per-GPU share) at 800x1328 against the ORACLE (8 x 1.1 TFLOP on the host cores, once;
    # the reference's own evaluation call shape is the same code path at batch 4, fgn_test.py:49,108) - proposal sets,
    # every matched pair, final masks - and against each episode run alone
    cfg = fgn_r50_c4_config(3, 3)
    EPB = 8                                      # BASELINE.json cfg4: 8 episodes per GPU per step
    both = make_batch(40, EPB, **CONFIGS['cfg4'])
    ref, tr_ref, got2, tr = _run(cfg, both)
    assert len(ref) == len(got2) == EPB
    _check_tolerance(ref, got2, tr_ref, tr, f'cfg4 batch {EPB}', cfg=cfg)
    del ref, tr_ref, tr
    model = FGN(3, 3, state_dict=init_state_dict(cfg, 0))
    untraced = model.simple_test(**both, rescale=True)
    for i in range(EPB):
        for key in ('dt_scores', 'dt_bboxes', 'dt_cat_ids'):
            assert np.array_equal(untraced[i][key], got2[i][key]), (i, key)     # the traced run is the product's bytes
        assert untraced[i]['dt_isegmaps_rle'] == got2[i]['dt_isegmaps_rle']
        one = make_batch(40 + i, 1, **CONFIGS['cfg4'])
        got1 = model.simple_test(**one, rescale=True)[0]
        a, b = got2[i], got1
        assert a['qry_img_shape'].tolist() == [800, 1328, 3] and int(a['idx']) == 40 + i
        assert len(b['dt_scores']) > 0
        pairs, ma, mb = match_detections(a['dt_bboxes'], a['dt_cat_ids'], b['dt_bboxes'], b['dt_cat_ids'])
        ia, ib = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])
        ds = np.abs(a['dt_scores'][ia] - b['dt_scores'][ib]).max()
        print(f'[cfg4 batch {EPB} vs alone, episode {i}] matched {len(pairs)}/{len(b["dt_scores"])}, '
              f'flips {len(ma)}/{len(mb)}, max|d score| {ds:.2e}')
        assert ds <= TOL and len(ma) == 0 and len(mb) == 0      # tile partition differs with the batch: fp32 order; 0 flips observed in every round
    del model
    # ---- cfg5
    shape = CONFIGS['cfg5']
    cfg5 = with_caps(fgn_r50_c4_config(5, 5), rpn_max=RPN_MAX_PER_IMG['cfg5'])
    m5 = FGN(5, 5, test_cfg=cfg5['test_cfg'], state_dict=init_state_dict(cfg5, 0))
    b5 = make_batch(3, 1, **shape)
    m5.debug_trace = {}
    g = m5.simple_test(**b5, rescale=True)[0]
    tr = m5.debug_trace
    m5.debug_trace = None
    again = m5.simple_test(**b5, rescale=True)[0]
    for key in ('dt_scores', 'dt_bboxes', 'dt_cat_ids'):
        assert np.array_equal(g[key], again[key]), key                     # bit-reproducible
    assert g['dt_isegmaps_rle'] == again['dt_isegmaps_rle']
    n_props = int(tr['n_props'][0])
    props = tr['proposals'][0, :n_props].cpu().numpy()
    assert 300 < n_props <= 1000 and np.all(np.diff(props[:, 4]) <= 0)
    assert props[:, :4].min() >= 0 and props[:, [0, 2]].max() <= 1024 and props[:, [1, 3]].max() <= 1024
    iou = _iou(props[:, :4], props[:, :4])
    np.fill_diagonal(iou, 0)
    assert iou.max() <= 0.7 + 1e-6
    d = len(g['dt_scores'])
    assert 0 < d <= 100 and np.all(np.diff(g['dt_scores']) <= 0) and g['dt_scores'].min() > 0.05
    assert set(np.unique(g['dt_cat_ids'])) <= set(range(5))
    bx = g['dt_bboxes'][:, [1, 0, 3, 2]]
    for c in range(5):
        bc = bx[g['dt_cat_ids'] == c]
        if len(bc) > 1:
            i2 = _iou(bc, bc)
            np.fill_diagonal(i2, 0)
            assert i2.max() <= 0.5 + 1e-6
    m = rle.decode(g['dt_isegmaps_rle'][0])
    assert m.shape == (1024, 1024)
    del m5
    # ---- cfg5 against the oracle at FULL size (5-way 5-shot, 1024x1024, 25 supports, R <= 1000 proposals: 2.9 TFLOP
    # on the host cores): every matched detection pair within north_star's tolerance, flips counted
    ref, tr_ref, got, tr = _run(cfg5, b5)
    n_ref, n_hip = len(tr_ref['proposals'][0]), int(tr['n_props'][0])
    print(f'[cfg5 full size] proposals ref/hip {n_ref}/{n_hip}, detections {len(ref[0]["dt_scores"])}/{len(got[0]["dt_scores"])}')
    assert n_ref > 300
    # ~1000 proposals out of 61 440 anchors with random weights: a few suppression decisions sit on the NMS threshold
    # (IoU within 1e-4 of 0.7 in one path's arithmetic) - every row of the symmetric difference is printed WITH that
    # reason and none may be unexplained; the final detections still agree with zero flips
    _check_tolerance(ref, got, tr_ref, tr, 'cfg5 full size', cfg=cfg5, max_prop_flips=CFG5_PROP_FLIPS)
    for key in ('dt_scores', 'dt_bboxes', 'dt_cat_ids'):
        assert np.array_equal(g[key], got[0][key]), key                    # same bytes as the un-traced runs above


def test_gathered_records_rebuild_the_result_dicts_and_slots_are_not_overwritten():
    """(1) What the RCCL gather carries (boxes, scores, labels, 14x14 mask probabilities) is enough for any rank to
    emit the complete result dict of an episode: ``results_from_gathered`` on the packed records == ``pack_results``
    of the producing rank, RLE strings byte for byte.  (2) Eight batches queued before the first is packed (more
    than the six host slots round 1 cycled through blindly): every batch still gets its own results."""
    from fgn_amd import dist as fd
    from fgn_amd.config import tiny_config
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import make_batch
    from fgn_amd.weights import init_state_dict
    cfg = tiny_config(3, 2, width_div=2)
    model = FGN(3, 2, backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'],
                test_cfg=cfg['test_cfg'], state_dict=init_state_dict(cfg, 0))
    batches = [make_batch(3 * q, 2, 3, 2, 160, 224, 64) for q in range(8)]
    serial = [model.simple_test(**b, rescale=True) for b in batches]
    queued = [model.detect_device(b['qry_img'], b['spp_imgs'], b['spp_bboxes'], b['spp_isegmaps'], b['img_shape'],
                                  qry_isegmaps=b['qry_isegmaps']) for b in batches]
    assert len({id(d[0]['host']) for d in queued}) == 8                 # eight distinct pinned slots in flight
    for b, dets, want in zip(batches, queued, serial):
        recs, cnts = fd.pack_detections(dets, 100)
        rebuilt = fd.results_from_gathered(recs, cnts, (160, 224), cfg['test_cfg']['rcnn']['mask_thr_binary'])
        got = model.pack_results(dets, 2, qry_isegmaps=b['qry_isegmaps'], img_shape=b['img_shape'])
        for w, g, r in zip(want, got, rebuilt):
            assert len(w['dt_scores']) > 0
            for key in ('dt_scores', 'dt_bboxes', 'dt_cat_ids'):
                assert np.array_equal(w[key], g[key]) and np.array_equal(w[key], r[key]), key
            assert w['dt_isegmaps_rle'] == g['dt_isegmaps_rle'] == r['dt_isegmaps_rle']
            assert w['qry_isegmaps_rle'] == g['qry_isegmaps_rle']
    assert all(not s['busy'] for ring in model._pinned.values() for s in ring)


def test_paste_semantics_of_the_cuda_reference_end_to_end():
    """VERDICT r3 "missing" 4: the reference runs on cuda:0 (main.py:365), where mmdet pastes masks over the whole image
    (skip_empty=False); north_star's CPU reference pastes inside the integer-expanded box.  ``FGN.paste_semantics``:
    at the configured threshold 0.5 both give byte-identical RLE strings; at 0.2 the two differ, and each agrees with
    the oracle run under the same semantic (mask IoU of matched detections >= 0.999: single pixels may sit on the
    threshold)."""
    import copy
    import warnings
    from fgn_amd import rle
    from fgn_amd.config import tiny_config
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import make_batch
    from fgn_amd.weights import init_state_dict
    from oracle import fgn_ref_cpu as O
    cfg = tiny_config(3, 2, width_div=2)
    sd = init_state_dict(cfg, 0)
    batch = make_batch(3, 1, 3, 2, 160, 224, 64)
    mk = lambda c: FGN(3, 2, backbone=c['backbone'], rpn_head=c['rpn_head'], roi_head=c['roi_head'], test_cfg=c['test_cfg'],
                       state_dict=sd)
    model = mk(cfg)
    cpu = model.simple_test(**batch, rescale=True)[0]
    model.paste_semantics = 'cuda'
    cuda = model.simple_test(**batch, rescale=True)[0]
    assert len(cpu['dt_scores']) > 0 and cpu['dt_isegmaps_rle'] == cuda['dt_isegmaps_rle']
    model.paste_semantics = 'gpu'
    with pytest.raises(ValueError):
        model.simple_test(**batch, rescale=True)
    low = copy.deepcopy(cfg)
    low['test_cfg']['rcnn']['mask_thr_binary'] = 0.2
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        m2 = mk(low)
        assert any('paste_semantics' in str(x.message) for x in w)
    got, n_diff = {}, 0
    for sem in ('cpu', 'cuda'):
        m2.paste_semantics = sem
        got[sem] = m2.simple_test(**batch, rescale=True)[0]
        ref = O.simple_test(sd, low, **batch, paste_skip_empty=sem == 'cpu')[0]
        pairs, _, _ = match_detections(ref['dt_bboxes'], ref['dt_cat_ids'], got[sem]['dt_bboxes'], got[sem]['dt_cat_ids'])
        assert len(pairs) >= len(ref['dt_scores']) - 2 and len(pairs) > 0
        for i, j in pairs:
            a, b = rle.decode(ref['dt_isegmaps_rle'][i]), rle.decode(got[sem]['dt_isegmaps_rle'][j])
            union = (a | b).sum()
            assert union == 0 or (a & b).sum() / union >= 0.999, (sem, i)
    for a, b in zip(got['cpu']['dt_isegmaps_rle'], got['cuda']['dt_isegmaps_rle']):
        n_diff += int((rle.decode(a) != rle.decode(b)).sum())
    assert n_diff > 0                    # below 0.5 the semantics really differ


def test_graph_cache_key_covers_what_a_captured_episode_bakes_in():
    """a captured hipGraph bakes in the paste semantic (an argument of the fused RLE kernel), the
    transfer arrangement (which streams the capture forks) and the phase mark (a captured kernel with the counter's
    address).  Changing any of them after the first graphed call must capture a NEW graph, not replay the old one: at a
    mask threshold below 0.5 the two paste semantics give different masks, and a replayed episode must bump the counter it
    was GIVEN for this call."""
    import copy
    from fgn_amd import ops, rle
    from fgn_amd.config import tiny_config
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import make_batch
    from fgn_amd.weights import init_state_dict
    cfg = copy.deepcopy(tiny_config(3, 2, width_div=2))
    cfg['test_cfg']['rcnn']['mask_thr_binary'] = 0.2
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = FGN(3, 2, backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'],
                    test_cfg=cfg['test_cfg'], state_dict=init_state_dict(cfg, 0))
    batch = make_batch(3, 1, 3, 2, 160, 224, 64)
    eager = {}
    for sem in ('cpu', 'cuda'):
        model.paste_semantics = sem
        eager[sem] = model.simple_test(**batch, rescale=True)[0]['dt_isegmaps_rle']
    assert eager['cpu'] != eager['cuda']
    model.use_graphs = True
    for sem in ('cpu', 'cuda', 'cpu'):
        model.paste_semantics = sem
        assert model.simple_test(**batch, rescale=True)[0]['dt_isegmaps_rle'] == eager[sem], sem
    assert len(model._graphs) == 2
    # phase mark: the counter is an argument of the call; two counters -> two graphs, each bumping its own
    model.phase_point = 'rpn'
    marks = torch.zeros(2, dtype=torch.int32, device='cuda')
    for k in (0, 1, 1, 0, 1):
        dets = model.detect_device(batch['qry_img'], batch['spp_imgs'], batch['spp_bboxes'], batch['spp_isegmaps'],
                                   batch['img_shape'], phase_counter=marks[k:k + 1])
        model.pack_results(dets, 1, img_shape=batch['img_shape'])
    torch.cuda.synchronize()
    assert marks.tolist() == [2, 3] and len(model._graphs) == 4
    n = len(model._graphs)
    model.transfer_stream(3)
    model.simple_test(**batch, rescale=True)
    assert len(model._graphs) == n + 1


def test_launch_records_and_packed_transfers():
    """(1) Launch records (``FGN.stamp_capacity``, fgn_profile_stamps): every launch of the persistent GEMM kernel inside a
    captured episode gets a record, and every REPLAY adds exactly one execution with a plausible span.  (2) The packed
    result record / direct uploads (``use_packed_transfers``) give the bytes of the per-field transfers."""
    from fgn_amd import ops
    from fgn_amd.config import fgn_r50_c4_config
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import CONFIGS, make_batch
    from fgn_amd.weights import init_state_dict
    cfg = fgn_r50_c4_config(3, 1)
    sd = init_state_dict(cfg, 0)
    batch = make_batch(5, 1, **CONFIGS['cfg2'])
    batch = {k: (v.pin_memory() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}
    res = {}
    for packed in (False, True):
        model = FGN(3, 1, state_dict=sd)
        model.use_packed_transfers = packed
        model.use_graphs = True
        model.transfer_stream(3)
        model.stamp_capacity = 128
        out = [model.simple_test(**batch, rescale=True) for _ in range(4)]
        for o in out[1:]:
            for key in ('dt_scores', 'dt_bboxes', 'dt_cat_ids'):
                assert np.array_equal(o[0][key], out[0][0][key])
            assert o[0]['dt_isegmaps_rle'] == out[0][0]['dt_isegmaps_rle'] and o[0]['qry_isegmaps_rle'] == out[0][0]['qry_isegmaps_rle']
        res[packed] = out[0][0]
        torch.cuda.synchronize()
        (ge,) = model._graphs.values()
        assert ge.stamps is not None and 0 < ge.stamp_count <= 128
        recs = ops.read_stamps(ge.stamps, ge.stamp_count)
        assert all(r['executions'] == 4 for r in recs), [r['executions'] for r in recs]
        assert all(0.5 < r['min_us'] <= r['max_us'] < 5000 and r['total_us'] >= 4 * r['min_us'] - 1e-6 for r in recs)
        ops.reset_stamps(ge.stamps)
        model.simple_test(**batch, rescale=True)
        torch.cuda.synchronize()
        assert all(r['executions'] == 1 for r in ops.read_stamps(ge.stamps, ge.stamp_count))
    assert len(res[True]['dt_scores']) > 0
    for key in ('dt_scores', 'dt_bboxes', 'dt_cat_ids'):
        assert np.array_equal(res[True][key], res[False][key]), key
    assert res[True]['dt_isegmaps_rle'] == res[False]['dt_isegmaps_rle']
    assert res[True]['qry_isegmaps_rle'] == res[False]['qry_isegmaps_rle']
