"""CPU-only tests: C-ABI surface, host logic, RLE, config mapping, episode sharding (gloo)."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_header_symbol():
    from fgn_amd import build, lib
    path = build.build(verbose=False)          # hipcc cross-compiles gfx950 without a GPU
    assert os.path.exists(path)
    header = open(os.path.join(ROOT, 'include', 'fgn_hip.h')).read()
    declared = set(re.findall(r'\b(fgn_[a-z0-9_]+)\s*\(', header))
    assert declared, 'no declarations parsed'
    assert declared == set(lib.SIGNATURES), declared ^ set(lib.SIGNATURES)
    handle = lib.load()                        # binds every symbol, raises if one is missing
    for name in declared:
        assert hasattr(handle, name)
    assert handle.fgn_abi_version() == lib.ABI_VERSION


def test_no_cpu_fallback():
    from fgn_amd import ops
    from fgn_amd.detector import FGN
    from fgn_amd.lib import FgnHipError
    with pytest.raises(FgnHipError):
        ops.maxpool3x3s2(torch.zeros(1, 4, 4, 4))
    if not torch.cuda.is_available():
        from fgn_amd.config import tiny_config
        from fgn_amd.episodes import make_batch
        cfg = tiny_config(1, 1, 2)
        m = FGN(1, 1, backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'])
        with pytest.raises(FgnHipError):
            m.simple_test(**make_batch(0, 1, 1, 1, 64, 64, 64))


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'fgn_amd')):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', src, re.M), f


def test_rle_roundtrip_and_known_answers():
    from fgn_amd import rle
    from oracle import fgn_ref_cpu as O
    assert rle.encode(np.zeros((2, 2), bool)) == {'size': [2, 2], 'counts': b'4'}
    assert rle.encode(np.ones((2, 2), bool))['counts'] == b'04'
    rng = np.random.RandomState(0)
    for shape in [(1, 1), (5, 7), (64, 33), (120, 200)]:
        for p in (0.02, 0.5, 0.98):
            m = rng.rand(*shape) < p
            enc = rle.encode(m)
            assert enc == O.rle_encode(m)
            assert np.array_equal(rle.decode(enc).astype(bool), m)
    big = np.zeros((800, 1333), bool)
    big[100:700, 200:1200] = True
    enc = rle.encode(big)
    assert np.array_equal(rle.decode(enc).astype(bool), big) and enc == O.rle_encode(big)


def test_reference_style_config_is_accepted():
    """The constructor takes the reference's mmcv dicts (fgn_r50_c4_densecl.py layout) unchanged."""
    from fgn_amd.config import fgn_r50_c4_config
    from fgn_amd.detector import normalise_config
    from _glue import reference_model_cfg
    ref_style = {k: v for k, v in reference_model_cfg().items() if k not in ('n_ways', 'k_shots')}
    got = normalise_config(3, 3, **ref_style)
    want = fgn_r50_c4_config(3, 3)
    for part in ('backbone', 'rpn_head', 'roi_head', 'test_cfg'):
        for k, v in want[part].items():
            g = got[part][k]
            if isinstance(v, dict):
                for kk, vv in v.items():
                    assert tuple(np.ravel(g[kk])) == tuple(np.ravel(vv)), (part, k, kk)
            else:
                assert tuple(np.ravel(g)) == tuple(np.ravel(v)), (part, k)


def test_state_dict_layout_and_loading():
    from fgn_amd.config import fgn_r50_c4_config
    from fgn_amd.detector import FGN
    from fgn_amd.weights import init_state_dict
    cfg = fgn_r50_c4_config(3, 3)
    sd = init_state_dict(cfg, 1)
    n_params = sum(v.numel() for k, v in sd.items() if 'running' not in k)
    assert 34.0e6 < n_params < 35.5e6          # ~34.8 M
    for key in ('backbone.layer3.5.conv3.weight', 'rpn_head.rpn_reg.bias', 'roi_head.shared_head.2.bn3.running_var',
                'roi_head.cls_reg_shared_conv_norm.weight', 'roi_head.mask_head.upsample.weight',
                'roi_head.bbox_head.fc_cls.weight'):
        assert key in sd
    assert tuple(sd['roi_head.cls_reg_shared_conv.weight'].shape) == (1024, 2048, 1, 1)
    m = FGN(3, 3, seed=0)
    m.load_state_dict({'state_dict': sd, 'meta': {}})          # mmcv checkpoint wrapper
    assert torch.equal(m.state_dict()['rpn_head.rpn_cls.weight'], sd['rpn_head.rpn_cls.weight'])
    bad = dict(sd)
    bad['rpn_head.rpn_cls.weight'] = torch.zeros(3, 3)
    with pytest.raises(ValueError):
        m.load_state_dict(bad)
    if not torch.cuda.is_available():                           # the training path has no CPU fallback either
        from fgn_amd.episodes import make_batch
        from fgn_amd.lib import FgnHipError
        with pytest.raises(FgnHipError):
            m(return_loss=True, **make_batch(0, 1, 3, 3, 64, 64, 32))


def test_collate_matches_reference_layout():
    from fgn_amd.episodes import make_batch
    b = make_batch(0, 2, 3, 2, 64, 96, 32, n_qry_objs=3)
    assert b['qry_img'].shape == (2, 3, 64, 96) and b['spp_imgs'].shape == (2, 6, 3, 32, 32)
    assert b['spp_bboxes'].shape == (2, 6, 4) and b['spp_isegmaps'].dtype == torch.bool
    assert isinstance(b['qry_bboxes'], list) and b['qry_bboxes'][0].shape == (3, 4)
    assert b['img_shape'].tolist() == [[64, 96, 3]] * 2 and b['img_shape'].dtype == torch.int32
    assert b['qry_isegmaps'][1].shape == (3, 64, 96)


def _fake_det(e, max_det=4, m=14):
    """Detections of episode e as FGN.detect_device returns them (device dict), with recognisable payloads:
    box values e, labels e % 3, mask probabilities e + 0.01 * detection + 1e-4 * pixel."""
    n = e % max_det
    prob = float(e) + 0.01 * torch.arange(max_det, dtype=torch.float32)[:, None, None] + \
        1e-4 * torch.arange(m * m, dtype=torch.float32).reshape(1, m, m)
    return dict(det_bboxes=torch.full((max_det, 5), float(e)), det_labels=torch.full((max_det,), e % 3, dtype=torch.int64),
                n_dets=torch.tensor([n], dtype=torch.int32), mask_prob=prob.contiguous())


def _dist_worker(rank, world, port, q):
    import torch.distributed as dist
    from fgn_amd import dist as fd
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    try:
        n_episodes = 5                                           # world 2: ranks hold 3 and 2 episodes
        mine = fd.shard_episodes(n_episodes, rank, world)
        per = fd.episodes_per_rank(n_episodes, world)
        recs, cnts = fd.pack_detections([_fake_det(e) for e in mine], 4, pad_to=per)
        assert recs.shape == (per, 4, 6 + 196) and cnts.shape == (per,)
        g_recs, g_cnts = fd.gather_detections(recs, cnts)
        allr, allc = fd.interleave(g_recs), fd.interleave(g_cnts)       # global episode order
        # every rank can emit the result dicts of every episode; a host RLE stand-in keeps this test CPU-only
        rle_fn = lambda prob, boxes, h, w, thr: [{'size': [h, w], 'counts': bytes([int(p[0, 0] * 100) % 256])} for p in prob]
        res = fd.results_from_gathered(allr[:n_episodes], allc[:n_episodes], (32, 48), rle_fn=rle_fn)
        q.put((rank, mine, allr[:, 0, 0].tolist(), allc.tolist(),
               [float(allr[e, 1, 6 + 5]) for e in range(n_episodes)],           # mask payload: detection 1, pixel 5
               [(len(r['dt_scores']), r['dt_cat_ids'].tolist(), r['dt_bboxes'].shape) for r in res]))
    finally:
        dist.destroy_process_group()


def test_episode_sharding_and_gather_world2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + os.getpid() % 1000
    procs = [ctx.Process(target=_dist_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0][1] == [0, 2, 4] and out[1][1] == [1, 3]
    # every rank sees the same globally ordered result: episodes 0,1,2,3,4 and the zero-count pad of rank 1
    for r in out:
        assert r[2] == [0.0, 1.0, 2.0, 3.0, 4.0, 0.0]
        assert r[3] == [0, 1, 2, 3, 0, 0]
        # the mask probabilities travel with their detection, in detection and pixel order
        assert np.allclose(r[4], [e + 0.01 + 5e-4 for e in range(5)], atol=1e-6)
        # result dicts of all episodes, materialised on every rank from the gathered records
        assert [x[0] for x in r[5]] == [0, 1, 2, 3, 0]
        assert r[5][3][1] == [0, 0, 0] and r[5][2][1] == [2, 2] and r[5][3][2] == (3, 4)


def _grad_worker(rank, world, port, q):
    import torch.distributed as dist
    from fgn_amd import dist as fd
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    try:
        g = {'b.weight': torch.full((3, 2), float(rank + 1)), 'a.bias': torch.arange(4, dtype=torch.float32) * (rank + 1)}
        out = fd.allreduce_mean(g)
        q.put((rank, out['b.weight'].tolist(), out['a.bias'].tolist()))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo():
    """Data-parallel training: every rank ends with the mean gradient, one flat bucket, key order independent of the
    dict's insertion order."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29600 + os.getpid() % 1000
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in out:
        assert r[1] == [[1.5, 1.5]] * 3 and r[2] == [0.0, 1.5, 3.0, 4.5]
    from fgn_amd import dist as fd
    g = {'x': torch.ones(2)}
    assert fd.allreduce_mean(g) is g                      # no process group: unchanged


@pytest.mark.parametrize('fail_rank', [-1, 1, 0])
def test_a_failing_rank_ends_the_whole_job(fail_rank):
    """the first multi-GPU run must diagnose itself.  A job of fresh child processes under
    ``torch.distributed.run`` (the launcher the driver uses for ``bench.py --gpus N``; bench.py's own self-launch starts
    the same module as a child and returns its exit code) gathers detection records step by step; when one rank raises
    mid-run the launcher must end EVERY rank and exit non-zero within the timeout - no rank may sit in the collective
    waiting for the dead one.  Control: without the injected failure the same job exits 0."""
    import socket
    import subprocess
    import sys
    import time
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_rank_fail_child.py')
    env = dict(os.environ, FGN_FAIL_RANK=str(fail_rank), FGN_FAIL_STEP='3', FGN_PG_TIMEOUT='30', OMP_NUM_THREADS='1')
    t0 = time.time()
    proc = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                           '--master-addr', '127.0.0.1', '--master-port', str(port), child],
                          env=env, capture_output=True, text=True, timeout=150)
    took = time.time() - t0
    if fail_rank < 0:
        assert proc.returncode == 0, proc.stderr[-2000:]
        assert proc.stdout.count('done') == 2
    else:
        assert proc.returncode != 0
        assert 'injected failure' in proc.stderr
        assert 'done' not in proc.stdout                       # the surviving rank did not run to completion either
        assert took < 120, took


def test_gather_is_identity_without_process_group():
    from fgn_amd import dist as fd
    recs, cnts = fd.pack_detections([_fake_det(3), _fake_det(6)], 4)
    g_recs, g_cnts = fd.gather_detections(recs, cnts)
    assert g_recs.shape == (1, 2, 4, 202) and g_cnts.tolist() == [[3, 2]]
    with pytest.raises(ValueError):
        fd.results_from_gathered(recs[:, :, :6], cnts, (8, 8))          # records without mask payload


def _res(h, w, gt_boxes, gt_cats, dt_boxes, dt_cats, dt_scores):
    """Result dict with rectangular masks; boxes are YXYX."""
    from fgn_amd import rle

    def m(b):
        a = np.zeros((h, w), bool)
        a[int(b[0]):int(b[2]), int(b[1]):int(b[3])] = True
        return rle.encode(a)
    return {'qry_img_shape': np.array([h, w, 3]), 'qry_bboxes': np.array(gt_boxes, np.float32).reshape(-1, 4),
            'qry_cat_ids': np.array(gt_cats, np.int64), 'qry_isegmaps_rle': [m(b) for b in gt_boxes],
            'dt_bboxes': np.array(dt_boxes, np.float32).reshape(-1, 4), 'dt_cat_ids': np.array(dt_cats, np.int64),
            'dt_scores': np.array(dt_scores, np.float32), 'dt_isegmaps_rle': [m(b) for b in dt_boxes]}


def test_fsiseg_eval_known_answers():
    from fgn_amd.fsiseg_eval import FSISEGEval, evaluate_results
    # perfect detections -> AP = AR = 1 for both IoU types
    r = _res(40, 50, [[2, 3, 20, 25], [10, 30, 30, 45]], [0, 1], [[2, 3, 20, 25], [10, 30, 30, 45]], [0, 1], [.9, .8])
    out = evaluate_results([r], 2)
    assert all(abs(out[k] - 1.0) < 1e-12 for k in ('bbox_mAP50', 'bbox_mAR', 'segm_mAP50', 'segm_mAR'))
    # one class, 2 GT; detections by score: TP, FP, TP  -> precision envelope [1, 2/3, 2/3], recall [.5,.5,1]
    r = _res(40, 50, [[0, 0, 10, 10], [20, 20, 30, 30]], [0, 0],
             [[0, 0, 10, 10], [30, 0, 39, 9], [20, 20, 30, 31]], [0, 0, 0], [.9, .8, .7])
    ev = FSISEGEval(results=[r], n_ways=1, iou_type='segm')
    got = ev.run()
    # recall thresholds 0..0.5 -> p=1 (6 points), 0.6..1.0 -> p=2/3 (5 points)
    assert abs(got['mAP'] - (6 * 1.0 + 5 * 2 / 3) / 11) < 1e-9 and got['mAR'] == 1.0
    # wrong class is a miss; category without GT and without detections is skipped (-1)
    r = _res(40, 50, [[0, 0, 10, 10]], [1], [[0, 0, 10, 10]], [0], [.9])
    got = FSISEGEval(results=[r], n_ways=3, iou_type='bbox').run()
    assert got['mAP'] == 0.0 and got['mAR'] == 0.0
    # a duplicate detection of an already matched GT is a false positive
    r = _res(40, 50, [[0, 0, 10, 10]], [0], [[0, 0, 10, 10], [0, 0, 10, 10]], [0, 0], [.9, .8])
    got = FSISEGEval(results=[r], n_ways=1).run()
    assert abs(got['mAP'] - 1.0) < 1e-12       # the TP comes first, precision 1 at every recall level


def test_synthetic_dataset_contract_and_evaluate(tmp_path):
    """BaseFewShotISEG surface used by the eval loop: len/getitem sample dict, chunked pickles, evaluate keys."""
    from torch.utils.data import DataLoader
    from fgn_amd.episodes import collate
    from fgn_amd.fewshot_ds import SyntheticFewShotISEG, write_chunked
    ds = SyntheticFewShotISEG(3, 2, length=5, height=70, width=90, spp_img_size=32, batch=2)
    assert len(ds) == 5 and (ds.height, ds.width) == (64, 80)          # x16 rounding for batch > 1
    s = ds[3]
    for key in ('idx', 'qry_child_idx', 'qry_img', 'qry_cat_ids', 'qry_bboxes', 'qry_isegmaps', 'spp_imgs',
                'spp_bboxes', 'spp_isegmaps', 'cats_ids_to_sample_real', 'spp_insts_ids', 'img_shape'):
        assert key in s
    batches = list(DataLoader(ds, batch_size=ds.batch, collate_fn=collate))
    assert len(batches) == 3 and batches[0]['qry_img'].shape == (2, 3, 64, 80)
    # fake "perfect" results: detections = ground truth
    fake = []
    for b in batches:
        one = []
        for i in range(b['qry_img'].shape[0]):
            gt = b['qry_isegmaps'][i].numpy()
            one.append(_res(64, 80, b['qry_bboxes'][i].numpy().tolist(), b['qry_cat_ids'][i].tolist(),
                            b['qry_bboxes'][i].numpy().tolist(), b['qry_cat_ids'][i].tolist(),
                            [0.9] * len(gt)))
        fake.append(one)
    write_chunked(iter(fake), str(tmp_path / 'ResultsChunked'), chunk=2)
    assert sorted(os.listdir(tmp_path / 'ResultsChunked')) == ['00.pkl', '01.pkl', '02.pkl']
    m = ds.evaluate(results_pkl_dir_fp=str(tmp_path / 'ResultsChunked'))
    assert set(m) == {'isegm_mAP', 'isegm_mAR', 'bbox_mAP', 'bbox_mAR'}
    assert m['bbox_mAP'] > 0.99 and m['isegm_mAP'] > 0.99


def test_cluttered_chars_dataset_contract():
    """MNISTISEG/OMNIISEG-shaped episodes; sample dict of base_fst.py:1248-1266."""
    from fgn_amd import cluttered_chars as cc
    from fgn_amd.episodes import collate
    from fgn_amd.fewshot_ds import ClutteredCharsFewShotISEG
    im = cc.make_image(3, 256, np.arange(10))
    n = len(im['cat_ids'])
    assert 2 <= n <= 6 and im['img'].dtype == np.uint8 and im['isegmaps'].shape == (n, 256, 256)
    for b, m in zip(im['bboxes'], im['isegmaps']):
        ys, xs = np.nonzero(m)
        assert m.sum() > 0 and ys.min() >= b[0] and ys.max() < b[2] and xs.min() >= b[1] and xs.max() < b[3]
    boxes = im['bboxes']
    for i in range(n):                          # placement rule: pairwise IoU below 0.2
        for j in range(i):
            assert cc._iou_one_to_many(boxes[i], boxes[j:j + 1])[0] < 0.2
    ds = ClutteredCharsFewShotISEG('OMNIISEG', n_ways=3, k_shots=2, n_imgs=12, img_size=256, spp_img_size=128)
    s = ds[4]
    assert s['spp_imgs'].shape == (6, 3, 128, 128) and s['spp_isegmaps'].shape == (6, 128, 128)
    assert s['qry_img'].shape == (3, 256, 256) and s['qry_img'].dtype == torch.float32
    assert set(s['qry_cat_ids'].tolist()) <= {0, 1, 2} and len(s['qry_cat_ids']) >= 1
    assert np.array_equal(s['cats_ids_to_sample_real'][s['qry_cat_ids']], s['qry_cat_ids_real'])
    # class-major supports of the sampled real classes, taken from other images, box fills ~0.8 of the crop
    for i, p in enumerate(s['spp_insts_ids']):
        assert ds.inst_cat[p] == s['cats_ids_to_sample_real'][i // 2]
        side = max(s['spp_bboxes'][i][2] - s['spp_bboxes'][i][0], s['spp_bboxes'][i][3] - s['spp_bboxes'][i][1])
        # base_fst.py:264-265: offset ratio 0.12 for fill ratio 0.8 -> the longer side fills 1 / 1.24 of the crop
        # (integer offsets / parity padding move it by a pixel or two of the source crop)
        assert abs(side / 128 - 1 / 1.24) < 0.06 and s['spp_isegmaps'][i].sum() > 0
    assert ds[4]['qry_img'].equal(s['qry_img'])                         # deterministic
    b = collate([ds[0], ds[1], ds[2]])
    assert b['qry_img'].shape == (3, 3, 256, 256) and isinstance(b['qry_cat_ids_real'], list)
    ds.shuffle = True
    ds.reshuffle(e=3)
    assert sorted(ds.order.tolist()) == list(range(12))
    # a pool smaller than the class vocabulary: supports come only from classes that have an instance
    small = ClutteredCharsFewShotISEG('OMNIISEG', n_ways=3, k_shots=1, n_imgs=4, img_size=256, spp_img_size=128)
    for i in range(4):
        smp = small[i]
        assert all(c in set(small.inst_cat.tolist()) for c in smp['cats_ids_to_sample_real'])


def _golden_data_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location('make_golden_data', os.path.join(
        os.path.dirname(os.path.abspath(__file__)), 'golden', 'make_golden_data.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)             # seeded input builders only: nothing reads /root/reference at import
    return m


def test_episode_geometry_matches_the_reference_data_side_goldens(golden_dir):
    """tests/golden/data_side.npz = the reference's own ``get_new_shape`` (create_img_from_chars.py:250-267),
    ``BaseFewShotISEG.cut_algorithm`` / ``get_crop`` (base_fst.py:991-1040) and the aspect-ratio-grouped branch of
    ``reshuffle`` (base_fst.py:626-727), run by tests/golden/make_golden_data.py.  Bit-exact."""
    from fgn_amd import fewshot_ds as F
    G = _golden_data_module()
    z = np.load(os.path.join(golden_dir, 'data_side.npz'))
    n_assert = 0
    for (h, w), want in zip(z['new_shape_hw'], z['new_shape_out']):
        try:
            got = F.get_new_shape(int(h), int(w))
        except AssertionError:               # the reference asserts on the aspect-ratio drift, so does the build
            got, n_assert = np.array([-1, -1]), n_assert + 1
        assert np.array_equal(got, want), (h, w, got, want)
    assert n_assert == int((z['new_shape_out'][:, 0] < 0).sum()) > 0
    assert np.array_equal(np.stack([F.get_new_shape(h, w, 128, 256) for h, w in ((100, 100), (64, 200), (300, 100))]),
                          z['new_shape_small'])
    assert F.get_new_shape(480, 640).tolist() == [800, 1066] and F.get_new_shape(333, 1000).tolist() == [443, 1333]
    assert np.array_equal(np.stack([F.cut_algorithm(*c) for c in G.cut_cases()]), z['cut_out'])
    assert int(z['crop_n']) == len(G.crop_cases())
    for i, (img, y0, x0, y1, x1, ho, wo, sq, mode) in enumerate(G.crop_cases()):
        crop, box = F.get_crop(img, y0, x0, y1, x1, ho, wo, crop_square=sq, mode=mode)
        mask, _ = F.get_crop(img[..., :1] > 127, y0, x0, y1, x1, ho, wo, crop_square=sq, mode='constant')
        assert np.array_equal(crop, z[f'crop{i}_out']) and np.array_equal(box, z[f'crop{i}_box']), i
        assert np.array_equal(mask, z[f'crop{i}_mask']) and mask.dtype == z[f'crop{i}_mask'].dtype, i
    sizes = G.ar_sizes()
    for tag in 'abc':
        shuffle, batch, seed = (int(v) for v in z[f'ar_{tag}_cfg'])
        order, groups, hws = F.ar_grouped_order([w / h for w, h in sizes], batch, shuffle=bool(shuffle), seed=seed)
        assert np.array_equal(order, z[f'ar_{tag}_order']) and np.array_equal(groups, z[f'ar_{tag}_groups'])
        assert np.array_equal(hws, z[f'ar_{tag}_hws'])
        rounded = np.around([w / h for w, h in sizes], 1)
        for c in range(0, len(order), batch):        # every chunk holds one aspect-ratio group
            assert len(set(rounded[order[c:c + batch]].tolist())) == 1 and len(set(groups[c:c + batch].tolist())) == 1
    assert F.spp_offset_ratio(0.8) == 0.12 and F.spp_offset_ratio(1.0) == 0.0 and F.spp_offset_ratio(0.5) == 0.5


def test_support_from_instance_follows_get_support_geometry():
    """``get_support`` (base_fst.py:1104-1155) on a synthetic instance: the golden-pinned crop, then the restated
    imgaug resize (longer side -> S) and centre pad; the box travels with the pixels."""
    from fgn_amd import fewshot_ds as F
    img = np.full((60, 90, 3), 255, np.uint8)
    mask = np.zeros((60, 90), bool)
    img[20:40, 10:70] = (10, 200, 30)
    mask[20:40, 10:70] = True
    crop, box, m = F.support_from_instance(img, [20., 10., 40., 70.], mask, 64, 0.8)
    assert crop.shape == (64, 64, 3) and m.shape == (64, 64) and box.dtype == np.float32
    # offsets floor(20 * .12) = 2, floor(60 * .12) = 7 -> 74 wide; squared: 74 x 74 -> scale 64 / 74
    sc = 64 / 74.0
    assert np.allclose(box, [(2 + 25) * sc, 7 * sc, (2 + 25 + 20) * sc, 67 * sc], atol=1e-4)
    y0, x0, y1, x1 = np.round(box).astype(int)
    inner = m[y0 + 1:y1 - 1, x0 + 1:x1 - 1]
    assert inner.all() and m.sum() <= (y1 - y0 + 2) * (x1 - x0 + 2)            # the mask fills its box, nothing outside
    assert (np.abs(crop[y0 + 2:y1 - 2, x0 + 2:x1 - 2].astype(int) - [10, 200, 30]) <= 2).all()
    # a tall instance at the image border: reflect context for the image, zeros for the mask, padding left / right
    crop2, box2, m2 = F.support_from_instance(img, [0., 0., 60., 20.], np.ones((60, 90), bool), 32, 0.8, crop_square=False)
    assert crop2.shape == (32, 32, 3) and (box2[2] - box2[0]) > (box2[3] - box2[1]) and not m2[:, :4].any()


def test_resnet18_extension_config_weights_and_reference_style_dict():
    """The R18 variant of cfg2 (BASELINE.json; no reference config exists): flat config, seeded weights in mmdet's
    BasicBlock layout, the mmcv-style backbone dict (depth=18) normalises to the same flat config, and the oracle
    runs it end to end."""
    from fgn_amd.config import fgn_r18_c4_config
    from fgn_amd.detector import normalise_config
    from fgn_amd.episodes import make_batch
    from fgn_amd.weights import init_state_dict
    from oracle import fgn_ref_cpu as O
    cfg = fgn_r18_c4_config(3, 1)
    sd = init_state_dict(cfg, 0)
    assert sd['backbone.layer1.0.conv1.weight'].shape == (64, 64, 3, 3) and 'backbone.layer1.0.downsample.0.weight' not in sd
    assert sd['backbone.layer3.0.downsample.0.weight'].shape == (256, 128, 1, 1) and 'backbone.layer3.2.conv1.weight' not in sd
    assert sd['roi_head.cls_reg_shared_conv.weight'].shape == (256, 512, 1, 1)
    assert sd['roi_head.shared_head.0.conv2.weight'].shape == (128, 128, 3, 3)
    n = normalise_config(3, 1, backbone=dict(type='ResNet', depth=18, num_stages=4, out_indices=(2,), strides=(1, 2, 2, 2),
                                             norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='pytorch'))
    assert n['backbone']['block'] == 'basic' and n['backbone']['stage_blocks'] == (2, 2, 2)
    out = O.simple_test(sd, cfg, **make_batch(2, 1, 3, 1, 96, 128, 64))
    assert set(out[0]) >= {'dt_scores', 'dt_bboxes', 'dt_cat_ids', 'dt_isegmaps_rle', 'qry_isegmaps_rle'}


def test_step_lr_schedule_of_the_reference():
    """fgn_train_schedule.py:17-23: Step policy [3] x0.1, min_lr 1e-6, linear warm-up over 100 iterations from 1 %."""
    from fgn_amd.train import step_lr
    assert step_lr(0.005, 0, 0) == pytest.approx(0.005 * 0.01)
    assert step_lr(0.005, 50, 0) == pytest.approx(0.005 * (1 - 0.5 * 0.99))
    assert step_lr(0.005, 100, 0) == 0.005 and step_lr(0.005, 10 ** 4, 2) == 0.005
    assert step_lr(0.005, 10 ** 4, 3) == pytest.approx(0.0005)
    assert step_lr(1e-6, 10 ** 4, 5) == 1e-6


def test_host_sampler_matches_the_oracle_sampler_under_the_same_seed():
    """fgn_amd.train._sample (the product's host-side RandomSampler bookkeeping, numpy on the copied assignment
    vector) against the oracle's restatement of BaseSampler.sample / RandomSampler, which is itself pinned by the
    vendored my_random_sampler.py golden."""
    from fgn_amd.train import _sample
    from oracle import fgn_train_cpu as T
    g = torch.Generator().manual_seed(3)
    for n, n_pos, num, frac in ((63000, 40, 64, 0.5), (63000, 5, 64, 0.5), (2000, 300, 128, 0.25), (50, 0, 128, 0.25)):
        gi = torch.zeros(n, dtype=torch.long)
        gi[torch.randperm(n, generator=g)[:n // 7]] = -1
        gi[torch.randperm(n, generator=g)[:n // 9]] = -2                     # anchors outside the image
        if n_pos:
            gi[torch.randperm(n, generator=g)[:n_pos]] = torch.randint(1, 4, (n_pos,), generator=g)
        torch.manual_seed(11)
        pos, neg = _sample(gi.numpy().astype(np.int32), num, frac, torch.randperm)
        # the reference samples on the compacted list of inside anchors: same relative order, same permutation
        inside = gi != -2
        torch.manual_seed(11)
        ref = T.random_sample(gi[inside], torch.zeros(int(inside.sum()), 4), torch.zeros(3, 4), None, num, frac, False)
        full = torch.nonzero(inside).view(-1)
        assert np.array_equal(pos, full[ref['pos_inds']].numpy())
        assert np.array_equal(neg, full[ref['neg_inds']].numpy())


def test_trainer_refuses_the_trainable_backbone_config():
    from fgn_amd.config import tiny_config
    from fgn_amd.detector import FGN
    from fgn_amd.train import Trainer
    cfg = tiny_config(3, 1, width_div=8, scratch=True)
    m = FGN(3, 1, backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'])
    with pytest.raises(NotImplementedError):
        Trainer(m)


def test_backbone_only_pretrained_checkpoint_loads_into_backbone_only(tmp_path):
    """The reference's headline config initialises the BACKBONE from a DenseCL ResNet-50 file through mmcv's
    ``Pretrained`` init_cfg (fgn_r50_c4_densecl.py:4-11, 39-41; ``model.init_weights()``, main.py:431-434): a
    torchvision-style state dict without module prefix, with layer4 / fc entries the C4 model lacks."""
    import warnings
    from fgn_amd.config import tiny_config
    from fgn_amd.detector import FGN
    from fgn_amd.weights import init_state_dict
    cfg = tiny_config(3, 1, width_div=8)
    donor = init_state_dict(cfg, 7)
    file_sd = {k[len('backbone.'):]: v for k, v in donor.items() if k.startswith('backbone.')}
    file_sd['layer4.0.conv1.weight'] = torch.zeros(4, 4, 1, 1)
    file_sd['fc.weight'] = torch.zeros(10, 4)
    path = str(tmp_path / 'densecl_like.pth')
    torch.save({'state_dict': {'module.' + k: v for k, v in file_sd.items()}, 'meta': {}}, path)
    bb = dict(type='ResNet', depth=50, num_stages=4, strides=(1, 2, 2, 2), out_indices=(2,), frozen_stages=4,
              norm_cfg=dict(type='BN', requires_grad=False), norm_eval=True, style='pytorch',
              init_cfg=[dict(type='Pretrained', checkpoint=path)])
    m = FGN(3, 1, backbone=bb, seed=0)                      # full-width R50 from the reference-style dict ...
    assert m.backbone_pretrained == path
    m = FGN(3, 1, backbone=dict(cfg['backbone'], init_cfg=[dict(type='Pretrained', checkpoint=path)]),
            rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'], test_cfg=cfg['test_cfg'], seed=0)   # ... narrow for the load
    before = m.state_dict()
    m.init_weights()
    after = m.state_dict()
    for k, v in after.items():
        if k.startswith('backbone.'):
            assert torch.equal(v, donor[k]), k
        else:
            assert torch.equal(v, before[k]), k
    rep = m.load_backbone_pretrained(file_sd)                # bare dict, no prefix
    assert rep['missing'] == [] and sorted(rep['unexpected']) == ['backbone.fc.weight', 'backbone.layer4.0.conv1.weight']
    with pytest.raises(KeyError):
        m.load_backbone_pretrained({'fc.weight': torch.zeros(2, 2)})
    # a checkpoint without the optional num_batches_tracked buffers still loads strictly
    m.load_state_dict({k: v for k, v in donor.items() if not k.endswith('num_batches_tracked')})
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        c2 = tiny_config(3, 1)
        c2['test_cfg']['rcnn']['mask_thr_binary'] = 0.3
        FGN(3, 1, backbone=c2['backbone'], rpn_head=c2['rpn_head'], roi_head=c2['roi_head'], test_cfg=c2['test_cfg'])
    assert any('mask_thr_binary' in str(x.message) for x in w)


def test_rank_placement_from_a_kfd_topology(tmp_path):
    """bench.pin_rank: ranks are pinned to the cores local to their GPU (sysfs ``local_cpulist`` of the GPU's DRM render
    node, found through the KFD topology), split evenly among the ranks whose GPUs share that list; a node without
    locality information falls back to an even split of the allowed cores.  Fake sysfs tree: 2 CPU nodes + 4 GPUs."""
    import os
    import bench
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 4:
        pytest.skip('needs 4 cores')
    half = len(allowed) // 2
    lists = [allowed[:half], allowed[half:]]
    fmt = lambda cpus: ','.join(str(c) for c in cpus)
    nodes = tmp_path / 'class' / 'kfd' / 'kfd' / 'topology' / 'nodes'
    for n in range(6):                                   # nodes 0, 1: CPUs (simd_count 0); 2..5: GPUs
        d = nodes / str(n)
        d.mkdir(parents=True)
        gpu = n >= 2
        (d / 'properties').write_text(f'cpu_cores_count {0 if gpu else 64}\nsimd_count {1024 if gpu else 0}\n'
                                      f'drm_render_minor {126 + n if gpu else 0}\nname x\n')
        if gpu:
            dev = tmp_path / 'class' / 'drm' / f'renderD{126 + n}' / 'device'
            dev.mkdir(parents=True)
            (dev / 'local_cpulist').write_text(fmt(lists[(n - 2) // 2]) + '\n')
            (dev / 'numa_node').write_text(f'{(n - 2) // 2}\n')
    gpus = bench.gpu_local_cpus(str(tmp_path))
    assert [g['numa'] for g in gpus] == [0, 0, 1, 1] and gpus[0]['cpus'] == lists[0] and gpus[3]['cpus'] == lists[1]
    got = [bench.pin_rank(r, 4, sysfs=str(tmp_path), apply=False) for r in range(4)]
    assert all(g['policy'] == 'gpu-local' and g['sharers'] == 2 for g in got)
    sets = [set(g['cpu_list']) for g in got]
    assert sets[0] | sets[1] <= set(lists[0]) and sets[2] | sets[3] <= set(lists[1])
    assert not (sets[0] & sets[1]) and not (sets[2] & sets[3]) and all(len(s_) == half // 2 for s_ in sets)
    # more ranks than GPUs (the 5-rank rehearsal on one card, HIP_VISIBLE_DEVICES per rank): ranks wrap around the GPUs
    # and the cores of a cpulist are split among ALL the ranks that land on it - no two ranks overlap
    if half >= 4:
        got = [bench.pin_rank(r, 6, sysfs=str(tmp_path), apply=False) for r in range(6)]
        sets = [set(g['cpu_list']) for g in got]
        assert all(g['policy'] == 'gpu-local' for g in got) and [g['sharers'] for g in got] == [4, 4, 2, 2, 4, 4]
        assert all(not (sets[a] & sets[b]) for a in range(6) for b in range(a + 1, 6))
        assert [g['gpu'] for g in got] == [0, 1, 2, 3, 0, 1]
    # no topology at all -> even contiguous split of the allowed cores; one rank -> untouched
    got = [bench.pin_rank(r, 4, sysfs=str(tmp_path / 'nothing'), apply=False) for r in range(4)]
    assert all(g['policy'] == 'even-split' for g in got) and not (set(got[0]['cpu_list']) & set(got[1]['cpu_list']))
    assert bench.pin_rank(0, 1, sysfs=str(tmp_path))['policy'] == 'none'
    assert bench._cpulist('0-3,8,10-11') == [0, 1, 2, 3, 8, 10, 11]


def test_single_thread_randperm_draws_the_same_permutation():
    """fgn_amd.train._perm runs large ``torch.randperm`` draws on one intra-op thread (the parallel identity fill costs
    ms on a many-core host); the permutation - and with it seed parity with mmdet's RandomSampler - must not change."""
    from fgn_amd.train import _perm
    k = torch.get_num_threads()
    for n in (100, 32768, 32769, 63000, 200000):
        torch.manual_seed(n)
        want = torch.randperm(n)
        torch.manual_seed(n)
        got = _perm(torch.randperm, n)
        assert torch.equal(want, got), n
    assert torch.get_num_threads() == k
    g = torch.Generator().manual_seed(3)
    assert _perm(lambda m: torch.randperm(m, generator=g), 5).numel() == 5        # any callable passes through


def test_collate_matches_the_reference_collate_fn_new(golden_dir):
    """tests/golden/data_side.npz holds the batch the reference's own ``collate_fn_new`` (main.py:62-76) builds from two
    seeded sample dicts: ``episodes.collate`` must give the same keys in the same order, the same container types,
    dtypes, shapes and values."""
    from fgn_amd.episodes import collate
    G = _golden_data_module()
    z = np.load(os.path.join(golden_dir, 'data_side.npz'))
    got = collate(G.collate_samples())
    assert list(got) == [str(k) for k in z['collate_keys']]
    for k, v in got.items():
        if isinstance(v, list):
            assert len(v) == int(z[f'collate__{k}__n'])
            for j, t in enumerate(v):
                want = z[f'collate__{k}__{j}']
                assert isinstance(t, torch.Tensor) and t.numpy().dtype == want.dtype and np.array_equal(t.numpy(), want), (k, j)
        else:
            want = z[f'collate__{k}']
            assert isinstance(v, torch.Tensor) and v.numpy().dtype == want.dtype and np.array_equal(v.numpy(), want), k


def test_evaluator_records_and_summary_match_the_reference_fsisegeval(tmp_path):
    """tests/golden/fsiseg_eval.npz holds what the reference's FSISEGEval builds from seeded result pickles
    (records, parameters, (image, category) grouping) and its summarize_short on a synthetic eval; everything
    evaluate()/accumulate() do is pycocotools and stays unpinned."""
    import pickle
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), 'golden'))
    import make_golden_eval as G
    from fgn_amd.fsiseg_eval import FSISEGEval
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'fsiseg_eval.npz'))
    results = G.make_results()
    for name, chunk in (('00.pkl', results[:3]), ('01.pkl', results[3:])):
        with open(tmp_path / name, 'wb') as fh:
            pickle.dump(chunk, fh)
    for kind in ('segm', 'bbox'):
        ev = FSISEGEval(results_pkl_dir_fp=str(tmp_path), n_ways=3, iou_type=kind)
        imgs, gts, dts = ev.annotations()
        pre = kind + '__'
        np.testing.assert_array_equal(np.array([[im['height'], im['width']] for im in imgs]), z[pre + 'imgs'])
        for tag, recs in (('gt', gts), ('dt', dts)):
            for key in ('image_id', 'id', 'category_id', 'area'):
                np.testing.assert_array_equal(np.array([r[key] for r in recs]), z[pre + tag + '_' + key], err_msg=tag + key)
            np.testing.assert_array_equal(np.array([r['bbox'] for r in recs]).reshape(-1, 4), z[pre + tag + '_bbox'])
        np.testing.assert_array_equal(np.array([r['score'] for r in dts]), z[pre + 'dt_score'])
        assert all(r['iscrowd'] == 0 and not r['ignore'] for r in gts) and not z[pre + 'gt_flags'].any()
        np.testing.assert_array_equal(ev.rec_thrs, z[pre + 'recThrs'])
        np.testing.assert_array_equal([ev.iou_thr, ev.max_dets, ev.area_rng[0], ev.area_rng[1], 1], z[pre + 'scalars'])
        np.testing.assert_array_equal(ev.img_ids, z[pre + 'imgIds'])
        np.testing.assert_array_equal(ev.cat_ids, z[pre + 'catIds'])
        g, d = ev.groups()
        n_g = int(z[pre + 'n_gt_groups'])
        keys, sizes = z[pre + 'group_keys'], z[pre + 'group_sizes']
        assert sorted(g) == [tuple(k) for k in keys[:n_g]] and [g[k] for k in sorted(g)] == list(sizes[:n_g])
        assert sorted(d) == [tuple(k) for k in keys[n_g:]] and [d[k] for k in sorted(d)] == list(sizes[n_g:])
    p, r = G.eval_arrays()
    ev.eval = {'precision': p[0, :, :, 0, 0], 'recall': r[0, :, 0, 0]}
    out = ev.summarize_short()
    np.testing.assert_array_equal([out['mAP'], out['mAR']], z['summary'])
    ev.eval = {'precision': -np.ones((11, 3)), 'recall': -np.ones(3)}
    out = ev.summarize_short()
    np.testing.assert_array_equal([out['mAP'], out['mAR']], z['summary_empty'])


def _reference_shaped_detector(cfg):
    """An nn.Module tree with the REGISTRATION ORDER of the reference's detector (mmdet 2.18 ``ResNet`` with all four
    stages - main.py:402-405 only shortens the list of stage names -, ``RPNHead``, then ``FGNRoIHead``: bbox_head and
    mask_head from ``StandardRoIHead.__init__``, shared_head / cls_reg_shared_conv / cls_reg_shared_conv_norm added by
    fgn_roi_head.py:197-200), built independently of ``fgn_amd.train.reference_param_order``: torch's own
    ``named_parameters()`` then gives the index space of the reference's optimizer."""
    import torch.nn as nn
    bb = cfg['backbone']

    class Bottleneck(nn.Module):
        def __init__(self, cin, planes, out, down):
            super().__init__()
            self.conv1 = nn.Conv2d(cin, planes, 1, bias=False); self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False); self.bn2 = nn.BatchNorm2d(planes)
            self.conv3 = nn.Conv2d(planes, out, 1, bias=False); self.bn3 = nn.BatchNorm2d(out)
            self.relu = nn.ReLU()
            self.downsample = nn.Sequential(nn.Conv2d(cin, out, 1, bias=False), nn.BatchNorm2d(out)) if down else None

    def res_layer(cin, planes, out, n, force_down=True):
        return nn.Sequential(*[Bottleneck(cin if i == 0 else out, planes, out, i == 0 and (force_down or cin != out))
                               for i in range(n)])

    class Backbone(nn.Module):
        def __init__(self):
            super().__init__()
            c = bb['stem_channels']
            self.conv1 = nn.Conv2d(3, c, 7, bias=False); self.bn1 = nn.BatchNorm2d(c)
            planes = list(bb['stage_planes']) + [2 * bb['stage_planes'][-1]]
            blocks = list(bb['stage_blocks']) + [3]
            for i, (p, n) in enumerate(zip(planes, blocks)):
                setattr(self, f'layer{i + 1}', res_layer(c, p, 4 * p, n))
                c = 4 * p

    C = 4 * bb['stage_planes'][-1]
    A = 15

    class RPN(nn.Module):
        def __init__(self):
            super().__init__()
            self.rpn_conv = nn.Conv2d(C, C, 3); self.rpn_cls = nn.Conv2d(C, A, 1); self.rpn_reg = nn.Conv2d(C, 4 * A, 1)

    class BBoxHead(nn.Module):
        def __init__(self):
            super().__init__()
            self.fc_cls = nn.Linear(C, 2); self.fc_reg = nn.Linear(C, 4)

    mc = cfg['roi_head']['mask_head']['conv_out_channels']

    class ConvModule(nn.Module):
        def __init__(self, cin):
            super().__init__()
            self.conv = nn.Conv2d(cin, mc, 3)

    class MaskHead(nn.Module):
        def __init__(self):
            super().__init__()
            self.convs = nn.ModuleList([ConvModule(C if i == 0 else mc) for i in range(4)])
            self.upsample = nn.ConvTranspose2d(mc, mc, 2, 2); self.conv_logits = nn.Conv2d(mc, 1, 1)

    class RoIHead(nn.Module):
        def __init__(self):
            super().__init__()
            self.bbox_head = BBoxHead(); self.mask_head = MaskHead()
            self.shared_head = res_layer(C, C // 2, C, 3, force_down=False)
            self.cls_reg_shared_conv = nn.Conv2d(2 * C, C, 1); self.cls_reg_shared_conv_norm = nn.GroupNorm(32, C)

    class Det(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = Backbone(); self.rpn_head = RPN(); self.roi_head = RoIHead()
    return Det()


def test_optimizer_index_space_is_the_reference_models_named_parameters():
    """a reference checkpoint's Adagrad state indexes EVERY parameter (mmcv's constructor lists frozen ones
    under paramwise_cfg; Adagrad creates state for all).  ``reference_param_order`` must be ``named_parameters()`` of a
    detector with the reference's registration order - names AND shapes, layer4 included."""
    from fgn_amd.config import tiny_config
    from fgn_amd.train import reference_param_order, trainable_names
    from fgn_amd.weights import init_state_dict
    cfg = tiny_config(3, 2, width_div=2)
    sd = init_state_dict(cfg, 0)
    det = _reference_shaped_detector(cfg)
    want = [(k, tuple(p.shape)) for k, p in det.named_parameters()]
    got = reference_param_order(sd, cfg['backbone'])
    assert [k for k, _ in got] == [k for k, _ in want]
    assert got == want
    # every state-dict tensor of this build that is a parameter sits in that order; only layer4 is extra
    names = [k for k, _ in got]
    mine = [k for k in sd if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))]
    assert set(mine) < set(names) and all(k.startswith('backbone.layer4.') for k in set(names) - set(mine))
    assert set(trainable_names(sd)) == {k for k in names if not k.startswith('backbone.')}
    # a real torch Adagrad over ALL parameters, one group each (frozen backbone included), has that many entries
    for k, p in det.named_parameters():
        p.requires_grad_(not k.startswith('backbone.'))
    opt = torch.optim.Adagrad([{'params': [p]} for p in det.parameters()], lr=0.005, weight_decay=1e-5)
    osd = opt.state_dict()
    assert len(osd['state']) == len(osd['param_groups']) == len(names)
    # the from-scratch configuration (fgn_r50_c4_scratch.py:9-30: num_stages=3, deep stem, GroupNorm) registers NO layer4
    # and other norm names: its index space is exactly this build's own parameters, nothing phantom
    scfg = tiny_config(3, 2, width_div=2, scratch=True)
    ssd = init_state_dict(scfg, 0)
    sgot = [k for k, _ in reference_param_order(ssd, scfg['backbone'])]
    mine = [k for k in ssd if not k.endswith(('running_mean', 'running_var', 'num_batches_tracked'))]
    assert sorted(sgot) == sorted(mine) and not any('layer4' in k for k in sgot)
    assert sgot[0] == 'backbone.stem.0.weight' and 'backbone.layer1.0.gn1.weight' in sgot
    assert sgot.index('backbone.layer3.5.gn3.bias') < sgot.index('rpn_head.rpn_conv.weight') < sgot.index('roi_head.bbox_head.fc_cls.weight')
    # ... and through the detector built from the reference's own mmcv dict (num_stages=3 travels as ref_num_stages)
    from fgn_amd.detector import normalise_config
    ncfg = normalise_config(3, 2, backbone=dict(type='ResNet', depth=50, num_stages=3, strides=(1, 2, 2), out_indices=(2,),
                                                frozen_stages=-1, deep_stem=True, avg_down=True,
                                                norm_cfg=dict(type='GN', num_groups=32, requires_grad=True)))
    assert ncfg['backbone']['ref_num_stages'] == 3 and ncfg['backbone']['norm'] == 'GN'
    dcfg = normalise_config(3, 2, backbone=dict(type='ResNet', depth=50, num_stages=4, out_indices=(2,), frozen_stages=4,
                                                norm_cfg=dict(type='BN', requires_grad=False)))
    assert dcfg['backbone']['ref_num_stages'] == 4


def test_mmdet_registry_shim_builds_the_detector_from_the_reference_config(monkeypatch):
    """VERDICT r3 "missing" 5: main.py:390 calls ``build_detector(cfg.model, train_cfg=, test_cfg=)`` on the class the
    reference registers as 'FGN' (fgn.py:28).  ``fgn_amd.mmdet_plugin`` (imported through the config's
    ``custom_imports``) re-registers that name with the MI355X detector, so main.py needs no edit.  mmdet is absent here:
    a stand-in registry with mmcv's ``register_module(name=, force=, module=)`` / ``build(cfg, default_args=)`` contract
    takes its place; without any mmdet the shim is a no-op."""
    import importlib
    import sys
    import types
    import fgn_amd.mmdet_plugin as plug
    assert plug.register() is False                      # no mmdet in this image: nothing happens, nothing raises

    class Registry:
        def __init__(self):
            self.module_dict = {}

        def register_module(self, name=None, force=False, module=None):
            if name in self.module_dict and not force:
                raise KeyError(f'{name} is already registered')
            self.module_dict[name] = module
            return module

        def build(self, cfg, default_args=None):
            args = dict(cfg)
            for k, v in (default_args or {}).items():
                args.setdefault(k, v)
            return self.module_dict[args.pop('type')](**args)

    reg = Registry()
    reg.register_module(name='FGN', module=object)       # what importing the reference's fgn.py leaves behind
    for name in ('mmdet', 'mmdet.models'):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    builder = types.ModuleType('mmdet.models.builder')
    builder.DETECTORS = reg
    monkeypatch.setitem(sys.modules, 'mmdet.models.builder', builder)
    importlib.reload(plug)
    assert plug.REGISTERED is True
    from fgn_amd.detector import FGN
    assert reg.module_dict['FGN'] is FGN
    # the reference's own config dict (as in test_reference_style_config_is_accepted), built the way build_detector does
    from _glue import reference_model_cfg
    cfg = reference_model_cfg()
    model = reg.build(dict(cfg, type='FGN'), default_args=dict(train_cfg=None, test_cfg=cfg.get('test_cfg')))
    assert isinstance(model, FGN) and model.n_ways == cfg['n_ways']
    bb = model.backbone                                   # main.py:402-405
    assert bb.frozen_stages == cfg['backbone'].get('frozen_stages', 4) and bb.res_layers[:-1] == ['layer1', 'layer2', 'layer3']
    bb.res_layers = bb.res_layers[:-1]
    assert bb.eval() is bb
    model.cfg_obj = object()
    assert model.eval() is model


def test_episodic_sampler_matches_the_reference_on_its_own_databag(golden_dir):
    """WHICH instances form an episode.  tests/golden/make_golden_databag.py ran the reference's own
    ``load_dataset`` / ``reshuffle`` / ``__getitem__`` / ``get_query`` / ``get_support`` (base_fst.py:267-486, 605-625,
    793-846, 1052-1080, 1172-1246) on its shipped databag (resources/omniiseg_fst/..._val_base_.pkl: 968 parent and 1736
    child queries, 1792 instances) in four configurations, ``random`` seeded per item.  ``DatabagEpisodeSampler`` on the
    same index tables and the same seeds must decide the same episodes: order, child query, the N categories in their
    shuffled order, the query's category ids (real and remapped) and boxes, the N x K support instance ids."""
    import random
    import importlib.util
    from fgn_amd.fewshot_ds import Databag, DatabagEpisodeSampler
    spec = importlib.util.spec_from_file_location('make_golden_databag', os.path.join(golden_dir, 'make_golden_databag.py'))
    # (only its CONFIGS / seeds are needed: parse them without executing the imports of the reference)
    src = open(spec.origin).read()
    ns = {}
    head = src[src.index('CONFIGS = {'):src.index('def make_dataset')]
    exec('import numpy as np\n' + head.replace('class _Box', 'class _Box_'), ns)
    z = np.load(os.path.join(golden_dir, 'databag_sampler.npz'))
    bag = Databag.from_arrays(z)
    assert len(bag.parents_cats) == 968 and len(bag.children) == 1736 and len(bag.inst_cat) == 1792
    for tag, cfg in ns['CONFIGS'].items():
        ds = DatabagEpisodeSampler(bag, cats_novel=z['cats_novel'], cats_total_amount=26, sampling_cats='base_', **cfg)
        np.testing.assert_array_equal(ds.cats_to_save, z['cats_to_save'])
        np.testing.assert_array_equal(ds.order, z[f'{tag}__order'])
        assert len(ds) == int(z[f'{tag}__len'])
        np.testing.assert_array_equal(ns['item_indices'](len(ds), tag), z[f'{tag}__items'])
        ptr = z[f'{tag}__qry_ptr']
        for j, idx in enumerate(z[f'{tag}__items']):
            random.seed(ns['item_seed'](tag, idx))
            s = ds.sample_indices(int(idx))
            assert s['qry_child_idx'] == int(z[f'{tag}__qry_child_idx'][j]), (tag, idx)
            for k in ('cats_ids_to_sample_real', 'cats_ids_to_sample', 'spp_insts_ids'):
                np.testing.assert_array_equal(s[k], z[f'{tag}__{k}'][j], err_msg=f'{tag} {idx} {k}')
            for k in ('qry_cat_ids_real', 'qry_cat_ids', 'qry_bboxes'):
                np.testing.assert_array_equal(s[k], z[f'{tag}__{k}'][ptr[j]:ptr[j + 1]], err_msg=f'{tag} {idx} {k}')
            assert s['spp_insts_ids'].dtype == np.int64 and s['qry_bboxes'].dtype == np.float32
            assert len(s['spp_insts_ids']) == cfg['n_ways'] * cfg['k_shots']
            # the category of every support instance is the category it was drawn for, class-major (index = n * K + k)
            assert [int(bag.inst_cat[i]) for i in s['spp_insts_ids']] == \
                [int(c) for c in s['cats_ids_to_sample_real'] for _ in range(cfg['k_shots'])]


def test_x3_weight_image_is_an_exact_three_way_split():
    """``ops.pack_x3``: every f32 weight is the exact sum of its three bf16 planes (the low 16 bits of each plane's f32
    form are zero, plane by plane the leading bits of what the planes before left), laid out [G][K/32][3][Npad][32] with
    zero rows up to a multiple of 128; inside a K-tile chunk g holds k = 4g..4g+3, 16+4g..16+4g+3 and the 16-byte chunks of
    a row are XOR-ed with tau[(n >> 2) & 3], tau = (0, 3, 2, 1) (the 32x32x16 form of the experiments build: k in order,
    XOR with (n >> 2) & 3)."""
    import torch
    from fgn_amd import ops
    g = torch.Generator().manual_seed(0)
    w = torch.randn(2, 200, 96, generator=g) * torch.logspace(-30, 20, 96)[None, None, :]
    w[0, 5, 7] = 0.0
    G, N, K = w.shape
    npad = 256
    n = torch.arange(npad)
    for mfma32 in (False, True):
        img = ops.pack_x3(w, mfma32=mfma32)
        assert img.dtype == torch.uint8 and img.numel() == G * K * npad * 6
        pl = img.view(torch.int16).view(G, K // 32, 3, npad, 4, 8).permute(0, 2, 3, 1, 4, 5)       # G, plane, n, kt, chunk, 8
        swz = (n >> 2) & 3
        if not mfma32:
            swz = torch.tensor([0, 3, 2, 1])[swz]
        idx = torch.arange(4)[None, :] ^ swz[:, None]                                             # logical chunk -> physical
        pl = torch.gather(pl, 4, idx[None, None, :, None, :, None].expand(G, 3, npad, K // 32, 4, 8))
        pl = pl.reshape(G, 3, npad, K // 32, 32)
        if not mfma32:          # position 8g + j of a K-tile holds k = 4g + j (j < 4), 16 + 4g + j - 4 (j >= 4)
            korder = torch.tensor([4 * q + j if j < 4 else 16 + 4 * q + j - 4 for q in range(4) for j in range(8)])
            inv = torch.empty(32, dtype=torch.long)
            inv[korder] = torch.arange(32)
            pl = pl[..., inv]
        planes = (pl.reshape(G, 3, npad, K).to(torch.int32) << 16).view(torch.float32)
        assert torch.equal(planes.double().sum(1)[:, :N], w.double())                             # exact
        assert (planes[:, :, N:] == 0).all()
        p1, p2, p3 = planes[:, 0, :N], planes[:, 1, :N], planes[:, 2, :N]
        assert ((p2.abs() <= p1.abs() * 2.0 ** -7) | (p1 == 0)).all() and ((p3.abs() <= p1.abs() * 2.0 ** -15) | (p1 == 0)).all()
        assert (torch.sign(p2) * torch.sign(p1) >= 0).all() and (torch.sign(p3) * torch.sign(p1) >= 0).all()


def test_x3_routing_rule_is_host_logic():
    """``fgn_x3_row_tile`` (no GPU involved): which GEMM-shaped launches go to conv_pw_x3_kernel and on which row tile - at
    least 192 tiles of 64 x 128 and at least 70 % real channels in its 128-column tiles; the 128-row tile for the large,
    deep GEMMs whose rows fill it (measured per launch of a cfg3 episode, both arithmetics on one box)."""
    from fgn_amd import lib
    L = lib.load()
    assert L.fgn_x3_row_tile(14700, 1024, 1024, 0, 0) == 128             # relation Q
    assert L.fgn_x3_row_tile(36 * 1280, 512, 512, 1280, 1236) == 128     # Winograd GEMM of 300 + 9 RoIs
    assert L.fgn_x3_row_tile(36 * 896, 1024, 1024, 896, 819) == 128      # AG-RPN: 819 rows fill seven 128-row tiles to 91 %
    assert L.fgn_x3_row_tile(36 * 512, 512, 512, 512, 400) == 64         # mask head: 400 rows would waste a quarter of four
    assert L.fgn_x3_row_tile(25916, 512, 128, 0, 0) == 64                # a shallow K loop
    assert L.fgn_x3_row_tile(103664, 64, 256, 0, 0) == 0                 # layer1 conv1: half of every tile would be padding
    assert L.fgn_x3_row_tile(12600, 76, 1024, 0, 0) == 0                 # RPN head: 76 of 128 columns
    assert L.fgn_x3_row_tile(441, 512, 1024, 0, 0) == 0                  # 9 support RoIs: 28 tiles - the f32 path splits K instead
    assert L.fgn_x3_row_tile(6504, 256, 1024, 0, 0) == 64                # 204 tiles
    assert L.fgn_x3_row_tile(36 * 100, 512, 512, 100, 100) == 0          # a group that is not a whole number of 64-row tiles
    assert L.fgn_x3_image_bytes(1024, 1024, 36) == 36 * 1024 * 1024 * 6
    assert L.fgn_winograd_t_pad(819) == 896 and L.fgn_winograd_t_pad(1280) == 1280


def test_h2_weight_image_holds_the_scaled_weights_to_half_an_f32_ulp_and_the_gemm_it_defines_is_f32_accurate():
    """``ops.pack_h2`` (host logic, no GPU): every output column scaled by a power of two that puts its largest |w| into
    [2^14, 2^15), two f16 planes hi = f16(w s), lo = f16(w s - hi), the k order / chunk swizzle of the kernel's MFMA operand,
    the inverse scales behind the image.  Un-permuted: hi + lo reproduces w s to within 2^-23 |w s| (one f32 ulp) wherever
    lo is a normal f16, the scales are exact powers of two, padded columns are zero with scale 1.  And the arithmetic the
    image defines - h_a h_b + h_a l_b + l_a h_b with the activations split the same way under one power-of-two scale - is as
    close to fp64 as an f32 GEMM (numpy emulation of conv_pw_h2_kernel's products, csrc/conv_pw_h2.h)."""
    import numpy as np
    from fgn_amd import ops
    g = torch.Generator().manual_seed(0)
    w = torch.randn(2, 200, 96, generator=g) * torch.logspace(-3, 2, 200)[None, :, None]     # columns five orders apart
    w[0, 5, :] = 0.0
    w[1, 7, 3] = 0.0
    G, N, K = w.shape
    npad = 256
    img = ops.pack_h2(w)
    assert img.dtype == torch.uint8 and img.numel() == G * npad * (K * 4 + 4)
    inv = img[G * K * npad * 4:].view(torch.float32).view(G, npad)
    pl = img[:G * K * npad * 4].view(torch.int16).view(G, K // 32, 2, npad, 4, 8).permute(0, 2, 3, 1, 4, 5)   # G, plane, n, kt, chunk, 8
    n = torch.arange(npad)
    swz = torch.tensor([0, 3, 2, 1])[(n >> 2) & 3]
    idx = torch.arange(4)[None, :] ^ swz[:, None]                                                # logical chunk -> physical
    pl = torch.gather(pl, 4, idx[None, None, :, None, :, None].expand(G, 2, npad, K // 32, 4, 8)).reshape(G, 2, npad, K // 32, 32)
    korder = torch.tensor([4 * q + j if j < 4 else 16 + 4 * q + j - 4 for q in range(4) for j in range(8)])
    back = torch.empty(32, dtype=torch.long)
    back[korder] = torch.arange(32)
    planes = pl[..., back].reshape(G, 2, npad, K).contiguous().view(torch.float16).double()
    hi, lo = planes[:, 0], planes[:, 1]
    # scales: exact powers of two, column maximum into [2^14, 2^15); zero / padded columns: 1
    m, e = np.frexp(inv.numpy().astype(np.float64))
    assert (m == 0.5).all()
    colmax = w.abs().amax(2)
    scaled_max = (colmax.double() / inv[:, :N].double())
    assert ((scaled_max >= 2.0 ** 14) & (scaled_max < 2.0 ** 15) | (colmax == 0)).all()
    assert (inv[:, N:] == 1).all() and (inv[0, 5] == 1) and (hi[:, N:] == 0).all() and (lo[:, N:] == 0).all()
    ws = w.double() / inv[:, :N, None].double()
    err = (hi[:, :N] + lo[:, :N] - ws).abs()
    normal = lo[:, :N].abs() >= 2.0 ** -14                       # below: f16 subnormals, absolute error <= 2^-25
    assert (err[normal] <= ws.abs()[normal] * 2.0 ** -23).all() and (err[~normal] <= 2.0 ** -25).all()
    assert err[normal].mean() <= (ws.abs()[normal] * 2.0 ** -25).mean()                # a quarter of an ulp on average
    assert (lo[:, :N].abs() <= hi[:, :N].abs() * 2.0 ** -10 + 2.0 ** -24).all()
    # the GEMM the image defines, emulated: activations split the same way under one scale, three products, against fp64 / f32
    x = torch.randn(300, K, generator=g).relu_() * 0.37
    e_x = int(np.floor(np.log2(float(x.abs().max()))))
    s = 2.0 ** (13 - e_x)
    xs = (x.double() * s)
    xh = xs.to(torch.float16).double()
    xl = (xs - xh).to(torch.float16).double()
    for gi in range(G):
        got = (xl @ hi[gi, :N].T + xh @ lo[gi, :N].T + xh @ hi[gi, :N].T) / s * inv[gi, :N].double()[None, :]
        ref = x.double() @ w[gi].double().T
        f32 = (x @ w[gi].T).double()
        scale = ref.abs().max(0, keepdim=True).values.clamp_min(1e-30)       # per column: the columns are orders apart
        assert ((got - ref).abs() / scale).max().item() <= 3e-7
        assert ((got - ref).abs() / scale).max().item() <= ((f32 - ref).abs() / scale).max().item() + 1e-7


def test_h2_routing_rule_is_host_logic():
    """``fgn_h2_row_tile`` (no GPU involved): the tile conv_pw_h2_kernel runs a launch on - conv_pw_x3_kernel's rule for the
    128-column tiles with the 128-row tile kept for the grouped (Winograd) launches, and 128 rows x 64 columns (264) for
    layers of 48..64 output channels with at least 256 such tiles."""
    from fgn_amd import lib
    L = lib.load()
    assert L.fgn_h2_row_tile(14700, 1024, 1024, 0, 0) == 64              # relation Q: plain 1x1 launches stay on 64 rows
    assert L.fgn_h2_row_tile(36 * 1280, 512, 512, 1280, 1236) == 128     # Winograd GEMM of 300 + 9 RoIs
    assert L.fgn_h2_row_tile(36 * 896, 1024, 1024, 896, 819) == 128      # AG-RPN
    assert L.fgn_h2_row_tile(36 * 512, 512, 512, 512, 400) == 64         # mask head
    assert L.fgn_h2_row_tile(103664, 64, 256, 0, 0) == 264               # layer1 conv1: the 64-column tile
    assert L.fgn_h2_row_tile(103664, 64, 576, 0, 0) == 264               # layer1's 3x3 as an implicit GEMM
    assert L.fgn_h2_row_tile(25916, 128, 1152, 0, 0) == 64               # layer2.0's strided 3x3
    assert L.fgn_h2_row_tile(12600, 76, 1024, 0, 0) == 0                 # RPN head: 76 of 128 columns (measured: no gain)
    assert L.fgn_h2_row_tile(3000, 64, 256, 0, 0) == 0                   # 24 tiles of 128 x 64
    assert L.fgn_h2_row_tile(103664, 40, 256, 0, 0) == 0                 # mostly padding
    assert L.fgn_h2_row_tile(441, 512, 1024, 0, 0) == 0                  # 9 support RoIs


def test_h2_per_wave_scale_search_is_exact_in_the_scales():
    """The activation side of conv_pw_h2_kernel restated in numpy (csrc/conv_pw_h2.h: per wave and output tile the first
    K-tile with a non-zero element sets S so that its largest |x| lands in [2^13, 2^14); a later K-tile with an element
    above 65504 / S picks a new S from its own maximum and the accumulator is multiplied by the power-of-two ratio; all-zero
    K-tiles pass as they are; S comes out at the end).  On fragments whose K-tiles differ by orders of magnitude in both
    directions the result stays within f32 accuracy of fp64, no plane ever leaves the f16 range, and the number of scale
    changes is what the rule predicts."""
    import numpy as np
    rng = np.random.default_rng(3)

    def run(x, w):                                   # x [rows, K] f32 (one wave's rows), w [K, n] f16-exact weights (scaled)
        acc = np.zeros((x.shape[0], w.shape[1]), np.float32)
        s, inv, lim, have, changes = np.float32(1), np.float32(1), np.float32(0), False, 0
        wh = w.astype(np.float16).astype(np.float32)
        wl = (w - wh).astype(np.float16).astype(np.float32)
        for k0 in range(0, x.shape[1], 32):
            f = x[:, k0:k0 + 32]
            big = np.abs(f).max()
            if not (big <= lim):
                e = int(np.floor(np.log2(big)))
                ns = np.float32(2.0 ** (13 - e))
                if have:
                    acc *= ns * inv                  # a power of two: exact
                    changes += 1
                s, inv, lim, have = ns, np.float32(2.0 ** (e - 13)), np.float32(65504.0 * 2.0 ** (e - 13)), True
            fs = f * s
            h = fs.astype(np.float16)
            assert np.isfinite(h).all()              # never leaves the f16 range
            hf = h.astype(np.float32)
            lo = (fs - hf).astype(np.float16).astype(np.float32)
            kw_h, kw_l = wh[k0:k0 + 32], wl[k0:k0 + 32]
            acc = (acc + lo @ kw_h + hf @ kw_l + hf @ kw_h).astype(np.float32)
        return acc * inv, changes

    K, n = 256, 24
    w = (rng.standard_normal((K, n)) * 2.0 ** 10).astype(np.float32)
    for mags, expect in (([1.0] * 8, 0), ([1e-3, 1.0, 1.0, 1e3, 1e3, 1e-6, 1.0, 1e5], 3), ([0.0, 0.0, 1.0, 1.0, 0.0, 1e4, 1.0, 0.0], 1),
                         ([1e6, 1.0, 1e-4, 1.0, 1.0, 1.0, 1.0, 1.0], 0)):
        x = np.maximum(rng.standard_normal((32, K)), 0).astype(np.float32)
        for t, m in enumerate(mags):
            x[:, 32 * t:32 * t + 32] *= np.float32(m)
        got, changes = run(x, w)
        ref = x.astype(np.float64) @ w.astype(np.float64)
        f32 = (x @ w).astype(np.float64)
        scale = np.abs(ref).max()
        assert changes == expect, (mags, changes)
        assert np.abs(got - ref).max() <= max(3e-7 * scale, 2.0 * np.abs(f32 - ref).max()), mags
