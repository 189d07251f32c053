"""``FGN.forward_train`` on the HIP path (reference: subprojects/sp02_omniiseg_fgn_mmdet/fgn.py:125-185).

Returns the reference's loss dict - ``loss_rpn_cls`` / ``loss_rpn_bbox`` (lists of one tensor, AG-RPN,
fgn_ag_rpn_head.py:58-79), ``loss_cls`` / ``ACC-Unbalanced`` / ``ACC-Balanced`` / ``loss_bbox`` (FGNBBoxHead.loss,
fgn_roi_head.py:58-118) and ``loss_mask`` (fgn_roi_head.py:384-417) - as FORWARD VALUES: the HIP path has no
autograd graph, so the tensors carry no ``grad_fn`` (the backward kernels of the heads are the next step of this row,
DESIGN.md section 8).  What runs where:

  * backbone (frozen, BatchNorm in eval mode: frozen_stages=4 / norm_eval=True, fgn_r50_c4_densecl.py:31-36 with
    fgn.py:67-77), AG-RPN convolutions, RoIAlign, relation head, mask head: the inference kernels
  * shared head: its BatchNorm layers are in TRAINING mode (norm_cfg BN requires_grad=True inside a module that is in
    train(), fgn_roi_head.py:202-238): raw convolutions + ``ops.bn_train`` (batch statistics, running update)
  * MaxIoUAssigner for anchors and proposals, box encoding, the five loss reductions, the 12000 -> 2000 proposal
    stage: ``csrc/train.hip`` / ``csrc/rpn_post.hip``
  * RandomSampler: the permutation is drawn like mmdet does, ``torch.randperm(n)`` on the CPU generator
    (my_random_sampler.py:58), and applied on the device; index bookkeeping (nonzero / gather / cat) is torch
    device plumbing and, like the reference's assign + sample loop, host-driven (one sync per sampled set).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import ops


class _SharedBlockTrain:
    """One Bottleneck of the shared head with train-mode BatchNorm: conv -> bn_train(+ReLU) x3, identity shortcut
    (stride 1, inplanes == planes * expansion: no downsample, fgn_roi_head.py:207-225)."""

    def __init__(self, sd: dict, prefix: str, winograd: int):
        self.conv1 = ops.pack_conv(sd[prefix + '.conv1.weight'])
        w2 = sd[prefix + '.conv2.weight']
        self.conv2 = ops.pack_conv(w2, pad=1)
        self.conv2_wg = ops.pack_winograd(w2, m=winograd) \
            if winograd and w2.shape[1] % 32 == 0 and w2.shape[0] % 4 == 0 else None
        self.conv3 = ops.pack_conv(sd[prefix + '.conv3.weight'])
        self.prefix = prefix
        self.bn = [{k: sd[f'{prefix}.bn{i}.{k}'].detach().float().clone() for k in
                    ('weight', 'bias', 'running_mean', 'running_var')} for i in (1, 2, 3)]

    def to(self, device):
        for l in (self.conv1, self.conv2, self.conv3, self.conv2_wg):
            if l is not None:
                l.to(device)
        self.bn = [{k: v.contiguous().to(device) for k, v in b.items()} for b in self.bn]
        return self

    def _bn(self, i, y, eps, momentum, relu, residual=None):
        b = self.bn[i]
        return ops.bn_train(y, b['weight'], b['bias'], eps, momentum, b['running_mean'], b['running_var'],
                            residual=residual, relu=relu)[0]

    def __call__(self, x, eps, momentum):
        y = self._bn(0, ops.conv2d(x, self.conv1), eps, momentum, True)
        wg = self.conv2_wg
        if wg is not None and ops.winograd_pays(y.shape[0], y.shape[1], y.shape[2], wg.cin, wg.cout, wg.m):
            y = ops.conv3x3_winograd(y, wg)
        else:
            y = ops.conv2d(y, self.conv2)
        y = self._bn(1, y, eps, momentum, True)
        return self._bn(2, ops.conv2d(y, self.conv3), eps, momentum, True, residual=x)   # relu(bn3(conv3) + identity)


def pack_train(model, device) -> None:
    """Raw (un-folded) shared-head layers and their BatchNorm parameters / running buffers on ``device``."""
    nb = model.cfg['roi_head']['shared_head']['num_blocks']
    model._PT = {'shared': [_SharedBlockTrain(model._sd, f'roi_head.shared_head.{b}', model.use_winograd).to(device)
                            for b in range(nb)],
                 'device': torch.device(device), 'anchors': {}}


def shared_head_train(model, x, momentum: float):
    eps = model.cfg['backbone']['bn_eps']
    for blk in model._PT['shared']:
        x = blk(x, eps, momentum)
    return x


def bn_buffers(model) -> dict:
    """The running statistics of the shared head as updated by ``forward_train`` (state_dict keys -> CPU tensors)."""
    out = {}
    for blk in model._PT['shared']:
        for i, b in enumerate(blk.bn):
            out[f'{blk.prefix}.bn{i + 1}.running_mean'] = b['running_mean'].detach().cpu()
            out[f'{blk.prefix}.bn{i + 1}.running_var'] = b['running_var'].detach().cpu()
    return out


# ------------------------------------------------------------------------------------------
def _anchors_for(model, fh: int, fw: int, img_hw, dev):
    """All anchors of the level [n,4] and their inside flags (AnchorGenerator.grid_priors / valid_flags +
    anchor_inside_flags, my_anchor_head.py:171-199, 233-236) for one image shape; host-built once, cached."""
    key = (fh, fw, int(img_hw[0]), int(img_hw[1]))
    hit = model._PT['anchors'].get(key)
    if hit is not None:
        return hit
    rp, tc = model.cfg['rpn_head'], model.cfg['train_cfg']['rpn']
    f = np.float32
    base = ops.base_anchors(rp['anchor_scales'], rp['anchor_ratios'], rp['anchor_stride'])
    stride = rp['anchor_stride']
    sx = (np.arange(fw, dtype=f) * f(stride)).astype(f)
    sy = (np.arange(fh, dtype=f) * f(stride)).astype(f)
    xx, yy = np.tile(sx, fh), np.repeat(sy, fw)
    shifts = np.stack([xx, yy, xx, yy], -1)
    anchors = (base[None] + shifts[:, None]).reshape(-1, 4).astype(f)
    ih, iw = key[2], key[3]
    vh, vw = min(int(np.ceil(ih / stride)), fh), min(int(np.ceil(iw / stride)), fw)
    valid = np.zeros((fh, fw), bool)
    valid[:vh, :vw] = True
    valid = np.repeat(valid.reshape(-1), base.shape[0])
    ab = tc['allowed_border']
    if ab >= 0:
        valid &= (anchors[:, 0] >= -ab) & (anchors[:, 1] >= -ab) & (anchors[:, 2] < iw + ab) & (anchors[:, 3] < ih + ab)
    hit = (torch.from_numpy(anchors).to(dev), torch.from_numpy(valid.astype(np.uint8)).to(dev), bool(valid.any()))
    model._PT['anchors'][key] = hit
    return hit


def _choose(cand: torch.Tensor, num: int, perm_fn) -> torch.Tensor:
    """RandomSampler._sample_pos/_sample_neg + the ``unique()`` of BaseSampler.sample: at most ``num`` of ``cand``
    (ascending indices), drawn with the CPU permutation mmdet draws (my_random_sampler.py:58-59)."""
    if cand.numel() > num:
        perm = perm_fn(cand.numel())[:num].to(cand.device)
        cand = torch.sort(cand[perm]).values
    return cand


def _sample(gt_inds: torch.Tensor, num: int, pos_fraction: float, perm_fn):
    pos = _choose(torch.nonzero(gt_inds > 0).view(-1), int(num * pos_fraction), perm_fn)
    neg = _choose(torch.nonzero(gt_inds == 0).view(-1), num - pos.numel(), perm_fn)
    return pos, neg


def _zero(dev):
    return torch.zeros((), device=dev, dtype=torch.float32)


def forward_train(model, qry_img, qry_bboxes, qry_cat_ids, qry_isegmaps, qry_bboxes_ignore=None, proposals=None,
                  spp_imgs=None, spp_bboxes=None, spp_isegmaps=None, img_shape=None, perm_fn=torch.randperm,
                  bn_momentum: float = 0.1, **kwargs) -> dict:
    if not torch.cuda.is_available():
        raise ops._lib.FgnHipError('FGN.forward_train needs a GPU: the HIP path has no CPU fallback')
    if qry_bboxes_ignore is not None and any(b is not None for b in qry_bboxes_ignore):
        raise NotImplementedError('qry_bboxes_ignore: the reference configures ignore_iof_thr=-1 (never used)')
    dev = torch.device('cuda', torch.cuda.current_device())
    if model._packed_device != dev:
        model._pack(dev)
    if getattr(model, '_PT', None) is None or model._PT['device'] != dev:
        pack_train(model, dev)
    P, cfg = model._P, model.cfg
    N, K = model.n_ways, model.k_shots
    tcfg = cfg['train_cfg']
    rh, rp = cfg['roi_head'], cfg['rpn_head']
    tr = model.debug_trace
    B = qry_img.shape[0]
    main = torch.cuda.current_stream()

    # ---- modify_input (fgn.py:79-108): H2D, YXYX -> XYXY on private copies
    qry = qry_img.to(dev, torch.float32, non_blocking=True)
    gt_xyxy = [b.to(dev, torch.float32)[:, [1, 0, 3, 2]].contiguous() for b in qry_bboxes]
    cat_ids = [torch.as_tensor(c).to(dev, torch.int64) for c in qry_cat_ids]
    ih, iw = int(img_shape[0][0]), int(img_shape[0][1])
    if any(int(s[0]) != ih or int(s[1]) != iw for s in img_shape):
        raise ValueError('all images of a batch must share img_shape (the dataset batches by size)')

    # ---- backbone passes (frozen) and the AG-RPN on the N guided maps of every image
    sc = model._support_front(spp_imgs, spp_bboxes, spp_isegmaps, B, dev, main)
    qry_fmap = model.extract_feat(qry)
    fh, fw, C = qry_fmap.shape[1:]
    wg = P['rpn_conv_wg']
    if wg is not None and ops.winograd_fits(B * N, fh, fw, C, wg.cout, wg.m):
        x = ops.conv3x3_winograd(qry_fmap, wg, in_scale=sc['vec'], a_img_div=N)
    else:
        x = ops.conv2d(ops.scale_channels(qry_fmap, sc['vec'], N), P['rpn_conv'])
    head = ops.conv2d(x, P['rpn_head'])                                     # [B*N,h,w,>=5A]: A logits | 4A deltas
    A = P['anchors'].shape[0]
    G, n_total = B * N, fh * fw * A

    # ---- AG-RPN loss (fgn_ag_rpn_head.py:58-79 -> RPNHead.loss -> my_anchor_head.py:201-520)
    tc = tcfg['rpn']
    anchors, inside, any_inside = _anchors_for(model, fh, fw, (ih, iw), dev)
    if not any_inside:
        raise ValueError('no anchor lies inside the image: the reference returns no RPN loss here '
                         '(my_anchor_head.py:237-238) and fails')
    flat = head.reshape(G, fh * fw, head.shape[-1])
    logits_all = flat[:, :, :A].reshape(G, n_total)
    deltas_all = flat[:, :, A:5 * A].reshape(G, n_total, 4)
    xs, ys, preds, tgts = [], [], [], []
    n_pos_total = n_neg_total = 0
    rpn_sets = []
    for g in range(G):
        i, j = divmod(g, N)
        gts = gt_xyxy[i][torch.nonzero(cat_ids[i] == j).view(-1)]          # per-class GT list, image-major order
        gi = ops.box_assign(anchors, gts, tc['pos_iou_thr'], tc['neg_iou_thr'], tc['min_pos_iou'],
                            tc['match_low_quality'], inside=inside)
        pos, neg = _sample(gi, tc['num'], tc['pos_fraction'], perm_fn)
        rpn_sets.append((pos, neg))
        n_pos_total += max(pos.numel(), 1)
        n_neg_total += max(neg.numel(), 1)
        xs += [logits_all[g][pos], logits_all[g][neg]]
        ys += [torch.ones(pos.numel(), device=dev), torch.zeros(neg.numel(), device=dev)]
        if pos.numel():
            preds.append(deltas_all[g][pos])
            tgts.append(ops.bbox2delta(anchors[pos].contiguous(), gts[(gi[pos] - 1).long()].contiguous(),
                                       rp['target_means'], rp['target_stds']))
    n_samples = n_pos_total + n_neg_total
    pw = 1.0 if tc['pos_weight'] <= 0 else float(tc['pos_weight'])
    x_cat, y_cat = torch.cat(xs).contiguous(), torch.cat(ys).contiguous()
    w_cat = None if pw == 1.0 else torch.where(y_cat > 0, pw, 1.0).float().contiguous()
    loss_rpn_cls = ops.bce_logits_sum(x_cat, y_cat, w_cat, n_samples) / N              # the 1/N balancer
    loss_rpn_bbox = (ops.smooth_l1_sum(torch.cat(preds).contiguous(), torch.cat(tgts).contiguous(), None, n_samples)
                     if preds else torch.zeros(1, device=dev)) / N
    losses = {'loss_rpn_cls': [loss_rpn_cls.view(())], 'loss_rpn_bbox': [loss_rpn_bbox.view(())]}
    if tr is not None:
        tr.update(qry_fmap=qry_fmap, rpn_head=head, rpn_sets=rpn_sets, rpn_num_total_samples=n_samples)

    # ---- proposals with train_cfg.rpn_proposal (fgn.py:161-167)
    if proposals is None:
        pc = tcfg['rpn_proposal']
        _, scores, deltas = ops.rpn_merge(head, B, N, A)
        props, n_props = ops.rpn_proposals(scores, deltas, P['anchors'], fh, fw, rp['anchor_stride'], ih, iw,
                                           rp['target_means'], rp['target_stds'], pc['nms_pre'], pc['min_bbox_size'],
                                           pc['nms_iou_threshold'], pc['max_per_img'])
        counts = n_props.tolist()
        proposals = [props[i, :counts[i]] for i in range(B)]
    else:
        proposals = [torch.as_tensor(p).to(dev, torch.float32).contiguous() for p in proposals]
    if tr is not None:
        tr['proposals'] = proposals

    # ---- FGNRoIHead.forward_train (fgn_roi_head.py:451-529): assign + sample per image
    rc = tcfg['rcnn']
    samples = []
    for i in range(B):
        pr, gts = proposals[i], gt_xyxy[i]
        k = gts.shape[0]
        gi = ops.box_assign(pr, gts, rc['pos_iou_thr'], rc['neg_iou_thr'], rc['min_pos_iou'],
                            rc['match_low_quality']) if pr.shape[0] else \
            torch.zeros((0,), device=dev, dtype=torch.int32)
        boxes = pr[:, :4]
        if rc['add_gt_as_proposals'] and k > 0:           # BaseSampler.sample: GT boxes in front, assigned to themselves
            boxes = torch.cat([gts, boxes])
            gi = torch.cat([torch.arange(1, k + 1, device=dev, dtype=torch.int32), gi])
        pos, neg = _sample(gi, rc['num'], rc['pos_fraction'], perm_fn)
        assigned = (gi[pos] - 1).long()
        samples.append(dict(pos_bboxes=boxes[pos], neg_bboxes=boxes[neg], pos_assigned_gt_inds=assigned,
                            pos_gt_bboxes=gts[assigned] if k else gts.view(-1, 4)[:0],
                            pos_gt_labels=cat_ids[i][assigned] if k else cat_ids[i][:0], pos_inds=pos, neg_inds=neg))
    if tr is not None:
        tr['samples'] = samples

    # count_spp with the shared head in training mode (fgn_roi_head.py:491, 419-449)
    model._support_back(sc, B, dev, shared=lambda t: shared_head_train(model, t, bn_momentum))

    # _bbox_forward_train (fgn_roi_head.py:344-358)
    rois = torch.cat([torch.cat([torch.full((len(s['pos_bboxes']) + len(s['neg_bboxes']), 1), float(i), device=dev),
                                 torch.cat([s['pos_bboxes'], s['neg_bboxes']])], 1) for i, s in enumerate(samples)])
    rois = rois.contiguous()
    n_rois = rois.shape[0]
    rel, bh = rh['relation'], rh['bbox_head']
    if n_rois:
        xr = ops.roi_align(qry_fmap, rois, rh['roi_out_size'], 1.0 / rh['featmap_stride'], rh['roi_sampling_ratio'],
                           True)
        feats = shared_head_train(model, xr, bn_momentum)
        Q = ops.conv2d(feats, P['rel_q'])
        cls_raw, reg_raw = ops.relation_gn_head(Q, sc['S'], rois, P['gn_w'], P['gn_b'], P['fc_w'], P['fc_b'], N,
                                                rel['gn_groups'], rel['gn_eps'])
        # count_modified_cls_bbox (fgn_roi_head.py:302-326)
        if N == 1:
            cls_score, bbox_pred = cls_raw[:, [1, 0]].contiguous(), reg_raw
        else:
            resh = cls_raw.view(n_rois, 2 * N)
            fg = resh[:, 1::2]
            bg = resh.gather(1, (fg.argmax(dim=1) * 2)[:, None])
            cls_score = torch.cat([fg, bg], 1).contiguous()
            bbox_pred = reg_raw.view(n_rois, 4 * N)
    else:
        feats = torch.zeros((0, rh['roi_out_size'], rh['roi_out_size'], C), device=dev)
        cls_score = torch.zeros((0, N + 1), device=dev)
        bbox_pred = torch.zeros((0, 4 * N), device=dev)
    # FGNBBoxHead.get_targets / loss (fgn_roi_head.py:58-160): background label = n_ways
    labels = torch.cat([torch.cat([s['pos_gt_labels'], torch.full((len(s['neg_bboxes']),), N, device=dev,
                                                                  dtype=torch.int64)]) for s in samples]).contiguous()
    pos_rows = torch.nonzero(labels < N).view(-1)
    pw = 1.0 if rc['pos_weight'] <= 0 else float(rc['pos_weight'])
    lw = None if pw == 1.0 else torch.where(labels < N, pw, 1.0).float().contiguous()
    avg = max(float(n_rois), 1.0)                     # every sampled RoI has label weight > 0
    if n_rois:
        losses['loss_cls'] = ops.softmax_ce_sum(cls_score, labels, lw, avg).view(())
        pred_h, lab_h = cls_score.argmax(dim=-1).cpu().numpy(), labels.cpu().numpy()     # get_accuracy: on the host
        acc = float((pred_h == lab_h).mean())
        bal = float(np.mean([(pred_h[lab_h == c] == c).mean() for c in np.unique(lab_h)]))
        losses['ACC-Unbalanced'], losses['ACC-Balanced'] = torch.Tensor([acc]), torch.Tensor([bal])
    if pos_rows.numel():
        pos_pred = bbox_pred.view(n_rois, -1, 4)[pos_rows, labels[pos_rows]].contiguous()
        pos_tgt = ops.bbox2delta(torch.cat([s['pos_bboxes'] for s in samples]).contiguous(),
                                 torch.cat([s['pos_gt_bboxes'] for s in samples]).contiguous(),
                                 bh['target_means'], bh['target_stds'])
        losses['loss_bbox'] = ops.smooth_l1_sum(pos_pred, pos_tgt, None, float(n_rois)).view(())
    else:
        losses['loss_bbox'] = _zero(dev)
    if tr is not None:
        tr.update(rois=rois, bbox_feats=feats, cls_score=cls_score, bbox_pred=bbox_pred, labels=labels)

    # ---- mask branch (fgn_roi_head.py:498-527, 384-417): shared RoI extractor -> the positives' bbox_feats
    n_pos = int(pos_rows.numel())
    if n_pos:
        img_of = rois[pos_rows, 0].long()
        vmask = sc['cat_mean_mp'][labels[pos_rows] + N * img_of].contiguous()          # spp_vecs_mask
        mlog, _ = model._mask_head(feats[pos_rows].contiguous(), vmask)
        # mask_target_single + BitmapMasks.crop_and_resize: RoIAlign(aligned, adaptive grid) of the GT bitmaps
        gt_masks = []
        for m in qry_isegmaps:
            m = torch.as_tensor(m).to(dev)
            gt_masks.append((m if m.dtype in (torch.bool, torch.uint8) else (m != 0)).to(torch.uint8))
        first = np.cumsum([0] + [m.shape[0] for m in gt_masks[:-1]])
        masks_all = torch.cat(gt_masks).contiguous()
        mh_, mw_ = masks_all.shape[-2:]
        pb = torch.cat([s['pos_bboxes'] for s in samples])
        pb = torch.stack([pb[:, 0].clamp(0, mw_), pb[:, 1].clamp(0, mh_), pb[:, 2].clamp(0, mw_),
                          pb[:, 3].clamp(0, mh_)], 1)
        gidx = torch.cat([s['pos_assigned_gt_inds'] + int(first[i]) for i, s in enumerate(samples)])
        mrois = torch.cat([gidx.float()[:, None], pb], 1).contiguous()
        ms = rc['mask_size']
        tgt = ops.roi_align_mask(masks_all, mrois, ms, 1.0, 0, True)                  # [n_pos, ms, ms] in [0,1]
        if tuple(mlog.shape[1:]) != (ms, ms):
            raise ValueError(f'mask head output {tuple(mlog.shape[1:])} != train_cfg mask_size {ms}')
        losses['loss_mask'] = ops.bce_logits_sum(mlog.contiguous(), tgt, None, float(mlog.numel()), y_threshold=0.5)
        if tr is not None:
            tr.update(mask_pred=mlog, mask_targets_soft=tgt)
    else:
        losses['loss_mask'] = _zero(dev)
    return losses
