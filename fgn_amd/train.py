"""``FGN.forward_train`` and the training step on the HIP path (reference: subprojects/sp02_omniiseg_fgn_mmdet/
fgn.py:125-185; the loop around it is mmcv's runner + OptimizerHook + torch.optim.Adagrad, fgn_train_schedule.py).

``forward_train`` returns the reference's loss dict - ``loss_rpn_cls`` / ``loss_rpn_bbox`` (lists of one tensor, AG-RPN,
fgn_ag_rpn_head.py:58-79), ``loss_cls`` / ``ACC-Unbalanced`` / ``ACC-Balanced`` / ``loss_bbox`` (FGNBBoxHead.loss,
fgn_roi_head.py:58-118) and ``loss_mask`` (fgn_roi_head.py:384-417).  The HIP path has no autograd graph, so the
tensors carry no ``grad_fn``; ``Trainer`` keeps a tape of the forward pass instead and runs the backward pass of the
trainable heads, the Adagrad update and the re-pack of the kernel layouts itself.  What runs where:

  * backbone (frozen, BatchNorm in eval mode: frozen_stages=4 / norm_eval=True, fgn_r50_c4_densecl.py:31-36 with
    fgn.py:67-77), AG-RPN convolutions, RoIAlign, relation head, mask head: the inference kernels
  * shared head: its BatchNorm layers are in TRAINING mode (norm_cfg BN requires_grad=True inside a module that is in
    train(), fgn_roi_head.py:202-238): raw convolutions + ``ops.bn_train`` (batch statistics, running update)
  * MaxIoUAssigner for anchors and proposals, box encoding, the five loss reductions, the 12000 -> 2000 proposal
    stage: ``csrc/train.hip`` / ``csrc/rpn_post.hip``
  * RandomSampler: both stages' assignment vectors are copied to the host once per step; candidate lists, the
    permutation mmdet draws (``torch.randperm(n)`` on the CPU generator, my_random_sampler.py:58) and the label /
    target gathers run there in numpy, the selected indices go back as small index tensors
  * backward: loss gradients, BatchNorm(train) / relation-GroupNorm / mask-logit backward, im2col, column sums, the
    weight-gradient GEMM (``fgn_gemm_tn_f32``, fp32 MFMA), Adagrad: ``csrc/train_bwd.hip``; data gradients: the forward
    convolution kernel with transposed (1x1) / flipped (3x3) weights; the 6-row fc products and the 75-channel
    AG-RPN head (shapes the MFMA kernels do not take) on ``fgn_gemm_small_f32`` - no rocBLAS call in the step
"""
from __future__ import annotations

import os
import threading
from typing import Optional

import numpy as np
import torch

from . import ops


class _SharedBlockTrain:
    """One Bottleneck of the shared head with train-mode BatchNorm: conv -> bn_train(+ReLU) x3, identity shortcut
    (stride 1, inplanes == planes * expansion: no downsample, fgn_roi_head.py:207-225).  ``sd`` maps state-dict names
    to torch-layout tensors (CPU state dict, or the trainer's device-resident master weights, which the BatchNorm
    affine parameters then alias); ``buffers`` (optional) holds the running statistics across re-packs."""

    def __init__(self, sd: dict, prefix: str, winograd: int, buffers: Optional[dict] = None):
        self.conv1 = ops.pack_conv(sd[prefix + '.conv1.weight'])
        w2 = sd[prefix + '.conv2.weight']
        self.conv2 = ops.pack_conv(w2, pad=1)
        self.conv2_wg = ops.pack_winograd(w2, m=winograd) \
            if winograd and w2.shape[1] % 32 == 0 and w2.shape[0] % 4 == 0 else None
        self.conv3 = ops.pack_conv(sd[prefix + '.conv3.weight'])
        self.prefix = prefix
        self.bn = []
        for i in (1, 2, 3):
            b = {k: sd[f'{prefix}.bn{i}.{k}'].detach().float() for k in ('weight', 'bias')}
            for k in ('running_mean', 'running_var'):
                name = f'{prefix}.bn{i}.{k}'
                b[k] = buffers[name] if buffers is not None else sd[name].detach().float().clone()
            self.bn.append(b)

    def repack_(self, sd: dict) -> None:
        """The updated weights into the packed layers this block already holds (in place; the BatchNorm affine
        parameters alias the trainer's master weights and the running buffers are the trainer's own)."""
        p = self.prefix
        ops.repack_conv_(self.conv1, sd[p + '.conv1.weight'])
        ops.repack_conv_(self.conv2, sd[p + '.conv2.weight'])
        if self.conv2_wg is not None:
            ops.repack_winograd_(self.conv2_wg, sd[p + '.conv2.weight'])
        ops.repack_conv_(self.conv3, sd[p + '.conv3.weight'])

    def to(self, device):
        for l in (self.conv1, self.conv2, self.conv3, self.conv2_wg):
            if l is not None:
                l.to(device)
        self.bn = [{k: v.contiguous().to(device) for k, v in b.items()} for b in self.bn]
        return self

    def _bn(self, i, y, eps, momentum, relu, residual=None, keep=False):
        b = self.bn[i]
        return ops.bn_train(y, b['weight'], b['bias'], eps, momentum, b['running_mean'], b['running_var'],
                            residual=residual, relu=relu, inplace=not keep)

    def __call__(self, x, eps, momentum, tape: Optional[list] = None):
        """``tape`` (a list) receives what the backward pass needs: the input, the three pre-norm convolution outputs
        with their batch statistics, and the post-ReLU activations."""
        keep = tape is not None
        c1 = ops.conv2d(x, self.conv1)
        y1, m1, v1 = self._bn(0, c1, eps, momentum, True, keep=keep)
        wg = self.conv2_wg
        if wg is not None and ops.winograd_pays(y1.shape[0], y1.shape[1], y1.shape[2], wg.cin, wg.cout, wg.m):
            c2 = ops.conv3x3_winograd(y1, wg)
        else:
            c2 = ops.conv2d(y1, self.conv2)
        y2, m2, v2 = self._bn(1, c2, eps, momentum, True, keep=keep)
        c3 = ops.conv2d(y2, self.conv3)
        out, m3, v3 = self._bn(2, c3, eps, momentum, True, residual=x, keep=keep)    # relu(bn3(conv3) + identity)
        if keep:
            tape.append(dict(x=x, c1=c1, y1=y1, c2=c2, y2=y2, c3=c3, out=out, stats=((m1, v1), (m2, v2), (m3, v3))))
        return out


def pack_train(model, device, weights: Optional[dict] = None, buffers: Optional[dict] = None) -> None:
    """Raw (un-folded) shared-head layers and their BatchNorm parameters / running buffers on ``device``."""
    nb = model.cfg['roi_head']['shared_head']['num_blocks']
    tr = model._trainer_alive() if hasattr(model, '_trainer_alive') else None
    if weights is None and tr is not None:         # never rebuild from the stale initial state dict while training
        weights, buffers = tr.W, tr.buffers
    src = model._sd if weights is None else weights
    anchors = model._PT['anchors'] if getattr(model, '_PT', None) else {}
    pinned = model._PT.get('pinned', {}) if getattr(model, '_PT', None) else {}
    with ops.gemm_math('f32'):      # layers whose weights an optimizer rewrites in place: no bf16-plane image to keep in step
        shared = [_SharedBlockTrain(src, f'roi_head.shared_head.{b}', model.use_winograd, buffers).to(device) for b in range(nb)]
    model._PT = {'shared': shared, 'device': torch.device(device), 'anchors': anchors, 'pinned': pinned}


def shared_head_train(model, x, momentum: float, tape: Optional[list] = None):
    eps = model.cfg['backbone']['bn_eps']
    for blk in model._PT['shared']:
        x = blk(x, eps, momentum, tape)
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


def _choose(cand: np.ndarray, num: int, perm_fn) -> np.ndarray:
    """RandomSampler._sample_pos/_sample_neg + the ``unique()`` of BaseSampler.sample: at most ``num`` of ``cand``
    (ascending indices), drawn with the CPU permutation mmdet draws (my_random_sampler.py:58-59)."""
    if cand.size > num:
        cand = np.sort(cand[_perm(perm_fn, cand.size)[:num].numpy()])
    return cand


_PERM_LOCK = threading.Lock()


def _pinned(model, name: str, n: int) -> torch.Tensor:
    """A cached pinned int32 host buffer of at least ``n`` elements (allocating pinned memory costs milliseconds)."""
    cache = model._PT.setdefault('pinned', {})
    buf = cache.get(name)
    if buf is None or buf.numel() < n:
        buf = cache[name] = torch.empty(max(n, 1), dtype=torch.int32, pin_memory=True)
    return buf[:max(n, 1)] if buf.numel() != max(n, 1) else buf


def _perm(perm_fn, n: int) -> torch.Tensor:
    """``perm_fn(n)``; ``torch.randperm`` of more than 32768 elements on ONE intra-op thread.  torch fills the
    identity permutation with ``at::parallel_for`` (grain 32768) before its sequential Fisher-Yates shuffle; on a
    128-thread host waking the sleeping OpenMP pool for that fill costs 4.5-5 ms per call (0.19 ms on one thread,
    tools/micro/randperm_probe.py) - six draws of ~60 000 anchors were 34 of the 43 ms of a ``forward_train`` call.
    The permutation does not depend on the thread count (tests/test_host_cpu.py), so seed parity with mmdet holds."""
    if perm_fn is torch.randperm and n > 32768:
        # the thread count is process-global: one caller at a time flips it, and the value to restore is read INSIDE
        # the lock, so a racing call can never read the other's temporary 1 and leave the pool there
        with _PERM_LOCK:
            k = torch.get_num_threads()
            if k == 1:
                return perm_fn(n)
            torch.set_num_threads(1)
            try:
                return perm_fn(n)
            finally:
                torch.set_num_threads(k)
    return perm_fn(n)


def _sample(gt_inds: np.ndarray, num: int, pos_fraction: float, perm_fn):
    """On the host copy of the assignment: one device->host copy per stage replaces a sync per sampled set."""
    pos = _choose(np.flatnonzero(gt_inds > 0), int(num * pos_fraction), perm_fn)
    neg = _choose(np.flatnonzero(gt_inds == 0), num - pos.size, perm_fn)
    return pos, neg


class _Staging:
    """Pinned staging memory for the small host arrays of a training step (sampled indices, labels, box targets: ~15
    per step).  A ``torch.from_numpy(a).to(dev)`` from pageable memory is a SYNCHRONOUS copy - the host waits ~50 us for
    each while the GPU idles between them (rocprofv3 timeline, round 4: 0.7 ms per step); from pinned memory the same
    copy is queued and the host moves on.  Two arenas alternate per ``forward_train`` call: a call waits for the GPU at
    least once after its first copy (the assignment vectors), so every copy of call n has left its arena before call
    n + 2 writes there.  An array that does not fit takes the pageable path."""

    def __init__(self, nbytes: int = 4 << 20):
        self.bufs = [torch.empty(nbytes, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        self.cur, self.off = 0, 0

    def next_call(self) -> None:
        self.cur ^= 1
        self.off = 0

    def h2d(self, a: np.ndarray, dev) -> torch.Tensor:
        a = np.ascontiguousarray(a)
        start = (self.off + 63) // 64 * 64
        if a.nbytes == 0 or start + a.nbytes > self.bufs[self.cur].numel():
            return torch.from_numpy(a).to(dev, non_blocking=True)
        self.off = start + a.nbytes
        host = self.bufs[self.cur][start:self.off]
        host.numpy().view(a.dtype).reshape(a.shape)[...] = a
        return host.view(torch.from_numpy(a[:0]).dtype).view(a.shape).to(dev, non_blocking=True)


def _staging(model) -> _Staging:
    cache = model._PT.setdefault('pinned', {})
    st = cache.get('h2d')
    if st is None:
        st = cache['h2d'] = _Staging()
    return st


def _h2d(model, a, dev, dtype=None) -> torch.Tensor:
    a = np.asarray(a)
    return _staging(model).h2d(a if dtype is None else a.astype(dtype, copy=False), dev)


def _dev_idx(a, dev, model=None) -> torch.Tensor:
    a = np.ascontiguousarray(a, dtype=np.int64)
    return _staging(model).h2d(a, dev) if model is not None else torch.from_numpy(a).to(dev, non_blocking=True)


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
    tape = getattr(model, '_tape', None)         # dict: the Trainer asks for what the backward pass needs
    B = qry_img.shape[0]
    main = torch.cuda.current_stream()

    # ---- modify_input (fgn.py:79-108): H2D, YXYX -> XYXY on private copies
    qry = qry_img.to(dev, torch.float32, non_blocking=True)
    gt_h = [torch.as_tensor(b).detach().cpu().float().reshape(-1, 4)[:, [1, 0, 3, 2]].contiguous().numpy()
            for b in qry_bboxes]                                            # host copies drive the bookkeeping
    cat_h = [torch.as_tensor(c).detach().cpu().long().reshape(-1).numpy() for c in qry_cat_ids]
    _staging(model).next_call()
    gt_xyxy = [_h2d(model, g, dev) for g in gt_h]
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
    # per-(image, class) GT lists (fgn_ag_rpn_head.py:58-73) and the assignment of all G guided passes
    grp_gt_h = [gt_h[g // N][cat_h[g // N] == (g % N)] for g in range(G)]
    gi_all = torch.empty((G, n_total), device=dev, dtype=torch.int32)
    for g in range(G):
        gts = _h2d(model, grp_gt_h[g], dev) if len(grp_gt_h[g]) else gt_xyxy[0][:0]
        ops.box_assign(anchors, gts, tc['pos_iou_thr'], tc['neg_iou_thr'], tc['min_pos_iou'], tc['match_low_quality'],
                       inside=inside, out=gi_all[g])
    # the anchor assignment starts its way to the host NOW (pinned buffer, no wait): the host samples the AG-RPN sets
    # from it while the GPU runs the proposal stage, the proposals' assignment and the support branch queued below
    # (round 4: one blocking copy after all of that left the GPU idle through ~1.5 ms of host sampling)
    gi_pin = _pinned(model, 'gi_all', G * n_total)
    gi_pin.copy_(gi_all.view(-1), non_blocking=True)
    gi_ready = torch.cuda.current_stream().record_event()
    # ---- proposals with train_cfg.rpn_proposal (fgn.py:161-167)
    rc = tcfg['rcnn']
    if proposals is None:
        pc = tcfg['rpn_proposal']
        _, scores, deltas = ops.rpn_merge(head, B, N, A)
        props, n_props = ops.rpn_proposals(scores, deltas, P['anchors'], fh, fw, rp['anchor_stride'], ih, iw,
                                           rp['target_means'], rp['target_stds'], pc['nms_pre'], pc['min_bbox_size'],
                                           pc['nms_iou_threshold'], pc['max_per_img'])
        prop_list = [props[i] for i in range(B)]           # rows past n_props[i] are zero boxes, never sampled
    else:
        prop_list = [torch.as_tensor(p).to(dev, torch.float32).contiguous() for p in proposals]
        n_props = torch.tensor([p.shape[0] for p in prop_list], device=dev, dtype=torch.int32)

    # FGNRoIHead.forward_train (fgn_roi_head.py:451-529): assignment of every image's proposals
    gis = [ops.box_assign(prop_list[i], gt_xyxy[i], rc['pos_iou_thr'], rc['neg_iou_thr'], rc['min_pos_iou'],
                          rc['match_low_quality']) if prop_list[i].shape[0] else
           torch.zeros((0,), device=dev, dtype=torch.int32) for i in range(B)]
    # count_spp with the shared head in training mode (fgn_roi_head.py:491, 419-449): queued before the host copy too
    spp_tape = [] if tape is not None else None
    model._support_back(sc, B, dev, shared=lambda t: shared_head_train(model, t, bn_momentum, spp_tape))
    # second (small) copy: the proposals' assignment vectors and counts; the reference-style sampling bookkeeping runs on
    # the host copies - the AG-RPN part as soon as the first copy has landed
    tail_dev = torch.cat(gis + [n_props.to(torch.int32)])
    tail_pin = _pinned(model, 'gi_tail', tail_dev.numel())
    tail_pin.copy_(tail_dev, non_blocking=True)
    tail_ready = torch.cuda.current_stream().record_event()
    gi_ready.synchronize()
    gi_host = gi_pin.numpy()[:G * n_total].reshape(G, n_total)
    n_pos_total = n_neg_total = 0
    sets_h, flat, ycat, pos_flat, pos_anchor, pos_gt = [], [], [], [], [], []
    for g in range(G):
        pos, neg = _sample(gi_host[g], tc['num'], tc['pos_fraction'], perm_fn)
        sets_h.append((pos, neg))
        n_pos_total += max(pos.size, 1)
        n_neg_total += max(neg.size, 1)
        flat += [g * n_total + pos, g * n_total + neg]
        ycat += [np.ones(pos.size, np.float32), np.zeros(neg.size, np.float32)]
        if pos.size:
            pos_flat.append(g * n_total + pos)
            pos_anchor.append(pos)
            pos_gt.append(grp_gt_h[g][gi_host[g][pos] - 1])
    n_samples = n_pos_total + n_neg_total
    pw = 1.0 if tc['pos_weight'] <= 0 else float(tc['pos_weight'])
    x_cat = logits_all.reshape(-1)[_dev_idx(np.concatenate(flat), dev, model)].contiguous()
    y_cat = _h2d(model, np.concatenate(ycat), dev)
    w_cat = None if pw == 1.0 else torch.where(y_cat > 0, pw, 1.0).float().contiguous()
    loss_rpn_cls = ops.bce_logits_sum(x_cat, y_cat, w_cat, n_samples) / N              # the 1/N balancer
    preds = tgts = None
    if pos_flat:
        preds = deltas_all.reshape(-1, 4)[_dev_idx(np.concatenate(pos_flat), dev, model)].contiguous()
        tgts = ops.bbox2delta(anchors[_dev_idx(np.concatenate(pos_anchor), dev, model)].contiguous(),
                              _h2d(model, np.concatenate(pos_gt), dev, np.float32),
                              rp['target_means'], rp['target_stds'])
        loss_rpn_bbox = ops.smooth_l1_sum(preds, tgts, None, n_samples) / N
    else:
        loss_rpn_bbox = torch.zeros(1, device=dev)
    losses = {'loss_rpn_cls': [loss_rpn_cls.view(())], 'loss_rpn_bbox': [loss_rpn_bbox.view(())]}
    rpn_sets = [(torch.from_numpy(p_), torch.from_numpy(n_)) for p_, n_ in sets_h]
    if tr is not None:
        tr.update(qry_fmap=qry_fmap, rpn_head=head, rpn_sets=rpn_sets, rpn_num_total_samples=n_samples)
    if tape is not None:
        tape['rpn'] = dict(x=x, head=head, sets=sets_h, x_cat=x_cat, y_cat=y_cat, w_cat=w_cat, n_samples=n_samples,
                           preds=preds, tgts=tgts, qry_fmap=qry_fmap, vec=sc['vec'], A=A, n_ways=N, n_total=n_total)

    tail_ready.synchronize()
    packed = tail_pin.numpy()[:tail_dev.numel()].copy()
    counts = packed[-B:].tolist()
    if tr is not None:
        tr['proposals'] = [prop_list[i][:counts[i]] for i in range(B)]
    samples, roi_parts, lab_parts, off = [], [], [], 0
    for i in range(B):
        gi = packed[off:off + counts[i]]
        off += prop_list[i].shape[0]
        k = gt_h[i].shape[0]
        boxes = prop_list[i][:, :4]
        if rc['add_gt_as_proposals'] and k > 0:           # BaseSampler.sample: GT boxes in front, assigned to themselves
            boxes = torch.cat([gt_xyxy[i], boxes])
            gi = np.concatenate([np.arange(1, k + 1, dtype=np.int32), gi])
        pos, neg = _sample(gi, rc['num'], rc['pos_fraction'], perm_fn)
        assigned = (gi[pos] - 1).astype(np.int64)
        sel = boxes[_dev_idx(np.concatenate([pos, neg]), dev, model)]
        samples.append(dict(pos_bboxes=sel[:pos.size], n_pos=int(pos.size), n_neg=int(neg.size),
                            pos_assigned_gt_inds=assigned, pos_gt_bboxes_h=gt_h[i][assigned].reshape(-1, 4),
                            pos_gt_labels=torch.from_numpy(cat_h[i][assigned]), pos_inds=torch.from_numpy(pos),
                            neg_inds=torch.from_numpy(neg)))
        roi_parts.append(torch.cat([torch.full((sel.shape[0], 1), float(i), device=dev), sel], 1))
        lab_parts.append(np.concatenate([cat_h[i][assigned], np.full(neg.size, N, np.int64)]))
    if tr is not None:
        tr['samples'] = samples

    # _bbox_forward_train (fgn_roi_head.py:344-358)
    rois = torch.cat(roi_parts).contiguous()
    n_rois = rois.shape[0]
    rel, bh = rh['relation'], rh['bbox_head']
    if n_rois:
        xr = ops.roi_align(qry_fmap, rois, rh['roi_out_size'], 1.0 / rh['featmap_stride'], rh['roi_sampling_ratio'],
                           True)
        roi_tape = [] if tape is not None else None
        feats = shared_head_train(model, xr, bn_momentum, roi_tape)
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
    lab_h = np.concatenate(lab_parts) if lab_parts else np.zeros(0, np.int64)
    labels = _h2d(model, lab_h, dev, np.int64)
    pos_rows_h = np.flatnonzero(lab_h < N)
    pos_rows = _dev_idx(pos_rows_h, dev, model)
    pw = 1.0 if rc['pos_weight'] <= 0 else float(rc['pos_weight'])
    lw = None if pw == 1.0 else torch.where(labels < N, pw, 1.0).float().contiguous()
    avg = max(float(n_rois), 1.0)                     # every sampled RoI has label weight > 0
    if n_rois:
        losses['loss_cls'] = ops.softmax_ce_sum(cls_score, labels, lw, avg).view(())
        # get_accuracy (fgn_roi_head.py:100-116: accuracy_score / balanced_accuracy_score of sklearn) on the DEVICE, in
        # fp64 like numpy: no host read of the predictions in the middle of the step (round 4: the step had three host
        # synchronisations - the assignment vectors, this one and the logit bias of the re-pack - and two of them only
        # made the GPU wait for the next burst of launches).  Returned as one-element device tensors.
        hit = (cls_score.argmax(dim=-1) == labels).double()
        onehot = torch.nn.functional.one_hot(labels, N + 1).double()
        cnt = onehot.sum(0)
        present = (cnt > 0).double()
        recall = (onehot * hit[:, None]).sum(0) / cnt.clamp(min=1.0)
        losses['ACC-Unbalanced'] = hit.mean().float().reshape(1)
        losses['ACC-Balanced'] = ((recall * present).sum() / present.sum()).float().reshape(1)
    pos_pred = pos_tgt = None
    if pos_rows_h.size:
        pos_pred = bbox_pred.view(n_rois, -1, 4)[pos_rows, labels[pos_rows]].contiguous()
        pos_tgt = ops.bbox2delta(torch.cat([s['pos_bboxes'] for s in samples]).contiguous(),
                                 _h2d(model, np.concatenate([s['pos_gt_bboxes_h'] for s in samples]), dev, np.float32),
                                 bh['target_means'], bh['target_stds'])
        losses['loss_bbox'] = ops.smooth_l1_sum(pos_pred, pos_tgt, None, float(n_rois)).view(())
    else:
        losses['loss_bbox'] = _zero(dev)
    if tr is not None:
        tr.update(rois=rois, bbox_feats=feats, cls_score=cls_score, bbox_pred=bbox_pred, labels=labels)
    if tape is not None:
        # (an episode without proposals and without ground truth samples no RoI: the RoI stage then contributes zero
        # gradients - a rank-local exception here would strand the other ranks of a data-parallel job in all_reduce)
        tape['roi'] = None if not n_rois else dict(rois=rois, blocks=roi_tape, feats=feats, Q=Q, S=sc['S'], cls_raw=cls_raw, cls_score=cls_score,
                           bbox_pred=bbox_pred, labels=labels, lw=lw, avg=avg, pos_rows=pos_rows, n_rois=n_rois,
                           pos_pred=pos_pred, pos_tgt=pos_tgt,
                           img_counts=[s['n_pos'] + s['n_neg'] for s in samples])
        tape['spp'] = dict(blocks=spp_tape, masks7=sc['masks7'], cat_mean=sc['cat_mean'], B=B)

    # ---- mask branch (fgn_roi_head.py:498-527, 384-417): shared RoI extractor -> the positives' bbox_feats
    n_pos = int(pos_rows_h.size)
    if n_pos:
        img_of_h = np.concatenate([np.full(s['n_pos'], i, np.int64) for i, s in enumerate(samples)])
        vrows = _dev_idx(lab_h[pos_rows_h] + N * img_of_h, dev, model)
        vmask = sc['cat_mean_mp'][vrows].contiguous()                                     # spp_vecs_mask
        mfeat = feats[pos_rows].contiguous()
        if tape is None:
            mlog, _ = model._mask_head(mfeat, vmask)
        else:
            mlog, macts, mup = _mask_head_taped(model, mfeat, vmask)
        # mask_target_single + BitmapMasks.crop_and_resize: RoIAlign(aligned, adaptive grid) of the GT bitmaps
        gt_masks = []
        for m_ in qry_isegmaps:
            m_ = torch.as_tensor(m_).to(dev, non_blocking=True)
            gt_masks.append((m_ if m_.dtype in (torch.bool, torch.uint8) else (m_ != 0)).to(torch.uint8))
        first = np.cumsum([0] + [m_.shape[0] for m_ in gt_masks[:-1]])
        masks_all = torch.cat(gt_masks).contiguous()
        mh_, mw_ = masks_all.shape[-2:]
        pb = torch.cat([s['pos_bboxes'] for s in samples])
        pb = torch.stack([pb[:, 0].clamp(0, mw_), pb[:, 1].clamp(0, mh_), pb[:, 2].clamp(0, mw_),
                          pb[:, 3].clamp(0, mh_)], 1)
        gidx = np.concatenate([s['pos_assigned_gt_inds'] + int(first[i]) for i, s in enumerate(samples)])
        mrois = torch.cat([_h2d(model, gidx, dev, np.float32)[:, None], pb], 1).contiguous()
        ms = rc['mask_size']
        tgt = ops.roi_align_mask(masks_all, mrois, ms, 1.0, 0, True)                  # [n_pos, ms, ms] in [0,1]
        if tuple(mlog.shape[1:]) != (ms, ms):
            raise ValueError(f'mask head output {tuple(mlog.shape[1:])} != train_cfg mask_size {ms}')
        losses['loss_mask'] = ops.bce_logits_sum(mlog.contiguous(), tgt, None, float(mlog.numel()), y_threshold=0.5)
        if tr is not None:
            tr.update(mask_pred=mlog, mask_targets_soft=tgt)
        if tape is not None:
            tape['mask'] = dict(mfeat=mfeat, vmask=vmask, acts=macts, up=mup, mlog=mlog.contiguous(), tgt=tgt, rows=vrows,
                                rows_h=lab_h[pos_rows_h] + N * img_of_h)
    else:
        losses['loss_mask'] = _zero(dev)
        if tape is not None:
            tape['mask'] = None
    return losses


def _mask_head_taped(model, mf, vmask):
    """``FGN._mask_head`` keeping every activation: -> logits [D,14,14], [m1..m4] (post-ReLU), up [D,7,7,4*C']."""
    P = model._P
    acts, m = [], mf
    for li, (layer, wg) in enumerate(zip(P['mask_convs'], P['mask_convs_wg'])):
        scale = vmask if li == 0 else None
        if wg is not None and ops.winograd_pays(m.shape[0], m.shape[1], m.shape[2], wg.cin, wg.cout, wg.m):
            m = ops.conv3x3_winograd(m, wg, in_scale=scale)
        else:
            m = ops.conv2d(m, layer, in_scale=scale)
        acts.append(m)
    up = ops.conv2d(m, P['upsample'])
    mlog, _ = ops.mask_logits(up, P['logit_w'], P['logit_b'], model.cfg['roi_head']['roi_out_size'], None)
    return mlog, acts, up


# ------------------------------------------------------------------------------------------
# backward of the trainable heads
# ------------------------------------------------------------------------------------------
def _mm_tn(a2: torch.Tensor, b2: torch.Tensor) -> torch.Tensor:
    """a2 [rows, M], b2 [rows, K] -> a2^T b2 [M, K]: the weight-gradient product (reduction over the rows) on the
    MFMA kernel ``fgn_gemm_tn_f32``; the 6-row fc gradient (M not a multiple of 4) on ``fgn_gemm_small_f32``."""
    if a2.shape[0] == 0:
        return torch.zeros((a2.shape[1], b2.shape[1]), device=a2.device, dtype=torch.float32)
    if a2.shape[1] % 4 or b2.shape[1] % 4:
        return ops.gemm_small(a2.contiguous(), b2.contiguous(), trans_a=True)
    return ops.gemm_tn(a2.contiguous(), b2.contiguous())


def _dgrad_1x1(dy: torch.Tensor, w2: torch.Tensor) -> torch.Tensor:
    """Data gradient of a 1x1 convolution, dy [..., Cout] x w2 [Cout, Cin] -> [..., Cin]: the forward convolution
    kernel with the transposed weight (Cout is the reduction, a multiple of 32 for every trainable 1x1 layer)."""
    cout, cin = w2.shape
    if cout % 32:
        return ops.gemm_small(dy.reshape(-1, cout).contiguous(), w2.contiguous()).view(tuple(dy.shape[:-1]) + (cin,))
    x = dy.contiguous().view(1, -1, 1, cout)
    return ops.conv2d(x, ops.pack_conv(w2.t().contiguous().view(cin, cout, 1, 1))).view(tuple(dy.shape[:-1]) + (cin,))


def _conv3x3_dgrad(dy: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """Data gradient of a 3x3 / stride 1 / pad 1 convolution with torch-layout weight [Cout,Cin,3,3]: the forward
    convolution kernel applied to dy with the kernel flipped and the channel roles swapped."""
    wd = w.flip(2, 3).transpose(0, 1).contiguous()                   # [Cin, Cout, 3, 3]
    return ops.conv2d(dy.contiguous(), ops.pack_conv(wd, pad=1))


def _conv3x3_wgrad(dy: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """-> dW [Cout,Cin,3,3] = dy^T . im2col(x)."""
    cout, cin = dy.shape[-1], x.shape[-1]
    return _mm_tn(dy.reshape(-1, cout), ops.im2col3x3(x.contiguous())).view(cout, 3, 3, cin).permute(0, 3, 1, 2) \
        .contiguous()


def _block_backward(blk: _SharedBlockTrain, W: dict, t: dict, dout: torch.Tensor, eps: float, need_dx: bool, grads: dict):
    """One shared-head Bottleneck (train-mode BN) backwards; weight gradients are ACCUMULATED into ``grads`` (the
    block runs on the RoI batch and on the support batch)."""
    p = blk.prefix

    def acc(name, g):
        grads[name] = g if name not in grads else grads[name] + g
    (m1, v1), (m2, v2), (m3, v3) = t['stats']
    w1 = W[p + '.conv1.weight'].view(W[p + '.conv1.weight'].shape[0], -1)
    w2 = W[p + '.conv2.weight']
    w3 = W[p + '.conv3.weight'].view(W[p + '.conv3.weight'].shape[0], -1)
    d3, dg3, db3, g_id = ops.bn_train_backward(t['c3'], t['out'], dout.contiguous(), m3, v3, blk.bn[2]['weight'], eps,
                                               want_g=True)
    acc(p + '.bn3.weight', dg3); acc(p + '.bn3.bias', db3)
    d3f = d3.view(-1, w3.shape[0])
    acc(p + '.conv3.weight', _mm_tn(d3f, t['y2'].reshape(-1, w3.shape[1])).view_as(W[p + '.conv3.weight']))
    dy2 = _dgrad_1x1(d3, w3).view_as(t['y2'])
    d2, dg2, db2 = ops.bn_train_backward(t['c2'], t['y2'], dy2, m2, v2, blk.bn[1]['weight'], eps)
    acc(p + '.bn2.weight', dg2); acc(p + '.bn2.bias', db2)
    acc(p + '.conv2.weight', _conv3x3_wgrad(d2, t['y1']))
    dy1 = _conv3x3_dgrad(d2, w2)
    d1, dg1, db1 = ops.bn_train_backward(t['c1'], t['y1'], dy1, m1, v1, blk.bn[0]['weight'], eps)
    acc(p + '.bn1.weight', dg1); acc(p + '.bn1.bias', db1)
    d1f = d1.view(-1, w1.shape[0])
    acc(p + '.conv1.weight', _mm_tn(d1f, t['x'].reshape(-1, w1.shape[1])).view_as(W[p + '.conv1.weight']))
    if not need_dx:
        return None
    return g_id + _dgrad_1x1(d1, w1).view_as(t['x'])


def _shared_backward(model, W, blocks_tape, dout, grads):
    eps = model.cfg['backbone']['bn_eps']
    blks = model._PT['shared']
    for bi in range(len(blks) - 1, -1, -1):
        dout = _block_backward(blks[bi], W, blocks_tape[bi], dout, eps, bi > 0, grads)


def backward(model, W: dict, tape: dict) -> dict:
    """Gradients of the summed losses (mmdet ``_parse_losses``: every key containing 'loss') with respect to the
    trainable parameters ``W`` (torch-layout master weights on the device), from the tape of ``forward_train``.
    The backbone is frozen (fgn.py:67-73), so nothing flows below RoIAlign / the AG-RPN input."""
    grads: dict = {}
    if tape['roi'] is not None:
        _backward_roi_stage(model, W, tape, grads)
    _backward_rpn_stage(model, W, tape, grads)
    return grads


def _backward_roi_stage(model, W: dict, tape: dict, grads: dict) -> None:
    cfg = model.cfg
    N, K = model.n_ways, model.k_shots
    P = model._P
    rh = cfg['roi_head']
    dev = tape['roi']['rois'].device

    # ---- mask head (fgn_roi_head.py:360-417) ------------------------------------------------------------
    tr_, tm = tape['roi'], tape['mask']
    feats = tr_['feats']
    d_feats = None
    C = feats.shape[-1]
    d_cat_mean_mp = None
    if tm is not None:
        mlog = tm['mlog']
        dlogit = ops.bce_logits_grad(mlog, tm['tgt'], None, 1.0 / float(mlog.numel()), y_threshold=0.5)
        lw = W['roi_head.mask_head.conv_logits.weight'].reshape(-1).contiguous()
        d_up, dlw = ops.mask_logits_backward(tm['up'], dlogit, lw, rh['roi_out_size'])
        grads['roi_head.mask_head.conv_logits.weight'] = dlw.view_as(W['roi_head.mask_head.conv_logits.weight'])
        grads['roi_head.mask_head.conv_logits.bias'] = dlogit.sum().view(1)
        wt = W['roi_head.mask_head.upsample.weight']                         # [Cin, Cout, 2, 2]
        cin_u, cout_u = wt.shape[:2]
        w4 = wt.permute(2, 3, 1, 0).reshape(4 * cout_u, cin_u)
        acts = tm['acts']
        d_upf = d_up.view(-1, 4 * cout_u)
        dw4 = _mm_tn(d_upf, acts[-1].reshape(-1, cin_u))                     # [4*Cout, Cin]
        grads['roi_head.mask_head.upsample.weight'] = dw4.view(2, 2, cout_u, cin_u).permute(3, 2, 0, 1).contiguous()
        grads['roi_head.mask_head.upsample.bias'] = ops.colsum(d_upf).view(4, cout_u).sum(0)
        dm = _dgrad_1x1(d_up, w4).view_as(acts[-1])
        for li in range(len(acts) - 1, -1, -1):
            name = f'roi_head.mask_head.convs.{li}.conv'
            dpre = ops.relu_backward(dm.contiguous(), acts[li])                # ReLU
            x_in = acts[li - 1] if li > 0 else (tm['mfeat'] * tm['vmask'][:, None, None, :]).contiguous()
            grads[name + '.weight'] = _conv3x3_wgrad(dpre, x_in)
            grads[name + '.bias'] = ops.colsum(dpre)
            dm = _conv3x3_dgrad(dpre, W[name + '.weight'])
        # dm = gradient of the guided input mfeat * vmask (fgn_roi_head.py:379)
        d_mfeat = ops.scale_channels(dm.contiguous(), tm['vmask'], 1)          # dm * vmask per (RoI, channel)
        d_vmask = (dm * tm['mfeat']).sum(dim=(1, 2))                         # [n_pos, C]
        d_feats = torch.zeros_like(feats)
        d_feats.index_add_(0, tr_['pos_rows'], d_mfeat)
        # sum over the positives that share a (image, class) support vector: column sums in a fixed order (an
        # index_add_ would accumulate with atomics, in an order that changes from run to run)
        d_cat_mean_mp = torch.zeros((tape['spp']['B'] * N, C), device=dev)
        for r_ in np.unique(tm['rows_h']):
            sel = _dev_idx(np.flatnonzero(tm['rows_h'] == r_), dev, model)
            ops.colsum(d_vmask[sel].contiguous(), out=d_cat_mean_mp[int(r_)])

    # ---- box head losses -> relation head (fgn_roi_head.py:58-118, 253-279, 302-326) ---------------------
    n = tr_['n_rois']
    dcls = ops.softmax_ce_grad(tr_['cls_score'], tr_['labels'], tr_['lw'], 1.0 / tr_['avg'])
    dbbox = torch.zeros((n, N, 4), device=dev)
    if tr_['pos_pred'] is not None:
        dpos = ops.smooth_l1_grad(tr_['pos_pred'], tr_['pos_tgt'], None, 1.0 / float(n))
        dbbox[tr_['pos_rows'], tr_['labels'][tr_['pos_rows']]] = dpos
    if N == 1:
        dcls_raw = dcls[:, [1, 0]]
    else:
        resh = tr_['cls_raw'].view(n, 2 * N)
        top = resh[:, 1::2].argmax(dim=1) * 2
        d_resh = torch.zeros_like(resh)
        d_resh[:, 1::2] = dcls[:, :N]
        d_resh[torch.arange(n, device=dev), top] += dcls[:, N]
        dcls_raw = d_resh.view(n * N, 2)
    d6 = torch.cat([dcls_raw, dbbox.view(n * N, 4)], 1).contiguous()
    rel = rh['relation']
    dQ, dZ, pooled, dgn_w, dgn_b = ops.relation_gn_head_backward(tr_['Q'], tr_['S'], tr_['rois'], P['gn_w'], P['gn_b'],
                                                                 P['fc_w'], d6, N, rel['gn_groups'], rel['gn_eps'])
    grads['roi_head.cls_reg_shared_conv_norm.weight'] = dgn_w
    grads['roi_head.cls_reg_shared_conv_norm.bias'] = dgn_b
    dfc = _mm_tn(d6, pooled)                                                 # [6, C]
    dfcb = ops.colsum(d6)
    grads['roi_head.bbox_head.fc_cls.weight'], grads['roi_head.bbox_head.fc_reg.weight'] = dfc[:2].contiguous(), \
        dfc[2:].contiguous()
    grads['roi_head.bbox_head.fc_cls.bias'], grads['roi_head.bbox_head.fc_reg.bias'] = dfcb[:2].contiguous(), \
        dfcb[2:].contiguous()
    wrel = W['roi_head.cls_reg_shared_conv.weight'].view(C, 2 * C)
    wq, ws = wrel[:, :C], wrel[:, C:]
    dQf = dQ.view(-1, C)
    d_from_q = _dgrad_1x1(dQ, wq.contiguous()).view_as(feats)
    d_feats = d_from_q if d_feats is None else d_feats + d_from_q
    dWq = _mm_tn(dQf, feats.reshape(-1, C))
    # dS[b, cls] = sum over the RoIs of image b of dZ[r, cls]  (RoIs are image-major: bbox2roi)
    B = tape['spp']['B']
    counts = tr_['img_counts']
    dS = torch.zeros_like(tr_['S'])
    r0 = 0
    for b in range(B):
        if counts[b]:
            ops.colsum(dZ[r0 * N:(r0 + counts[b]) * N].view(counts[b], -1), out=dS[b * N:(b + 1) * N].view(-1))
        r0 += counts[b]
    dSf = dS.view(-1, C)
    cat_mean = tape['spp']['cat_mean']
    dWs = _mm_tn(dSf, cat_mean.reshape(-1, C))
    grads['roi_head.cls_reg_shared_conv.weight'] = torch.cat([dWq, dWs], 1).view_as(
        W['roi_head.cls_reg_shared_conv.weight'])
    grads['roi_head.cls_reg_shared_conv.bias'] = ops.colsum(dSf)
    d_cat_mean = _dgrad_1x1(dS, ws.contiguous()).view_as(cat_mean)          # [B*N, 7, 7, C]

    # ---- shared head, RoI batch then support batch (count_spp, fgn_roi_head.py:419-449) ------------------
    _shared_backward(model, W, tr_['blocks'], d_feats, grads)
    # cat_mean = mean_k sfeat; cat_mean_mp = mean_{k,p} sfeat * masks7
    ps = cat_mean.shape[1]
    d_sfeat = (d_cat_mean / K).repeat_interleave(K, dim=0)
    if d_cat_mean_mp is not None:
        m7 = tape['spp']['masks7'].view(-1, ps, ps, 1)
        d_sfeat = d_sfeat + d_cat_mean_mp.repeat_interleave(K, dim=0)[:, None, None, :] * m7 / float(K * ps * ps)
    _shared_backward(model, W, tape['spp']['blocks'], d_sfeat.contiguous(), grads)


def _backward_rpn_stage(model, W: dict, tape: dict, grads: dict) -> None:
    N = model.n_ways
    dev = tape['rpn']['head'].device
    # ---- AG-RPN head (fgn_ag_rpn_head.py:48 -> RPNHead.forward_single; losses my_anchor_head.py:402-451) ----
    t = tape['rpn']
    A, head, G = t['A'], t['head'], t['head'].shape[0]
    fh, fw, CH = head.shape[1:]
    scale = 1.0 / (float(t['n_samples']) * N)                                # avg_factor and the 1/N balancer
    dx_cat = ops.bce_logits_grad(t['x_cat'], t['y_cat'], t['w_cat'], scale)
    dhead = torch.zeros((G, fh * fw, CH), device=dev)
    dpred = ops.smooth_l1_grad(t['preds'], t['tgts'], None, scale) if t['preds'] is not None else None
    # one scatter for the objectness columns, one for the delta columns (flat indices built on the host)
    hw = fh * fw
    li, ri = [], []
    for g, (pos, neg) in enumerate(t['sets']):
        for idx in (pos, neg):
            li.append((g * hw + idx // A) * CH + idx % A)
        ri.append(((g * hw + pos // A) * CH + A + 4 * (pos % A))[:, None] + np.arange(4)[None])
    dflat = dhead.view(-1)
    dflat[_dev_idx(np.concatenate(li), dev, model)] = dx_cat
    if dpred is not None:
        dflat[_dev_idx(np.concatenate(ri).reshape(-1), dev, model)] = dpred.reshape(-1)
    rows = torch.nonzero(dhead.abs().sum(-1).view(-1) > 0).view(-1)          # active (pass, pixel) rows: <= num * G
    Cf = t['x'].shape[-1]
    dH = dhead.view(-1, CH)[rows]
    X = t['x'].reshape(-1, Cf)[rows]
    wh = torch.cat([W['rpn_head.rpn_cls.weight'].view(A, Cf), W['rpn_head.rpn_reg.weight'].view(4 * A, Cf)], 0)
    dwh = _mm_tn(dH, X)[:5 * A]                                              # CH = 5A padded to a multiple of 4
    dbh = ops.colsum(dH[:, :5 * A].contiguous())
    grads['rpn_head.rpn_cls.weight'] = dwh[:A].reshape(W['rpn_head.rpn_cls.weight'].shape).contiguous()
    grads['rpn_head.rpn_reg.weight'] = dwh[A:].reshape(W['rpn_head.rpn_reg.weight'].shape).contiguous()
    grads['rpn_head.rpn_cls.bias'], grads['rpn_head.rpn_reg.bias'] = dbh[:A].contiguous(), dbh[A:].contiguous()
    dpre = ops.relu_backward(ops.gemm_small(dH[:, :5 * A], wh.contiguous()), X.contiguous())  # through the ReLU of rpn_conv
    # rpn_conv weight gradient: only the active pixels contribute; their 3x3 neighbourhoods of the guided map
    qf, vec = t['qry_fmap'], t['vec']                                        # [B,h,w,C], [B*N,C]
    Cin = qf.shape[-1]
    gi = rows // (fh * fw)
    py, px = (rows % (fh * fw)) // fw, rows % fw
    qpad = torch.nn.functional.pad(qf, (0, 0, 1, 1, 1, 1))                   # zero border (padding=1)
    taps = [qpad[gi // N, py + ky, px + kx] for ky in range(3) for kx in range(3)]
    patches = (torch.stack(taps, 1) * vec[gi][:, None, :]).reshape(rows.numel(), 9 * Cin)
    grads['rpn_head.rpn_conv.weight'] = _mm_tn(dpre, patches).view(Cf, 3, 3, Cin).permute(0, 3, 1, 2).contiguous()
    grads['rpn_head.rpn_conv.bias'] = ops.colsum(dpre)


TRAINABLE_PREFIXES = ('rpn_head.', 'roi_head.')


def trainable_names(sd: dict) -> list:
    return [k for k, v in sd.items() if k.startswith(TRAINABLE_PREFIXES) and v.is_floating_point()
            and 'running_' not in k and 'num_batches' not in k]


_REF_RANK = {'backbone': 0, 'rpn_head': 1, 'roi_head': 2,
             'stem': 0, 'conv1': 0, 'bn1': 1, 'gn1': 1, 'conv2': 2, 'bn2': 3, 'gn2': 3, 'conv3': 4, 'bn3': 5, 'gn3': 5,
             'downsample': 6, 'layer1': 10, 'layer2': 11, 'layer3': 12, 'layer4': 13,
             'rpn_conv': 0, 'rpn_cls': 1, 'rpn_reg': 2,
             'bbox_head': 0, 'mask_head': 1, 'shared_head': 2, 'cls_reg_shared_conv': 3, 'cls_reg_shared_conv_norm': 4,
             'fc_cls': 0, 'fc_reg': 1, 'convs': 0, 'upsample': 1, 'conv_logits': 2, 'conv': 0, 'weight': 0, 'bias': 1}
_BUFFER_TAILS = ('running_mean', 'running_var', 'num_batches_tracked')


def _layer4_params(bb: dict) -> list:
    """(name, shape) of the parameters of ``backbone.layer4``.  The reference builds all four stages
    (fgn_r50_c4_densecl.py:21 ``num_stages=4``) and main.py:402-405 only shortens ``res_layers``, the list of stage
    NAMES the forward pass walks: the module stays registered, so its (frozen, never used) parameters are part of
    ``model.named_parameters()``, of every checkpoint and of the optimizer's index space.  This build holds no
    layer4; its shapes follow from the architecture (mmdet ResNet ``arch_settings``: Bottleneck x3 at depth 50,
    BasicBlock x2 at depth 18; planes = 2 x the third stage's)."""
    # (a reference config with ``num_stages=3`` - fgn_r50_c4_scratch.py:12 - registers no layer4 at all: nothing phantom)
    if len(bb['stage_blocks']) != 3 or int(bb.get('ref_num_stages', 4)) != 4:
        return []
    planes = 2 * bb['stage_planes'][-1]
    bottleneck = bb.get('block', 'bottleneck') == 'bottleneck'
    exp = 4 if bottleneck else 1
    inpl, out = bb['stage_planes'][-1] * exp, planes * exp
    n_blocks = 3 if bottleneck else 2
    res = []
    for b in range(n_blocks):
        pre = f'backbone.layer4.{b}.'
        cin = inpl if b == 0 else out
        convs = [('conv1', (planes, cin, 1, 1)), ('conv2', (planes, planes, 3, 3)), ('conv3', (out, planes, 1, 1))] \
            if bottleneck else [('conv1', (planes, cin, 3, 3)), ('conv2', (planes, planes, 3, 3))]
        for i, (cname, shape) in enumerate(convs):
            res.append((pre + cname + '.weight', shape))
            res += [(pre + f'bn{i + 1}.weight', (shape[0],)), (pre + f'bn{i + 1}.bias', (shape[0],))]
        if b == 0:
            res.append((pre + 'downsample.0.weight', (out, cin, 1, 1)))
            res += [(pre + 'downsample.1.weight', (out,)), (pre + 'downsample.1.bias', (out,))]
    return res


def reference_param_order(sd: dict, backbone_cfg: dict) -> list:
    """[(name, shape)] of EVERY parameter of the reference's detector in the order mmcv's
    ``DefaultOptimizerConstructor.add_params`` visits them (= ``model.named_parameters()``: a module's own parameters,
    then its children in registration order), which is the index space of ``torch.optim.Adagrad.state_dict()`` in a
    reference checkpoint.  Under ``paramwise_cfg`` (fgn_train_schedule.py:10-15) the constructor appends one param
    group per parameter, FROZEN ONES INCLUDED (``if not param.requires_grad: params.append(param_group); continue``),
    and ``Adagrad.__init__`` creates ``step`` / ``sum`` state for every parameter of every group.
    Registration order: ``backbone`` (conv1, bn1, layer1..layer4; a Bottleneck registers conv1, bn1, conv2, bn2, conv3,
    bn3 and sets ``downsample`` last), ``rpn_head`` (rpn_conv, rpn_cls, rpn_reg), ``roi_head``: ``StandardRoIHead.__init__``
    builds bbox_head (fc_cls, fc_reg) and mask_head (convs, upsample, conv_logits) FIRST, then ``FGNRoIHead.__init__``
    adds shared_head, cls_reg_shared_conv, cls_reg_shared_conv_norm (fgn_roi_head.py:197-200, 240-243; the config passes
    ``shared_head=None``, fgn_r50_c4_densecl.py:68)."""
    items = [(k, tuple(v.shape)) for k, v in sd.items() if not k.endswith(_BUFFER_TAILS)]
    have = {k for k, _ in items}
    items += [(k, shp) for k, shp in _layer4_params(backbone_cfg) if k not in have]

    def key(name):
        try:
            return tuple(int(t) if t.isdigit() else _REF_RANK[t] for t in name.split('.'))
        except KeyError as e:
            raise ValueError(f'{name}: no place in the reference parameter order ({e})') from None
    return sorted(items, key=lambda it: key(it[0]))


def step_lr(base_lr: float, it: int, epoch: int, steps=(3,), gamma: float = 0.1, min_lr: float = 1e-6,
            warmup_iters: int = 100, warmup_ratio: float = 0.01) -> float:
    """mmcv ``StepLrUpdaterHook`` as configured in fgn_train_schedule.py:17-23: lr = base * gamma^(#steps <= epoch),
    floored at ``min_lr``; linear warm-up over the first ``warmup_iters`` iterations from ``warmup_ratio`` of it
    (mmcv: warmup_lr = regular_lr * (1 - (1 - it / warmup_iters) * (1 - warmup_ratio)))."""
    lr = max(base_lr * gamma ** sum(1 for s_ in steps if epoch >= s_), min_lr)
    if it < warmup_iters:
        lr *= 1.0 - (1.0 - it / float(warmup_iters)) * (1.0 - warmup_ratio)
    return lr


class Trainer:
    """Training of the heads on the HIP path: ``step(batch)`` = forward_train (losses) + backward + Adagrad update,
    the loop body the reference gets from mmcv's runner + torch.optim (main.py training branch with
    fgn_train_schedule.py:3-13: Adagrad, lr 0.005, weight decay 1e-5, lr_mult 0.1 under ``roi_head``).  The backbone
    is frozen (fgn_r50_c4_densecl.py:31).  Master weights live on the device in torch layout; the packed kernel
    layouts are re-derived from them after every update."""

    def __init__(self, model, lr: float = 0.005, weight_decay: float = 1e-5, roi_head_lr_mult: float = 0.1,
                 eps: float = 1e-10, bn_momentum: float = 0.1, freeze_backbone: bool = False):
        if model.cfg['backbone'].get('norm', 'BN') != 'BN' and not freeze_backbone:
            raise NotImplementedError('the from-scratch configuration (fgn_r50_c4_scratch.py: frozen_stages=-1, GroupNorm) '
                                      'trains its backbone; only the heads have backward kernels - pass '
                                      'freeze_backbone=True to train the heads on the fixed backbone')
        if not torch.cuda.is_available():
            raise ops._lib.FgnHipError('Trainer needs a GPU: the HIP path has no CPU fallback')
        self.model, self.lr, self.wd, self.mult, self.eps = model, lr, weight_decay, roi_head_lr_mult, eps
        # `lr` is the CURRENT (scheduled) learning rate the next step applies - a caller's schedule (``step_lr``) writes it
        # through ``set_lr``; `base_lr` is the schedule's base, what mmcv's LrUpdaterHook keeps as `initial_lr` in every
        # param group of a checkpoint and restores its schedule from on resume
        self.base_lr = lr
        self.bn_momentum = bn_momentum
        dev = torch.device('cuda', torch.cuda.current_device())
        if model._packed_device != dev:
            model._pack(dev)
        self.device = dev
        sd = model._sd
        self.W = {k: sd[k].to(dev).float().contiguous().clone() for k in trainable_names(sd)}
        self.state = {k: torch.zeros_like(v) for k, v in self.W.items()}
        self.buffers = {k: v.to(dev).float().contiguous().clone() for k, v in sd.items()
                        if k.startswith('roi_head.shared_head') and 'running_' in k}
        self.grads: dict = {}
        self.n_steps = 0
        # num_batches_tracked of the shared head's BatchNorms = the loaded counter + the train-mode passes made since
        # (one over the support RoIs per step, one over the sampled RoIs when the step sampled any)
        self._bn_tracked0 = {k: int(v) for k, v in sd.items()
                             if k.startswith('roi_head.shared_head') and k.endswith('num_batches_tracked')}
        self._bn_calls = 0
        import weakref
        model._trainer = weakref.ref(self)      # the model sources its weights from W / buffers while a trainer lives
        self.refresh()

    def set_lr(self, lr: float) -> None:
        """The scheduled learning rate of the next steps (``step_lr(self.base_lr, it, epoch)``); the base stays."""
        self.lr = float(lr)

    def refresh(self) -> None:
        """Re-derive every packed head layer from the master weights (device-side torch ops)."""
        m = self.model
        key = (m.use_winograd, m._packed_device, id(m._P))
        if getattr(self, '_repack_key', None) == key and os.environ.get('FGN_TRAIN_REPACK_IN_PLACE', '1') != '0':
            # every later refresh: the updated weights go INTO the packed layers built below (round 4: 3.8 ms of ~250
            # torch kernels per step -> one strided copy per convolution and one kernel per Winograd layer)
            m._repack_heads_(self.W)
            for blk in m._PT['shared']:
                blk.repack_(self.W)
        else:
            heads = m._pack_heads(self.W)
            for k, v in heads.items():
                m._P[k] = v if not isinstance(v, torch.Tensor) else v.float().contiguous()
            pack_train(m, self.device, self.W, self.buffers)
            self._repack_key = (m.use_winograd, m._packed_device, id(m._P))
        m._shared_dirty = {**self.W, **self.buffers}      # the inference form of the shared head is rebuilt lazily
        m._graphs = {}

    def adopt(self, sd: dict) -> None:
        """Take over weights / running statistics loaded into the model (``FGN.load_state_dict`` during training)."""
        for k in self.W:
            if k in sd:
                self.W[k].copy_(sd[k].to(self.device, torch.float32))
        for k in self.buffers:
            if k in sd:
                self.buffers[k].copy_(sd[k].to(self.device, torch.float32))
        if self.model._packed_device != self.device:
            self.model._pack(self.device)
        self.refresh()

    def forward_backward(self, batch: dict, perm_fn=torch.randperm) -> dict:
        m = self.model
        m._tape = {}
        try:
            with ops.gemm_math('f32'):      # the transposed-weight layers packed inside backward() live for one call
                losses = forward_train(m, perm_fn=perm_fn, bn_momentum=self.bn_momentum, **batch)
                self._bn_calls += 1 + (m._tape.get('roi') is not None)
                g = backward(m, self.W, m._tape)
        finally:
            m._tape = None
        # a gradient for EVERY trainable tensor, zeros where this batch produced none (no positive RoI -> the mask
        # head, no sampled RoI -> the whole RoI stage).  The reference gets zero - not None - gradients there
        # (loss_mask = mask_pred.sum() * 0 on an empty selection), so Adagrad still applies its weight decay; and the
        # data-parallel bucket has the same layout on every rank whatever each rank's batch held.
        self.grads = {k: g[k] if k in g else torch.zeros_like(w) for k, w in self.W.items()}
        return losses

    def step(self, batch: dict, perm_fn=torch.randperm) -> dict:
        """One optimisation step on this rank's batch.  Under an initialised ``torch.distributed`` group (one process
        per GPU, backend "nccl" = RCCL) the gradients are averaged over the ranks first - one all-reduce of one flat
        bucket - so every rank applies the same update (data-parallel training; BatchNorm statistics stay per rank,
        like torch DDP without SyncBN)."""
        losses = self.forward_backward(batch, perm_fn)
        from .dist import allreduce_mean
        self.grads = allreduce_mean(self.grads, keys=sorted(self.W))
        keys = list(self.W)
        if not hasattr(self, '_adagrad_table'):
            self._adagrad_table = {}
        ops.adagrad_multi([self.W[k] for k in keys], [self.grads[k].contiguous() for k in keys],
                          [self.state[k] for k in keys],
                          [self.lr * (self.mult if k.startswith('roi_head') else 1.0) for k in keys], self.wd, self.eps,
                          table=self._adagrad_table)
        self.n_steps += 1
        self.refresh()
        return losses

    def optimizer_state_dict(self) -> dict:
        """``torch.optim.Adagrad.state_dict()`` of the REFERENCE's optimizer, as mmcv saves it (checkpoint_config
        save_optimizer=True, fgn_train_schedule.py:33-38): 'state' {index: {'step', 'sum'}} and one param group per
        parameter over ``reference_param_order`` - every parameter of the detector, the frozen backbone (and the unused
        layer4) included, because mmcv's constructor lists frozen parameters too and Adagrad creates state for all of
        them.  Frozen parameters never see a gradient: their state is torch's initial one (step 0, zero sum) at the
        optimizer's default lr; every group carries mmcv's ``initial_lr`` (the schedule's base x the group's lr_mult)
        next to the current ``lr``.  The reference's ``optimizer.load_state_dict`` accepts this dict as it is (same group
        count, one parameter per group); ``param_names`` (extra key, ignored by torch) spells the index space out."""
        order = reference_param_order(self.model._sd, self.model.cfg['backbone'])
        state, groups = {}, []
        for i, (k, shape) in enumerate(order):
            if k in self.W:
                st = {'step': torch.tensor(float(self.n_steps)), 'sum': self.state[k].detach().cpu()}
                mult = self.mult if k.startswith('roi_head') else 1.0
            else:
                st = {'step': torch.tensor(0.0), 'sum': torch.zeros(shape)}
                mult = 1.0
            state[i] = st
            # 'initial_lr': mmcv's LrUpdaterHook.before_run does `group.setdefault('initial_lr', group['lr'])` and derives
            # every later lr from it - a checkpoint taken at a decayed / warm-up lr without the key would make the resumed
            # schedule start from the decayed value (applied twice)
            groups.append({'lr': self.lr * mult, 'initial_lr': self.base_lr * mult, 'lr_decay': 0, 'eps': self.eps, 'weight_decay': self.wd,
                           'initial_accumulator_value': 0, 'foreach': None, 'maximize': False, 'differentiable': False,
                           'fused': None, 'params': [i]})
        return {'state': state, 'param_groups': groups, 'param_names': [k for k, _ in order]}

    def checkpoint(self, meta: Optional[dict] = None) -> dict:
        """mmcv checkpoint layout: {'state_dict', 'optimizer' (torch Adagrad layout, see ``optimizer_state_dict``),
        'meta'}."""
        return {'state_dict': self.state_dict(), 'meta': dict(meta or {}, iter=self.n_steps),
                'optimizer': self.optimizer_state_dict()}

    def resume(self, ckpt: dict) -> None:
        """Continue from ``checkpoint()``, from a torch / mmcv checkpoint whose 'optimizer' is a
        ``torch.optim.Adagrad.state_dict()`` over the same parameter order, or from a plain state dict."""
        sd = ckpt.get('state_dict', ckpt)
        for k in self.W:
            self.W[k].copy_(sd[k].to(self.device, torch.float32))
        for k in self.buffers:
            if k in sd:
                self.buffers[k].copy_(sd[k].to(self.device, torch.float32))
        for k in self._bn_tracked0:
            if k in sd:
                self._bn_tracked0[k] = int(sd[k])                     # the loaded counter is the base from here on
        self._bn_calls = 0
        if 'iter' in ckpt.get('meta', {}):
            self.n_steps = int(ckpt['meta']['iter'])                  # mmcv's runner resumes its iteration from here
        opt = ckpt.get('optimizer')
        if opt is not None:
            if 'state_sum' in opt:                                    # this package's round-2 layout
                sums = opt['state_sum']
                self.lr, self.wd = opt.get('lr', self.lr), opt.get('weight_decay', self.wd)
                self.base_lr = opt.get('initial_lr', self.lr)
            elif 'state' in opt and 'param_groups' in opt:
                # index -> name: the dict's own `param_names`, else the reference's full parameter order (a checkpoint
                # written by mmcv / torch itself), else - a torch optimizer built over the trainable heads alone -
                # the heads in state-dict order.  Entries of frozen parameters are ignored.
                n_idx = sum(len(g['params']) for g in opt['param_groups'])
                full = [k for k, _ in reference_param_order(self.model._sd, self.model.cfg['backbone'])]
                if 'param_names' in opt:
                    names = list(opt['param_names'])
                elif n_idx == len(full):
                    names = full
                elif n_idx == len(self.W):
                    names = list(self.W)
                else:
                    names = []
                if n_idx != len(names) or not set(self.W) <= set(names):
                    raise ValueError(f'optimizer state covers {n_idx} parameters; the reference model has {len(full)} '
                                     f'({len(self.W)} of them trainable): cannot map indices to parameters '
                                     f'(pass param_names)')
                flat = [i for g in opt['param_groups'] for i in g['params']]
                name_of = {int(i): names[pos] for pos, i in enumerate(flat)}
                sums = {name_of[int(i)]: st['sum'] for i, st in opt['state'].items() if name_of[int(i)] in self.W}
                steps = [float(st['step']) for i, st in opt['state'].items() if name_of[int(i)] in self.W]
                if steps:
                    self.n_steps = int(max(steps))                    # Adagrad's own step count wins over meta
                by_name = {name_of[int(i)]: g for g in opt['param_groups'] for i in g['params']}
                rp = next((g for k, g in by_name.items() if k in self.W and not k.startswith('roi_head')), None)
                if rp is not None:
                    # `lr` of a group is the SCHEDULED lr at the time of the checkpoint; the schedule's base is mmcv's
                    # `initial_lr` (absent in a checkpoint written before any LrUpdaterHook ran: then lr IS the base)
                    self.lr, self.wd = float(rp['lr']), float(rp['weight_decay'])
                    self.base_lr = float(rp.get('initial_lr', rp['lr']))
            else:
                raise ValueError("unknown 'optimizer' entry: expected torch.optim.Adagrad.state_dict() layout")
            for k, v in sums.items():
                self.state[k].copy_(v.to(self.device, torch.float32).view_as(self.state[k]))
        self.refresh()

    def state_dict(self) -> dict:
        """The model's state dict with the trained heads and the updated running statistics (CPU tensors)."""
        sd = dict(self.model._sd)
        for k, v in self.W.items():
            sd[k] = v.detach().cpu()
        for k, v in self.buffers.items():
            sd[k] = v.detach().cpu()
        for k, v0 in self._bn_tracked0.items():       # train-mode passes actually made (support batch + RoI batch)
            sd[k] = torch.tensor(v0 + self._bn_calls, dtype=torch.long)
        return sd
