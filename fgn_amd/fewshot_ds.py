"""Synthetic episodic dataset with the BaseFewShotISEG interface.

The reference's datasets (datasets/fewshotiseg/base_fst.py) need cv2/imgaug and real
images; the hot path only depends on their *sample dict* (base_fst.py:1248-1266) and on
``evaluate`` (base_fst.py:1516-1601).  This class produces that contract from seeded
synthetic episodes so the reference's evaluation loop (main.py:269-326) runs unchanged
on top of ``fgn_amd.detector.FGN``; visualisation side effects of ``evaluate`` are not
reproduced.
"""
from __future__ import annotations

import os
import pickle

from torch.utils.data import Dataset

from . import episodes
from .fsiseg_eval import FSISEGEval


class SyntheticFewShotISEG(Dataset):
    def __init__(self, n_ways=3, k_shots=3, length=64, height=800, width=1333, spp_img_size=256, batch=4,
                 seed=1234, suffix='SYNTH_val_novel'):
        self.n_ways, self.k_shots = n_ways, k_shots
        self.length, self.batch = length, batch
        self.height, self.width, self.spp_img_size = height, width, spp_img_size
        if batch > 1:       # the reference rounds batched sizes to multiples of 16 (base_fst.py:693-694)
            self.height, self.width = height // 16 * 16, width // 16 * 16
        self.seed, self.suffix = seed, suffix
        self.sampling_origin_ds, self.sampling_origin_ds_subset = 'SYNTH', 'val'
        self.sampling_cats, self.sampling_scenario, self.finetune = 'novel', 'parents', 'None'

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        return episodes.make_episode(int(idx), self.n_ways, self.k_shots, self.height, self.width,
                                     self.spp_img_size, seed=self.seed)

    def reshuffle(self):
        self.seed += self.length

    def evaluate(self, results=None, results_pkl_dir_fp=None, model_dir=None, total=1):
        """Same return keys as BaseFewShotISEG.evaluate (base_fst.py:1597-1601)."""
        assert (results is not None) ^ (results_pkl_dir_fp is not None)
        out = {}
        for kind, key in (('segm', 'isegm'), ('bbox', 'bbox')):
            ev = FSISEGEval(results=results, results_pkl_dir_fp=results_pkl_dir_fp, n_ways=self.n_ways,
                            iou_type=kind).run()
            out[f'{key}_mAP'], out[f'{key}_mAR'] = ev['mAP'], ev['mAR']
        return out


def write_chunked(results_iter, out_dir, chunk=1000):
    """main.py:290-309: results pickled in chunks of 1000 as ResultsChunked/NN.pkl."""
    os.makedirs(out_dir, exist_ok=True)
    buf, counter = [], 0
    for res in results_iter:
        buf.extend(res)
        if len(buf) >= chunk:
            with open(os.path.join(out_dir, f'{counter:02}.pkl'), 'wb') as fh:
                pickle.dump(buf, fh)
            buf, counter = [], counter + 1
    if buf:
        with open(os.path.join(out_dir, f'{counter:02}.pkl'), 'wb') as fh:
            pickle.dump(buf, fh)
