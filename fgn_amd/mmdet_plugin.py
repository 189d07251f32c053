"""Registry shim: ``import fgn_amd.mmdet_plugin`` registers ``fgn_amd.detector.FGN`` in mmdet's ``DETECTORS`` registry
under the reference's name ``'FGN'`` (``force=True``: it replaces the class the reference registers at fgn.py:28), so
``build_detector(cfg.model, train_cfg=..., test_cfg=...)`` (main.py:390-394) builds the MI355X detector from the
reference's own config WITHOUT an edit to main.py - the import rides in the config file, where mmcv's ``Config``
executes it while loading:

    # fgn_r50_c4_densecl.py (or the experiment config that inherits it)
    custom_imports = dict(imports=['fgn_amd.mmdet_plugin'], allow_failed_imports=False)

Conditional by construction: without mmdet nothing happens (``register()`` returns False) and ``fgn_amd`` itself never
imports this module - the detector has no mmdet dependency.  What the registry hands to the class is exactly the
reference's constructor call, ``FGN(n_ways=, k_shots=, backbone=, rpn_head=, roi_head=, train_cfg=, test_cfg=)``
(fgn.py:41-52), which ``fgn_amd.detector.FGN`` accepts as it is.  What main.py does to the built object next and what
answers it here: ``model.backbone.frozen_stages`` / ``.res_layers`` / ``.eval()`` (main.py:402-405) -> ``FGN.backbone``
(a view of the backbone config: the C4 truncation is how this detector is built in the first place); ``model.cfg = cfg``,
``model.to(device)``, ``model.eval()`` -> plain ``nn.Module`` behaviour; ``model.init_weights()`` (main.py:431) ->
``FGN.init_weights``.  The mmcv runner / optimizer / hooks around the loop (main.py:412-454) are the reference's control
plane and stay out of scope: the training step on this path is ``fgn_amd.train.Trainer``."""
from __future__ import annotations


def register(force: bool = True) -> bool:
    """Register the detector with mmdet if mmdet is importable.  Returns True when the registry now builds
    ``fgn_amd.detector.FGN`` for ``type='FGN'``."""
    try:
        from mmdet.models.builder import DETECTORS
    except Exception:           # mmdet (or one of its own imports) is absent: nothing to register with
        return False
    from .detector import FGN
    DETECTORS.register_module(name='FGN', force=force, module=FGN)
    return True


REGISTERED = register()
