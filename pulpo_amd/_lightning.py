"""LightningModule base for pulpo_amd.models.PULPo.

When pytorch_lightning is installed (the reference's train.py / evaluate.py need it anyway) PULPo derives from the
real pl.LightningModule.  In images without it (this build image, the GPU test boxes) a small nn.Module base with the
handful of hooks PULPo itself calls keeps the model constructible, steppable and checkpointable.
"""
from __future__ import annotations

import inspect
from typing import Any, Dict

import torch.nn as nn

try:  # pragma: no cover - depends on the environment
    import pytorch_lightning as pl
    LightningModule = pl.LightningModule
    HAVE_LIGHTNING = True
except Exception:  # ModuleNotFoundError, or a lightning build that cannot import here
    HAVE_LIGHTNING = False

    class _AttrDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError as e:
                raise AttributeError(k) from e

        def __setattr__(self, k, v):
            self[k] = v

    class _NoTrainer:
        should_stop = False
        global_step = 0
        num_val_batches = [0]

    class LightningModule(nn.Module):
        """the subset of pl.LightningModule that PULPo uses"""

        def __init__(self) -> None:
            super().__init__()
            self._hparams = _AttrDict()
            self.logged: Dict[str, Any] = {}
            self.trainer = _NoTrainer()
            self.logger = None

        @property
        def hparams(self):
            return self._hparams

        def save_hyperparameters(self) -> None:
            frame = inspect.currentframe().f_back
            params = inspect.signature(type(self).__init__).parameters
            for name in params:
                if name != "self" and name in frame.f_locals:
                    self._hparams[name] = frame.f_locals[name]

        def log_dict(self, d: Dict[str, Any], **kwargs) -> None:
            self.logged.update(d)          # kept as device tensors: logging never synchronises the stream

        def log(self, name: str, value: Any, **kwargs) -> None:
            self.logged[name] = value

        @classmethod
        def load_from_checkpoint(cls, path, map_location=None, **kwargs):
            import torch
            ckpt = torch.load(path, map_location=map_location or "cpu", weights_only=True)
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(kwargs)
            model = cls(**hp)
            model.load_state_dict(ckpt["state_dict"])
            return model
