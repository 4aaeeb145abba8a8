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

        # ---- the hooks of pytorch_lightning 1.8's automatic optimization, with its default bodies (core/module.py, core/hooks.py)
        def on_train_batch_start(self, batch, batch_idx) -> None:
            pass

        def on_train_batch_end(self, outputs, batch, batch_idx) -> None:
            pass

        def on_before_zero_grad(self, optimizer) -> None:
            pass

        def optimizer_zero_grad(self, epoch, batch_idx, optimizer, optimizer_idx=0) -> None:
            optimizer.zero_grad()

        def on_before_backward(self, loss) -> None:
            pass

        def backward(self, loss, optimizer=None, optimizer_idx=0, *args, **kwargs) -> None:
            loss.backward(*args, **kwargs)

        def on_after_backward(self) -> None:
            pass

        def on_before_optimizer_step(self, optimizer, optimizer_idx=0) -> None:
            pass

        def optimizer_step(self, epoch, batch_idx, optimizer, optimizer_idx=0, optimizer_closure=None, on_tpu=False, using_native_amp=False,
                           using_lbfgs=False) -> None:
            optimizer.step(closure=optimizer_closure)

        @classmethod
        def load_from_checkpoint(cls, path, map_location=None, **kwargs):
            import torch
            ckpt = torch.load(path, map_location=map_location or "cpu", weights_only=True)
            hp = dict(ckpt.get("hyper_parameters", {}))
            hp.update(kwargs)
            model = cls(**hp)
            model.load_state_dict(ckpt["state_dict"])
            return model


class HookOrderTrainer:
    """The training loop of pytorch_lightning 1.8.1 (the reference's pinned version, package-list.txt:135) reduced to the order in which it
    calls a LightningModule during `Trainer.fit` with automatic optimization and one optimizer (loops/optimization/optimizer_loop.py,
    loops/optimization/closure.py, plugins/precision/precision_plugin.py, strategies/strategy.py):

        on_train_batch_start
        optimizer_step(epoch, batch_idx, optimizer, 0, closure)            -> optimizer.step(closure=closure)
            closure:  training_step(batch, batch_idx)
                      on_before_zero_grad(optimizer); optimizer_zero_grad(epoch, batch_idx, optimizer, 0)
                      on_before_backward(loss); backward(loss, optimizer, 0); on_after_backward()
                      on_before_optimizer_step(optimizer, 0)               (precision plugin, after the closure, before the update)
        on_train_batch_end

    pytorch_lightning is not installed in this image, so tests and `bench.py --loop lightning` drive pulpo_amd.models.PULPo through this
    order; with Lightning installed the real Trainer makes the same calls (train.py:106-116).  Not a Trainer: no loggers, callbacks,
    validation, checkpoints."""

    def __init__(self) -> None:
        self.should_stop = False
        self.global_step = 0
        self.current_epoch = 0
        self.num_val_batches = [0]
        self.calls = []                     # hook names in call order (tests)

    def attach(self, model):
        model.trainer = self
        self.model = model
        self.optimizer = model.configure_optimizers()
        return self.optimizer

    def _call(self, name, *args, **kwargs):
        self.calls.append(name)
        return getattr(self.model, name)(*args, **kwargs)

    def run_batch(self, batch, batch_idx: int = 0):
        m, opt = self.model, self.optimizer
        result = {}
        self._call("on_train_batch_start", batch, batch_idx)

        def closure():
            loss = self._call("training_step", batch, batch_idx)
            self._call("on_before_zero_grad", opt)
            self._call("optimizer_zero_grad", self.current_epoch, batch_idx, opt, 0)
            self._call("on_before_backward", loss)
            self._call("backward", loss, opt, 0)
            self._call("on_after_backward")
            result["loss"] = loss.detach()
            self._call("on_before_optimizer_step", opt, 0)
            return result["loss"]

        self._call("optimizer_step", self.current_epoch, batch_idx, opt, 0, closure)
        self.global_step += 1
        self._call("on_train_batch_end", result, batch, batch_idx)
        return result["loss"]

    def fit(self, model, batches):
        if getattr(self, "model", None) is not model:
            self.attach(model)
        model.train()
        out = None
        for i, batch in enumerate(batches):
            out = self.run_batch(batch, i)
            if self.should_stop:
                break
        return out
