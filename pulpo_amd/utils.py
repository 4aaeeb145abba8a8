"""Integer-keyed ModuleDict (drop-in for the reference's src/utils.py:26-48).

nn.ModuleDict only takes string keys; PULPo indexes its per-level blocks with ints, and the resulting state-dict key
names ('down_blocks.0....', 'encoders.3....') must stay the same for checkpoints to load.
"""
from __future__ import annotations

from typing import Iterator, Mapping, Optional, Tuple

import torch.nn as nn


class ModuleIntDict(nn.ModuleDict):
    def __init__(self, modules: Optional[Mapping[int, nn.Module]] = None) -> None:
        super().__init__()
        if modules is not None:
            for key, mod in modules.items():
                self[key] = mod

    def __getitem__(self, key: int) -> nn.Module:
        return super().__getitem__(str(key))

    def __setitem__(self, key: int, module: nn.Module) -> None:
        super().__setitem__(str(key), module)

    def __delitem__(self, key: int) -> None:
        super().__delitem__(str(key))

    def __contains__(self, key) -> bool:
        return super().__contains__(str(key))

    def keys(self) -> Iterator[int]:
        return (int(k) for k in super().keys())

    def items(self) -> Iterator[Tuple[int, nn.Module]]:
        return ((int(k), m) for k, m in super().items())
