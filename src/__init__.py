"""Drop-in `src` package: the reference's train.py / evaluate.py import `src.models`, `src.losses`,
`src.components.pulpo`, `src.network_blocks`, `src.utils` by name; these modules re-export the MI355X
implementation in pulpo_amd under those names.

The package path is EXTENDED, not replaced: sub-packages this repository does not provide (`src.data.OASIS`, `src.data.BraTS`, imported
by train.py:7-8 and evaluate.py:19-20) resolve from the reference's own `src/` directory when it is further down sys.path, so putting
this repository's root ahead of the reference on PYTHONPATH is enough (option 1)."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
