"""Drop-in `src` package: the reference's train.py / evaluate.py import `src.models`, `src.losses`,
`src.components.pulpo`, `src.network_blocks`, `src.utils` by name; these modules re-export the MI355X
implementation in pulpo_amd under those names."""
