from pulpo_amd.eval_metrics import resize_dfs, warp_landmarks  # noqa: F401  (reference src/components/utils.py)
