"""Monte-Carlo uncertainty of a registration (BASELINE config 5: "8-sample MC"), as `Evaluate` computes it in the reference
(evaluate.py:222-251, 3-D branch): `num_samples` stochastic `model.predict(x, y, N=1)` passes, the mean individual fields ->
combined / final fields -> warped image, and per-voxel sample standard deviations of the warped image, the individual and the
final fields (mean over the channel axis).  The reference stores every sample ((N, C, D, H, W) per level and quantity); here each
sample is folded into running moments by one streaming kernel, so memory does not grow with N.
The reference has no dropout: the randomness is the latent sampling."""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops


@torch.no_grad()
def mc_uncertainty(model, x: torch.Tensor, y: torch.Tensor, num_samples: int, mask_x: Optional[torch.Tensor] = None,
                   mean_of_samples: bool = False) -> Dict[str, Dict[int, torch.Tensor]]:
    """x, y: (1, 1, D, H, W) moving / fixed volumes (evaluate.py runs batch size 1).  Returns the dictionaries of evaluate.py:
    outputs, individual_dfs, combined_dfs, final_dfs (from the sample-mean individual fields) and output_std, individual_df_std,
    final_df_std ((D, H, W) per level; final_df_std is masked by the warped `mask_x` when one is given, evaluate.py:246-249).

    Reference quirk kept by default: evaluate.py:239 averages `individual_dfs` - the dictionary returned by the LAST predict() call,
    over its batch axis of size 1 - not `all_individual_dfs`, so the "average" fields, and everything derived from them, are the
    last sample's.  mean_of_samples=True uses the mean over the N samples (what the comment in the reference says)."""
    if num_samples < 1:
        raise ValueError("mc_uncertainty: num_samples must be >= 1")
    L = model.latent_levels
    m_out = {l: ops.StreamingMoments() for l in range(L)}
    m_ind = {l: ops.StreamingMoments() for l in range(L)}
    m_fin = {l: ops.StreamingMoments() for l in range(L)}
    individual = None
    # In eval mode the encoder pyramid is a deterministic function of (x, y): BatchNorm uses its running statistics and the latent
    # noise only enters the autoencoder.  evaluate.py recomputes it for every sample; here it is computed once and each sample runs
    # only the stochastic half (identical results, ~40 % less work per sample).  In training mode fall back to model.predict().
    down = model.downpath(x, y) if not model.training else None
    for _ in range(num_samples):
        if down is not None:
            outs = model.autoencoder(x, down)
            individual = outs[4]                               # predict(N=1): the mean over one sample is the sample
            _, final = model.combine_dfs(individual)
            outputs = {l: model.autoencoder.decoders[l].spatial_transform(final[l], x) for l in final}
        else:
            outputs, individual = model.predict(x, y, N=1)
            _, final = model.combine_dfs(individual)
        for l in range(L):
            m_out[l].update(outputs[l])
            m_ind[l].update(individual[l])
            m_fin[l].update(final[l])
    individual_dfs = {l: (m_ind[l].mean() if mean_of_samples else individual[l].mean(dim=0).unsqueeze(0)) for l in range(L)}
    combined_dfs, final_dfs = model.combine_dfs(individual_dfs)
    warp = lambda l, img: model.autoencoder.decoders[l].spatial_transform(final_dfs[l], img)
    outputs = {l: warp(l, x) for l in range(L)}
    res = {"outputs": outputs, "individual_dfs": individual_dfs, "combined_dfs": combined_dfs, "final_dfs": final_dfs,
           "output_std": {l: m_out[l].std_map()[0] for l in range(L)},
           "individual_df_std": {l: m_ind[l].std_map()[0] for l in range(L)}}
    if mask_x is not None:
        res["final_df_std"] = {l: m_fin[l].std_map(scale=warp(l, mask_x))[0] for l in range(L)}
    else:
        res["final_df_std"] = {l: m_fin[l].std_map()[0] for l in range(L)}
    return res
