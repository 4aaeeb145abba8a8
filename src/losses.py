from pulpo_amd.losses import *  # noqa: F401,F403
from pulpo_amd.losses import (HierarchicalKLLoss, HierarchicalReconstructionLoss, HierarchicalRegularization, JDetStd, KL_nondiagonal,  # noqa: F401
                              KL_two_gauss_with_diag_cov, L2_loss, L2_reg, NCC_loss, Soft_dice_loss, jacobian_det)
