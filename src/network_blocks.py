from pulpo_amd.network_blocks import *  # noqa: F401,F403
from pulpo_amd.network_blocks import (ConvSequence, ConvUnit, DFAdder, FixedNoiseSampler, MuSigmaBlock, ResizeTransform,  # noqa: F401
                                      SpatialTransformer, VecInt, VelocityField, gauss_sampler)
