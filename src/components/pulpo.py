from pulpo_amd.components.pulpo import Autoencoder, DownPath, PULPoEncoder, PULPoPrior, SVFDecoder  # noqa: F401
