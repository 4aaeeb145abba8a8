"""PULPo registration hot path on AMD Instinct MI355X (gfx950): hand-written HIP kernels behind the reference API."""
