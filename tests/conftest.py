import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """npz fixture -> dict of numpy arrays (allow_pickle stays False)"""
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get


@pytest.fixture(autouse=True)
def _clean_kernel_switches():
    """Every test starts from the library's default kernel selection (ops.CONV_ALGO None = per-shape choice, fp32 operands) and must
    leave it that way: a switch leaking out of one test would silently change which kernel the later golden comparisons exercise."""
    mod = sys.modules.get("pulpo_amd.ops")
    fusion = None
    if mod is not None:
        assert mod.CONV_ALGO is None, f"ops.CONV_ALGO leaked from an earlier test: {mod.CONV_ALGO!r}"
        assert mod.CONV_PRECISION == "fp32", f"ops.CONV_PRECISION leaked from an earlier test: {mod.CONV_PRECISION!r}"
        assert mod.ACT_BF16 is False, "ops.ACT_BF16 leaked from an earlier test"
        fusion = mod.BN_REDUCE_IN_DGRAD
        want_det = os.environ.get("PULPO_DETERMINISTIC", "0") == "1"
        assert mod.DETERMINISTIC == want_det, "ops.DETERMINISTIC leaked from an earlier test"
    yield
    mod = sys.modules.get("pulpo_amd.ops")
    if mod is not None:
        leaked = (mod.CONV_ALGO, mod.CONV_PRECISION, mod.ACT_BF16)
        mod.CONV_ALGO = None
        mod.set_conv_precision("fp32")
        assert leaked == (None, "fp32", False), f"test left ops.CONV_ALGO / CONV_PRECISION / ACT_BF16 = {leaked!r}"
        det_left = mod.DETERMINISTIC
        mod.set_deterministic(os.environ.get("PULPO_DETERMINISTIC", "0") == "1")
        assert det_left == mod.DETERMINISTIC, "test left ops.DETERMINISTIC switched"
    orc = sys.modules.get("oracle.pulpo_oracle")
    if orc is not None:
        left = (orc.CONV_PRECISION, orc.ACT_PRECISION)
        orc.CONV_PRECISION, orc.ACT_PRECISION = "fp32", "fp32"
        assert left == ("fp32", "fp32"), f"test left the oracle in {left!r}"
        if fusion is not None:
            now = mod.BN_REDUCE_IN_DGRAD
            mod.BN_REDUCE_IN_DGRAD = fusion
            assert now == fusion, f"test left ops.BN_REDUCE_IN_DGRAD = {now!r}"
