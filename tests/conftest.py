import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    z = np.load(os.path.join(HERE, "golden", "mcraw_golden.npz"))
    cases = {}
    for name in z["names"]:
        name = str(name)
        typ, w, h, ret = (int(v) for v in z[name + "/meta"])
        cases[name] = dict(type=typ, w=w, h=h, ret=ret, buf=z[name + "/buf"], out=z[name + "/out"])
    return cases


@pytest.fixture(scope="session")
def gpu_ctx():
    """One decode context on cuda:0 for the GPU tests (fails loudly without the HIP library)."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    import motioncam_decoder_amd as M
    ctx = M.Context(0)
    yield ctx
    ctx.close()
