import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "lightweight-multi-modal-scene-understanding-via-knowledge-distillation_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# GPU parity modules opt in with `pytest.mark.usefixtures("gemm_arith")`: every test then runs once per GEMM
# arithmetic of the library ("split": bf16x6 piece products on the bf16 matrix pipe, the default; "fp32": exact-fp32
# MFMA products).  Both must meet the same tolerances.
def pytest_generate_tests(metafunc):
    if "gemm_arith" in metafunc.fixturenames:
        metafunc.parametrize("gemm_arith", ["split", "fp32"], indirect=True)


@pytest.fixture
def gemm_arith(request):
    from kdrt import ops
    prev = ops.set_gemm_arithmetic(request.param)
    yield request.param
    ops.set_gemm_arithmetic(prev)
