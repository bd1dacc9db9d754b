import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _fresh_rasterizer_policy(request):
    """GPU tests start from the rasterizer's default capacity policy ("auto") with nothing learnt about any scene: what one test's
    scene taught the drop-in path (longest list per image size) must not steer the next test's first render."""
    if "gpu" not in request.keywords:
        yield
        return
    from splatam_amd import rasterizer as rz
    rz.set_sync_mode("auto")
    rz.reset_scene_stats()
    rz.USE_TILE_RECS = None
    yield
    rz.set_sync_mode("auto")
    rz.reset_scene_stats()
    rz.USE_TILE_RECS = None
