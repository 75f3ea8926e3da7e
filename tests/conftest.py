import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _frame_buffers_given_back(request):
    """GPU tests cache their scenes, and a scene keeps the frame buffers of every stream it rendered on (a 4K frame: gigabytes).
    Once more than 6 GiB of the device are in use, every live scene gives its frame buffers back (rt_scene_trim); the scenes
    themselves stay uploaded and the next test's frames allocate again."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    import torch
    free, total = torch.cuda.mem_get_info()
    if total - free > 6 * 2**30:
        from raytracer_amd import api
        api.trim_all()


def pytest_sessionfinish(session, exitstatus):
    """GPU runs: device memory still held when the session ends (cached scenes keep their working buffers), for the record."""
    try:
        import torch
        if torch.cuda.is_available():
            free, total = torch.cuda.mem_get_info()
            print(f"\ndevice memory in use at session end: {(total - free) / 2**30:.1f} of {total / 2**30:.1f} GiB")
    except Exception:
        pass
