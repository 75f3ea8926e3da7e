import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionfinish(session, exitstatus):
    """GPU runs: device memory still held when the session ends (cached scenes keep their working buffers), for the record."""
    try:
        import torch
        if torch.cuda.is_available():
            free, total = torch.cuda.mem_get_info()
            print(f"\ndevice memory in use at session end: {(total - free) / 2**30:.1f} of {total / 2**30:.1f} GiB")
    except Exception:
        pass
