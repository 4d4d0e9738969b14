import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


_MODULE_PEAKS = []


@pytest.fixture(scope="module", autouse=True)
def _release_device_memory(request):
    """The full-size modules hold tens of GiB per model (workspace with the conv tables, logits of replayed rollouts):
    once a module is done, drop what its fixtures kept and hand the cached blocks back, so the next module starts
    from an empty card whatever the collection order."""
    yield
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
            _MODULE_PEAKS.append((request.module.__name__, torch.cuda.max_memory_allocated() / 2 ** 30,
                                  torch.cuda.max_memory_reserved() / 2 ** 30))
            torch.cuda.empty_cache()
            torch.cuda.reset_peak_memory_stats()
    except ImportError:
        pass


def pytest_terminal_summary(terminalreporter):
    """One line on how close each GPU test module came to the card's memory (the full-size modules allocate 25-90 GiB
    per model; blocks the caching allocator holds are handed back to the driver before it reports out-of-memory)."""
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            terminalreporter.write_line("peak device memory in use / held by the caching allocator, GiB of %.0f: %s"
                                        % (torch.cuda.get_device_properties(0).total_memory / 2 ** 30,
                                           ", ".join("%s %.0f / %.0f" % (n.replace("test_gpu_", ""), a, r)
                                                     for n, a, r in _MODULE_PEAKS if r > 1)))
    except Exception:
        pass
