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


@pytest.fixture(scope="module", autouse=True)
def _release_device_memory():
    """The full-size modules hold tens of GiB per model (workspace with the conv tables, logits of replayed rollouts):
    once a module is done, drop what its fixtures kept and hand the cached blocks back, so the next module starts
    from an empty card whatever the collection order."""
    yield
    import gc
    gc.collect()
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
    except ImportError:
        pass


def pytest_terminal_summary(terminalreporter):
    """One line on how close the GPU suite came to the card's memory (the full-size modules allocate 25-90 GiB each)."""
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            terminalreporter.write_line("peak device memory reserved by this process: %.1f GiB of %.0f"
                                        % (torch.cuda.max_memory_reserved() / 2 ** 30,
                                           torch.cuda.get_device_properties(0).total_memory / 2 ** 30))
    except Exception:
        pass
