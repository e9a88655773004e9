import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _cpu_threads():
    """The CPU oracle is the checker in most tests.  torch defaults to one thread per VISIBLE core (256 on the GPU boxes)
    while the cgroup grants 16: the oversubscribed oracle ran 10x slower there (470 s for the headline test instead of
    ~45 s).  Pin the pool to the cores this process may really use (bench.usable_cores)."""
    import torch
    from bench import usable_cores
    torch.set_num_threads(min(usable_cores(), 64))
    yield


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]

    return load
