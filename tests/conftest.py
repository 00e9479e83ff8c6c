import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


GOLDEN = os.path.join(ROOT, "tests", "golden")
FIXTURES = ["nucleic_gtr_g4", "proteic_lg_g4", "nucleic_gtr_g4_inv", "nucleic_zero_w", "synth_nt_300x40", "synth_aa_90x24"]


@pytest.fixture(scope="session")
def golden():
    import phyg
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = phyg.load(os.path.join(GOLDEN, name + ".phyg"))
        return cache[name]
    return get
