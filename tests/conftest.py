import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
Q16 = 65537
Q61 = 2305843009211596801


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import load_oracle

    return load_oracle()


@pytest.fixture(scope="session")
def pkg():
    import fhe_study_amd

    if fhe_study_amd.binding.needs_build():
        fhe_study_amd.build()
    fhe_study_amd.load_library()
    return fhe_study_amd


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


def golden_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}
