import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: takes more than ~20 s on CPU")


def golden_cases():
    # one reference run per file; auc.npz, the whole-config result tables (full_c<config>.npz) and the final
    # distributions (dist_<case>.npz, a companion of <case>.npz) are other schemas
    return sorted(f[:-4] for f in os.listdir(GOLDEN)
                  if f.endswith(".npz") and f not in ("auc.npz", "frontend.npz", "caffe_proto.npz") and not f.startswith(("full_c", "dist_", "instability", "rasteralt")))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
