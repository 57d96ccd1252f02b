import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def snapshot_names():
    d = os.path.join(GOLDEN, "snapshots")
    return sorted(f[:-5] for f in os.listdir(d) if f.endswith(".json"))


def load_snapshot(name):
    with open(os.path.join(GOLDEN, "snapshots", name + ".json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import walker
    walker.build()
    return walker
