import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def golden_cases():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "planner_cases.json")) as f:
        return [c for c in json.load(f)["cases"] if not c.get("ignored")]


@pytest.fixture(scope="session")
def helper_tables():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "helper_cases.json")) as f:
        return json.load(f)["tables"]
