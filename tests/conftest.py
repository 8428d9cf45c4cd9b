import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _gpu_run(config):
    m = getattr(config.option, "markexpr", "") or ""
    return "gpu" in m and "not gpu" not in m


def _cpu_workers(config):
    """The CPU suite (`-m "not gpu"`) on a few pytest-xdist workers when that plugin is there and the caller gave no -n:
    the emulated kernels are single threaded, the box has idle cores.  GPU runs stay in one process (one device, and the
    driver records which libraries THAT process mapped).  BLANCE_TESTS_WORKERS=0 keeps the CPU suite serial too."""
    if hasattr(config, "workerinput") or not config.pluginmanager.hasplugin("xdist"):
        return 0
    if getattr(config.option, "numprocesses", None) is not None or _gpu_run(config):
        return 0
    if getattr(config.option, "usepdb", False) or getattr(config.option, "collectonly", False):
        return 0
    want = os.environ.get("BLANCE_TESTS_WORKERS")
    if want is not None:
        return max(0, int(want))
    return max(0, min(4, (os.cpu_count() or 1) // 2))


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    n = _cpu_workers(config)
    if n > 1:
        config.option.numprocesses = n          # what `-n <n>` sets (xdist.plugin.pytest_cmdline_main does the same)
        config.option.dist = "load"
        config.option.tx = ["popen"] * n
        config._blance_prebuild = True


def _prebuild():
    """Every native piece a CPU test builds when it is stale, built ONCE here in the controlling process before the
    workers start -- two workers never compile the same target."""
    import __graft_entry__ as g
    g.build()                                   # HIP library (cross-compiled), oracle, host mirror, wire codec, move index
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    from test_simt_emulated import build_emu
    build_emu()
    import asan_emulated
    asan_emulated.build()
    from oracle import naive_loader
    if hasattr(naive_loader, "build"):
        naive_loader.build()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    if getattr(config, "_blance_prebuild", False) and not hasattr(config, "workerinput"):
        _prebuild()


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
