"""Memory safety of the kernels: the SIMT emulator built with AddressSanitizer (tests/tools/asan_emulated.py) on a few
short, ragged and mid-size cases through every engine of a chain pass.  A read or write outside a library buffer aborts
the run with the kernel's file and line."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not shutil.which("g++"), reason="needs g++ with libasan")
def test_kernels_touch_no_memory_outside_their_buffers():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "asan_emulated.py"), "16", "0"],
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "no memory error reported" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    assert "AddressSanitizer" not in out.stderr


@pytest.mark.skipif(not shutil.which("g++"), reason="needs g++ with libasan")
def test_periodic_form_copies_nothing_after_an_escaped_walk():
    """Regression (round 4): the plane automaton escapes inside the SECOND walked period of a region -- it publishes nothing,
    the counters stay where the first period left them, and "the same d = 0 on every leaf" used to pass for a verdict: the
    outputs of steps nobody had written were replicated and counted, node ids out of whatever the buffer held (in bounds on
    the device by luck, a wild atomicAdd under the sanitizer).  k_period_verdict now refuses when a chain has escaped."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "asan_emulated.py"), "4", "49000"],
                         capture_output=True, text=True, timeout=1500)
    assert out.returncode == 0 and "no memory error reported" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    assert "AddressSanitizer" not in out.stderr
