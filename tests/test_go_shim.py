"""The Go shim cannot be compiled here (no Go toolchain): keep at least its struct field list and the
entry points it calls in step with include/blance_hip.h, and its files balanced."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_go_shim_matches_header():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_go_shim.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_go_files_are_balanced():
    for name in ("intern.go", "plan_hip.go", "moves_hip.go", "hip_test.go", "orchestrate_index.go"):
        text = open(os.path.join(ROOT, "go", "blance", name)).read()
        assert text.count("{") == text.count("}"), name
        assert text.count("(") == text.count(")"), name
        assert "package blance" in text
