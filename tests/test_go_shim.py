"""The Go shim cannot be compiled here (no Go toolchain): keep at least its struct field list and the
entry points it calls in step with include/blance_hip.h, its files balanced, and free of the compile errors a
blind edit is most likely to leave behind (tools/go_lint.py: unused locals and imports, := without a new name)."""
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


def test_go_files_pass_the_lint():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "go_lint.py")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_the_lint_finds_what_it_is_for(tmp_path):
    bad = tmp_path / "bad.go"
    bad.write_text("""package blance

import (
\t"fmt"
\t"sort"
)

func f(a []int) int {
\tx := 1
\ty := 2
\tx := 3
\tfor i, v := range a {
\t\tfmt.Println(v)
\t}
\tz := strconv.Itoa(x)
\treturn len(z)
}
""")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "go_lint.py"), str(bad)], capture_output=True, text=True)
    assert out.returncode == 1
    for needle in ("'sort' imported and not used", "'strconv' used without import", "'y' declared and not used",
                   "'i' declared and not used", "no new variables on left side of := ('x')"):
        assert needle in out.stdout, (needle, out.stdout)
    bad.write_text("package blance\n\nfunc g() int {\n\treturn (1\n}\n")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "go_lint.py"), str(bad)], capture_output=True, text=True)
    assert out.returncode == 1 and "does not close" in out.stdout
