"""The GPU run scripts, probes' summaries and the bench harness are only ever executed on a GPU box: at least they must parse here (a syntax error in
one of them would cost a GPU call to find), and the shell scripts must name files that exist."""
import os
import py_compile
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY = [os.path.join(d, f) for sub in ("tools", "tools/round4", "probes", "benches", ".") for d in [os.path.join(ROOT, sub)] if os.path.isdir(d)
      for f in sorted(os.listdir(d)) if f.endswith(".py")]
SH = [os.path.join(d, f) for sub in ("tools", "tools/round4", "probes") for d in [os.path.join(ROOT, sub)] if os.path.isdir(d)
      for f in sorted(os.listdir(d)) if f.endswith(".sh")]


@pytest.mark.parametrize("path", PY, ids=lambda p: os.path.relpath(p, ROOT))
def test_python_script_parses(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "x.pyc"), doraise=True)


@pytest.mark.parametrize("path", [p for p in SH if "/round4/" not in p and os.path.basename(p) in ("gpu_round_end5.sh", "profile_round.sh", "profile_directions.sh")],
                         ids=lambda p: os.path.relpath(p, ROOT))
def test_recipe_scripts_name_existing_files(path):
    text = open(path).read()
    for rel in set(re.findall(r"(?:python|bash) ((?:tools|probes|benches|tests)/[\w./-]+\.(?:py|sh))", text)):
        assert os.path.exists(os.path.join(ROOT, rel)), (path, rel)
