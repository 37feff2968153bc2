"""The C ABI from C: examples/c_roundtrip.c compiled with gcc against include/density_hip.h and libdensity_hip.so and run on the GPU box —
the reference's nine symbols and the container entry points, the way a C or FFI caller reaches them."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_example_builds_and_round_trips(tmp_path):
    exe = str(tmp_path / "c_roundtrip")
    libdir = os.path.join(ROOT, "density_amd")
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_roundtrip.c"),
                    "-L" + libdir, "-ldensity_hip", "-o", exe], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0 and "all round trips ok" in r.stdout, r.stdout + r.stderr
