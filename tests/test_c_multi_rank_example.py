"""examples/c_multi_rank.c — config 5's placement arithmetic and the multi-rank container's front matter from plain C (gcc against include/density_hip.h and
libdensity_hip.so).  Pure host arithmetic: runs without a GPU."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_multi_rank_example_builds_and_runs(tmp_path):
    from density_amd import build
    build.build()
    exe = str(tmp_path / "c_multi_rank")
    libdir = os.path.join(ROOT, "density_amd")
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_multi_rank.c"),
                    "-L" + libdir, "-ldensity_hip", "-o", exe], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "multi-rank layout ok" in r.stdout, r.stdout + r.stderr
    assert r.stdout.count("blob ") == 4
