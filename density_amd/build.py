"""Builds libdensity_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdensity_hip.so")
LIB_DEBUG = os.path.join(HERE, "libdensity_hip_debug.so")      # -DDENSITY_HIP_DEBUG: the only build that reads tuning / diagnostic switches from the environment
SOURCES = ["api.hip", "api_stream.hip", "api_host.hip", "chameleon.hip", "rotor.hip", "container.hip", "serial_codec.hip", "stream_parse.hip", "exchange_stages.hip", "decode_passes.hip", "placement.hip"]
HEADERS = ["api_internal.hpp", "common.hpp", "chameleon_dev.hpp", "kernels.hpp", os.path.join("..", "..", "include", "density_hip.h")]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def kernels_id():
    """A short hash of the kernel sources: density_hip_version() carries it, and measurements that belong to one kernel generation (the PMC
    traffic in profiles/r*_pmc_summary.json) are quoted by bench.py only for the library they were taken on."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(f for f in SOURCES + HEADERS if not os.path.basename(f).startswith(("api", "placement", "density_hip.h"))):   # the kernels, not the host-side C ABI
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:12]


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, debug=False):
    """The shipped library; debug=True: libdensity_hip_debug.so, the same sources with -DDENSITY_HIP_DEBUG (profiling / tuning tools load it through
    density_amd._lib.use_debug_build(); nothing in tests/, bench.py or __graft_entry__ does)."""
    LIB = LIB_DEBUG if debug else globals()["LIB"]
    if not force and not needs_build(LIB):
        return LIB
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-value",
           f'-DDENSITY_HIP_KERNELS_ID="{kernels_id()}"'] + (["-DDENSITY_HIP_DEBUG"] if debug else []) + ["-o", LIB] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    # the encoder's hand-issued loads rely on registers the compiler must not have used: checked in the compiled code, every build
    check = os.path.join(os.path.dirname(HERE), "tools", "check_isa.py")
    if os.path.exists(check):
        r = subprocess.run([sys.executable, check, LIB], capture_output=True, text=True)
        if r.returncode != 0:
            os.remove(LIB)
            raise RuntimeError("tools/check_isa.py rejected the compiled rotation encoder:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True, debug="--debug" in sys.argv))
