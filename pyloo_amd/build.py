"""Build libpyloo_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m pyloo_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting ``pyloo_amd/lib/libpyloo_amd.so`` is
git-ignored but travels with the tree to the GPU box.
"""

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpyloo_amd.so")
SOURCES = ["pla_kernels.hip", "pla_capi.hip"]
PUBLIC_HEADER = os.path.join(HERE, "..", "include", "pyloo_amd.h")
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    # every file under csrc/ (the kernels live in headers) + the public header
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] + [PUBLIC_HEADER]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    """Compile the shared library if it is missing or older than its sources."""
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [
        _hipcc(), "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC", "-shared",
        # the wave kernel keeps a whole row in registers; hoisting loop-invariant constants out of
        # its row loop (MachineLICM) pushes it over the register budget and it spills
        "-mllvm", "-disable-machine-licm",
        "-o", LIB_PATH + ".tmp",
    ] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + proc.stdout + proc.stderr)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
