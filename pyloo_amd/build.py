"""Build libpyloo_amd.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

    python -m pyloo_amd.build [--force] [-DPLA_EXPERIMENT ...]

hipcc cross-compiles without a GPU; the resulting ``pyloo_amd/lib/libpyloo_amd.so`` is
git-ignored but travels with the tree to the GPU box.

The kernels are split over several translation units (``csrc/pla_k_*.hip``, see ``csrc/pla_launch.h``) that are
compiled in parallel -- one hipcc process per unit, as many at a time as there are cores -- and an object is rebuilt
only when one of the files it includes (hipcc's own dependency file) or the flags changed.
"""

import concurrent.futures
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libpyloo_amd.so")
SOURCES = [
    "pla_k_general.hip", "pla_k_wave_f64.hip", "pla_k_wave_f32.hip", "pla_k_chunked_f64.hip", "pla_k_chunked_f32.hip",
    "pla_k_fit.hip", "pla_k_lwout.hip", "pla_k_waic.hip", "pla_k_col.hip", "pla_k_eloo.hip", "pla_capi.hip",
]
PUBLIC_HEADER = os.path.join(HERE, "..", "include", "pyloo_amd.h")
ARCH = "gfx950"
FLAGS = [
    "-O3", "-std=c++17", f"--offload-arch={ARCH}", "-fPIC",
    # the wave kernel keeps a whole row in registers; hoisting loop-invariant constants out of
    # its row loop (MachineLICM) pushes it over the register budget and it spills
    "-mllvm", "-disable-machine-licm",
]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _obj_dir(tag):
    return os.path.join(LIB_DIR, "obj" + ("_" + tag if tag else ""))


def _deps_of(depfile):
    """Prerequisites listed in a make-style dependency file written by hipcc -MD."""
    try:
        text = open(depfile).read()
    except OSError:
        return None
    words = text.replace("\\\n", " ").split()
    return [w for w in words[1:] if not w.endswith(":")]


def _stale(src, obj, stamp):
    if not os.path.exists(obj):
        return True
    try:
        if open(obj + ".flags").read() != stamp:
            return True
    except OSError:
        return True
    deps = _deps_of(obj + ".d")
    if deps is None:
        return True
    t = os.path.getmtime(obj)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps + [src])


def needs_build(extra=(), lib_path=LIB_PATH, tag=""):
    if not os.path.exists(lib_path):
        return True
    stamp = _stamp(extra)
    od = _obj_dir(tag)
    if not os.path.isdir(od):
        # a tree that travelled without its objects (the GPU box): the library against every source file, as one unit
        t = os.path.getmtime(lib_path)
        deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))] + [PUBLIC_HEADER]
        return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))
    for s in SOURCES:
        obj = os.path.join(od, s[:-4] + ".o")
        if _stale(os.path.join(CSRC, s), obj, stamp) or os.path.getmtime(obj) > os.path.getmtime(lib_path):
            return True
    return False


def _stamp(extra):
    return hashlib.sha1(" ".join(FLAGS + list(extra)).encode()).hexdigest()


def build(force=False, verbose=False, extra=(), lib_path=LIB_PATH, tag="", jobs=None):
    """Compile the shared library if it is missing or older than its sources.

    extra: additional hipcc flags (e.g. ``-DPLA_EXPERIMENT``); tag: object directory suffix for such a variant;
    lib_path: where the variant is linked.
    """
    extra = list(extra)
    if not force and not needs_build(extra, lib_path, tag):
        return lib_path
    os.makedirs(LIB_DIR, exist_ok=True)
    od = _obj_dir(tag)
    os.makedirs(od, exist_ok=True)
    hipcc, stamp = _hipcc(), _stamp(extra)
    todo = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(od, s[:-4] + ".o")
        if force or _stale(src, obj, stamp):
            todo.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + extra + ["-c", "-MD", "-MF", obj + ".d", "-o", obj + ".tmp", src]
        proc = subprocess.run(cmd, capture_output=True, text=True)
        if proc.returncode != 0:
            return src, " ".join(cmd) + "\n" + proc.stdout + proc.stderr
        os.replace(obj + ".tmp", obj)
        with open(obj + ".flags", "w") as f:
            f.write(stamp)
        return src, None

    jobs = jobs or max(1, min(len(todo), os.cpu_count() or 1))
    if verbose and todo:
        print(f"{hipcc} {' '.join(FLAGS + extra)} -c  [{len(todo)} units, {jobs} at a time]: " + " ".join(os.path.basename(s) for s, _ in todo), flush=True)
    errors = []
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
        for src, err in pool.map(compile_one, todo):
            if err:
                errors.append(err)
    if errors:
        raise RuntimeError("hipcc failed:\n" + "\n".join(errors))
    objs = [os.path.join(od, s[:-4] + ".o") for s in SOURCES]
    cmd = [hipcc, f"--offload-arch={ARCH}", "-fPIC", "-shared", "-o", lib_path + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("link failed:\n" + proc.stdout + proc.stderr)
    os.replace(lib_path + ".tmp", lib_path)
    return lib_path


if __name__ == "__main__":
    args = sys.argv[1:]
    extra = [a for a in args if a.startswith("-D") or a.startswith("-mllvm") or a.startswith("-f")]
    out, tag = LIB_PATH, ""
    for a in args:
        if a.startswith("--alt="):  # a variant for A/B runs: pyloo_amd/lib/alt_<name>.so (use with PYLOO_AMD_LIB=...)
            tag = a[len("--alt="):]
            out = os.path.join(LIB_DIR, f"alt_{tag}.so")
    print(build(force="--force" in args, verbose=True, extra=extra, lib_path=out, tag=tag))
