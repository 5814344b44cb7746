"""Build libmcedm_hip.so for gfx950 with hipcc (no GPU needed: hipcc cross-compiles).

    python m-cedm_amd/build.py [--force]

Objects go to m-cedm_amd/csrc/_build/, the library to m-cedm_amd/libmcedm_hip.so (git-ignored,
but shipped to the GPU box by gpurun).  A source is recompiled only when it or a header changed.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
LIB = os.path.join(HERE, "libmcedm_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
COMMON = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}",
          "-Wall", "-Wno-unused-function"] + os.environ.get("MCEDM_EXTRA_HIPCC_FLAGS", "").split()
# the fp64 sampler arithmetic must follow the reference's evaluation order: no fma contraction there
PER_FILE = {"edm.hip": ["-ffp-contract=off"], "pde.hip": ["-ffp-contract=off"]}


def _digest(paths, extra):
    h = hashlib.sha256(repr(extra).encode())
    for p in sorted(paths):
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hdrs.append(os.path.join(ROOT, "include", "mcedm_hip.h"))
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        flags = COMMON + PER_FILE.get(s, [])
        stamp = obj + ".sha"
        dig = _digest([src] + hdrs, flags)
        fresh = (not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig)
        jobs.append((src, obj, flags, stamp, dig, fresh))

    def compile_one(job):
        src, obj, flags, stamp, dig, fresh = job
        if fresh:
            return obj
        cmd = [HIPCC] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(stamp, "w") as f:
            f.write(dig)
        return obj

    try:
        with ThreadPoolExecutor(max_workers=4) as ex:
            objs = list(ex.map(compile_one, jobs))
    except subprocess.CalledProcessError:
        if os.path.exists(LIB):
            os.remove(LIB)          # never leave a library behind that is older than the sources it claims to be built from
        raise
    if force or not os.path.exists(LIB) or any(not j[5] for j in jobs):
        cmd = [HIPCC, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
