"""In-tree build of libmi355interp.so (hand-written HIP, gfx950 only)."""
import glob
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmi355interp.so")

# -ffp-contract=off: the fp64/fp32 blends must round every product and sum
# separately (bit parity with oracle/); correctly rounded fp32 div for Restrict.
HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall", "-Wno-unused-value",
]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _deps():
    return sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(REPO_ROOT, "include", "*.h"))


def source_hash():
    """sha256 over the device / ABI sources the library is built from (sorted paths, contents only): what a committed
    profile is stamped with, so that bench.py can tell whether a measured traffic figure belongs to the running build."""
    import hashlib
    h = hashlib.sha256()
    for p in sorted(_deps()):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(p) > t for p in _deps())


def build_lib(force=False, verbose=False):
    """Compile csrc/*.hip -> libmi355interp.so with hipcc (cross-compiles without a GPU)."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libmi355interp.so cannot be built (there is no CPU fallback)")
    extra = os.environ.get("MI_EXTRA_HIPCC_FLAGS", "").split()
    cmd = [hipcc] + HIPCC_FLAGS + extra + ["-I", os.path.join(REPO_ROOT, "include"), "-I", CSRC,
                                   "-o", LIB_PATH + ".tmp"] + sources()
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB_PATH + ".tmp", LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
