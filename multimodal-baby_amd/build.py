"""Build libcvcl_hip.so for gfx950 in-tree (multimodal-baby_amd/lib/).

    python multimodal-baby_amd/build.py [--force]

One hipcc invocation per translation unit (parallel), then one link.  The built library travels to
the GPU box with the repo snapshot; nothing is JIT-compiled at run time.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libcvcl_hip.so")
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"] + os.environ.get("CVCL_EXTRA_FLAGS", "").split()
# experiments: CVCL_EXTRA_FLAGS=-DCVCL_PLAIN_STORES CVCL_LIB_SUFFIX=_plain python build.py -> lib/libcvcl_hip_plain.so ($CVCL_HIP_LIB)
SUFFIX = os.environ.get("CVCL_LIB_SUFFIX", "")
if SUFFIX:
    LIB = os.path.join(LIBDIR, f"libcvcl_hip{SUFFIX}.so")
    OBJDIR = os.path.join(HERE, "build" + SUFFIX)


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _digest():
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in sorted(os.listdir(root)):
            with open(os.path.join(root, f), "rb") as fh:
                h.update(f.encode())
                h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, f"libcvcl_hip{SUFFIX}.stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == dig:
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}; libcvcl_hip.so must be prebuilt")

    def compile_one(src):
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
        cmd = [HIPCC, "-x", "hip", *FLAGS, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(6, len(sources()))) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB, *objs]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
