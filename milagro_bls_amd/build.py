"""Build the gfx950 shared library in-tree: milagro_bls_amd/libmbls_hip.so (hipcc, no torch dependency)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmbls_hip.so")
SOURCES = ["mbls_kernels.hip"]
DEPS = ["mbls_fp.h", "mbls_fp_asm.inc", "mbls_fpd_asm.inc", "mbls_towerd_asm.inc", "mbls_tower.h", "mbls_curve.h", "mbls_hash.h", "mbls_pairing.h", "mbls_lanes.h", "mbls_ops.h", "mbls_coop.h", "mbls_coop_prog.inc",
        "mbls_constants.inc", os.path.join("..", "..", "include", "mbls.h")]


STAMP = LIB + ".srchash"


def source_hash():
    """content hash of everything the library is built from (file times do not survive the copy to the GPU box)"""
    import hashlib
    h = hashlib.sha256(os.environ.get("MBLS_EXTRA_HIPCC_FLAGS", "").encode())
    for f in SOURCES + DEPS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    if os.environ.get("MBLS_TRUST_PREBUILT") == "1":
        return False
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = os.environ.get("MBLS_EXTRA_HIPCC_FLAGS", "").split()
    objs, procs = [], []
    for src in SOURCES:
        obj = os.path.join(HERE, "_" + src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, src), "-o", obj] + extra
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    for p, cmd in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    with open(STAMP, "w") as fh:
        fh.write(source_hash() + "\n")
    for o in objs:
        os.remove(o)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print(LIB)
