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

# The one generated file that is NOT tracked: 25 MB of straight-line routines (tools/gen_tower_d.py, ~3 s). It is written into csrc/ (git-ignored, it still travels
# to the GPU box with the snapshot like the built library) whenever it is missing or older than what the generator's sources say; tests/test_asm_sim_d_cpu.py still
# compares it with a fresh run of the generator.
ROOT = os.path.dirname(HERE)
GENERATED = {"mbls_towerd_asm.inc": ("gen_tower_d.py", ["gen_tower_d.py", "gen_fp_asm.py", "gen_fpd_asm.py"])}


def _gen_inputs_hash(inputs):
    import hashlib
    h = hashlib.sha256()
    for f in inputs:
        with open(os.path.join(ROOT, "tools", f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def ensure_generated(verbose=False):
    """(re)generate the untracked generated sources when they are missing or stale against their generators"""
    for name, (script, inputs) in GENERATED.items():
        out, stamp = os.path.join(CSRC, name), os.path.join(CSRC, name + ".genhash")
        want = _gen_inputs_hash(inputs)
        if os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == want:
            continue
        if os.path.exists(out) and not os.path.isdir(os.path.join(ROOT, "tools")):
            continue                                    # a tree without the generators (never the repository): take the file as it is
        cmd = [sys.executable, os.path.join(ROOT, "tools", script)]
        if verbose:
            print(" ".join(cmd), flush=True)
        env = dict(os.environ); env.pop("MBLS_GEN_OUT_DIR", None)
        subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL, env=env)
        with open(stamp, "w") as fh:
            fh.write(want + "\n")


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
    ensure_generated()
    if os.environ.get("MBLS_TRUST_PREBUILT") == "1":
        return False
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    ensure_generated(verbose)
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
