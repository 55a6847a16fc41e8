"""The C++ host-side mirror of the reference's API (include/milagro_bls.hpp): compiles and links against libmbls_hip.so
on CPU; runs the restated reference tests on the GPU."""
import os
import subprocess

import pytest

import helpers

SRC = os.path.join(helpers.ROOT, "tests", "cpp", "test_api.cpp")
EXE = os.path.join(helpers.ROOT, "tests", "cpp", "test_api")


def build_exe():
    from milagro_bls_amd import build
    lib = build.build()
    libdir = os.path.dirname(lib)
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(helpers.ROOT, "include"), SRC, "-o", EXE, "-L", libdir, "-lmbls_hip",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-pthread"]
    subprocess.check_call(cmd)
    return EXE


def test_cpp_mirror_compiles_and_links():
    exe = build_exe()
    assert os.path.exists(exe)


def test_cpp_key_generate_matches_hkdf_restatement():
    """SecretKey::key_generate of include/milagro_bls.hpp (host-only C++: SHA-256, HMAC, HKDF, mod r) against the Python mirror,
    which uses hashlib/hmac (reference src/keys.rs:45-77). No GPU involved."""
    import random
    from milagro_bls_amd.api import SecretKey, AmclError
    libdir = os.path.join(helpers.ROOT, "milagro_bls_amd")
    exe = os.path.join(helpers.ROOT, "tests", "cpp", "test_keygen")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(helpers.ROOT, "include"), os.path.join(helpers.ROOT, "tests", "cpp", "test_keygen.cpp"),
                           "-o", exe, "-L", libdir, "-lmbls_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    rnd = random.Random(5)
    cases = [(bytes(range(32)), b""), (bytes(32), b""), (rnd.randbytes(32), b"info"), (rnd.randbytes(48), rnd.randbytes(7)), (rnd.randbytes(100), b"")]
    args = []
    for ikm, info in cases:
        args += [ikm.hex(), info.hex() or "-"]
    out = subprocess.run([exe] + args + ["00" * 31, "-"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.split("\n")
    for (ikm, info), line in zip(cases, lines):
        assert line == SecretKey.key_generate(ikm, info).as_bytes().hex()
    assert lines[len(cases)] == "AmclError 5"                       # ikm shorter than 32 bytes: InvalidSecretKeySize
    with pytest.raises(AmclError):
        SecretKey.key_generate(bytes(31))


def test_key_generate_external_vectors():
    """External pins for KeyGenerate (reference src/keys.rs:45-77 holds none): RFC 5869 A.1-A.3 for the HKDF primitives and the four
    EIP-2333 master-key cases (HKDF_mod_r with key_info = "" is exactly SecretKey::key_generate(seed, b"")), on BOTH host mirrors --
    include/milagro_bls.hpp (its own SHA-256 / HMAC / HKDF / mod r) and milagro_bls_amd/api.py. No GPU involved."""
    import json
    from milagro_bls_amd import api
    with open(os.path.join(helpers.ROOT, "tests", "golden", "keygen_vectors.json")) as f:
        vec = json.load(f)
    libdir = os.path.join(helpers.ROOT, "milagro_bls_amd")
    exe = os.path.join(helpers.ROOT, "tests", "cpp", "test_keygen")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(helpers.ROOT, "include"), os.path.join(helpers.ROOT, "tests", "cpp", "test_keygen.cpp"),
                           "-o", exe, "-L", libdir, "-lmbls_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    for v in vec["rfc5869"]:
        ikm, salt, info = bytes.fromhex(v["ikm"]), bytes.fromhex(v["salt"]), bytes.fromhex(v["info"])
        prk = api.hkdf_extract(salt, ikm)
        assert prk.hex() == v["prk"] and api.hkdf_expand(prk, info, v["L"]).hex() == v["okm"], v["name"]
        out = subprocess.run([exe, "hkdf", v["ikm"] or "-", v["salt"] or "-", v["info"] or "-", str(v["L"])], capture_output=True, text=True, timeout=60)
        assert out.returncode == 0 and out.stdout.split() == [v["prk"], v["okm"]], (v["name"], out.stdout, out.stderr)
    args = []
    for v in vec["eip2333_master_sk"]:
        assert api.SecretKey.key_generate(bytes.fromhex(v["seed"]), b"").as_raw() == int(v["sk"])
        args += [v["seed"], "-"]
    out = subprocess.run([exe] + args, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert [int(x, 16) for x in out.stdout.split()] == [int(v["sk"]) for v in vec["eip2333_master_sk"]]


@pytest.mark.gpu
def test_cpp_mirror_reference_tests_on_gpu():
    exe = build_exe()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all C++ API checks passed" in out.stdout
