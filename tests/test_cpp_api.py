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
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return EXE


def test_cpp_mirror_compiles_and_links():
    exe = build_exe()
    assert os.path.exists(exe)


@pytest.mark.gpu
def test_cpp_mirror_reference_tests_on_gpu():
    exe = build_exe()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "all C++ API checks passed" in out.stdout
