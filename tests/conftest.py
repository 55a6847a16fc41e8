import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "pymodel"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # the untracked generated source (milagro_bls_amd/csrc/mbls_towerd_asm.inc, 25 MB: written by the build, milagro_bls_amd/build.py) exists before any test reads it;
    # a tree that holds the file and its stamp is left alone (the GPU box)
    try:
        from milagro_bls_amd import build
        build.ensure_generated()
    except Exception as e:                                  # noqa: BLE001 -- the tests that need the file say so themselves
        sys.stderr.write("conftest: ensure_generated failed: %r\n" % (e,))


@pytest.fixture(scope="session")
def vectors():
    import helpers
    return helpers.load_vectors()


@pytest.fixture(scope="session")
def emul():
    import helpers
    return helpers.load_emulator()


COOP_ENV = ("MBLS_COOP_MAX_ITEMS", "MBLS_COOP_HASH_MAX_ITEMS", "MBLS_SPLIT_MAX_ITEMS")


@pytest.fixture(params=["waves", "lanes", "lanes2pair"])
def engine(request):
    """Every GPU test that compares with the oracle or the golden file runs once per engine:
    'waves' = the library's defaults (small batches take the cooperative one-wave-per-item programs, mbls_coop.h);
    'lanes' = every batch forced onto the one-lane-per-item kernels (k_hash, k_miller*, k_sig_verdict -- the signature's subgroup test read
    off the Miller loop --, k_final) in the form a batch of that size takes by default: lane pairs below a quarter of a round (k_hash2,
    k_miller_split4, k_final2), the two pairs of an item on two lanes below half a round (k_miller_split); 'lanes2pair' = the same with the two-pair loop k_miller, the kernel the headline number is measured on:
    mbls_ctx_set_coop_max_items(0) + mbls_ctx_set_coop_hash_max_items(0) (+ mbls_ctx_set_lane_shaping(0, ...)) on the default context, and
    the same through the environment for contexts the test creates itself (mbls_ctx_create reads it; mbls_multi_create makes its contexts
    that way)."""
    from milagro_bls_amd import _native
    ctx = _native.default_context()
    old = {k: os.environ.get(k) for k in COOP_ENV}
    if request.param != "waves":
        for k in COOP_ENV[:2]:
            os.environ[k] = "0"
        ctx.set_coop_max_items(0); ctx.set_coop_hash_max_items(0)
    if request.param == "lanes2pair":
        os.environ[COOP_ENV[2]] = "0"
        ctx.set_lane_shaping(0, (1 << 64) - 1)
    try:
        yield request.param
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        ctx.reset_tuning()
