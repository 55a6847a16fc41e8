"""milagro_bls_amd -- MI355X (gfx950) batch BLS12-381 signature verification behind milagro_bls's API.

The package holds only what the verification path needs: csrc/ (HIP kernels + the C ABI of libmbls_hip.so,
declared in include/mbls.h), `api` (host-side mirror of the reference's types, reference src/lib.rs:17-22)
and `batch` (batch / device-pointer entry points). There is no CPU fallback."""
from ._native import Context, MblsError, default_context, PK_COMPRESSED, PK_UNCOMPRESSED  # noqa: F401
from .api import (AggregatePublicKey, AggregateSignature, AmclError, Keypair, PublicKey, SecretKey, Signature,  # noqa: F401
                  G1_BYTES, G2_BYTES, SECRET_KEY_BYTES)
from . import batch  # noqa: F401
