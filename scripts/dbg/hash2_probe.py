import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "oracle"); sys.path.insert(0, "tests")
from milagro_bls_amd import batch as mb, _native as N
N.default_context()
msgs = bytes(32)
print("mode1", mb.hash_to_g2_batch(msgs, 1, mode=1).hex())
print("mode3", mb.hash_to_g2_batch(msgs, 1, mode=3).hex())
