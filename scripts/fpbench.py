import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from milagro_bls_amd import batch
for lanes in (65536, 131072, 262144, 524288):
    ms = batch.fp_mul_bench(lanes, 4000)
    print("lanes=%d: %.2f ms -> %.3f us per wave-mul at %d waves/SIMD, %.3e Fp mul/s" % (lanes, ms, ms*1e3/4000/max(1,lanes//65536), lanes//65536, lanes*4000/(ms*1e-3)))
