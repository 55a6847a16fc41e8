// mbls_kernels_w2.hip -- the pipeline kernels compiled for 256 registers per lane (2 waves per SIMD), see mbls_kernels.hip.
#define MBLS_WAVES_PER_SIMD 2
#define MBLS_KSUF(x) x##_w2
#define MBLS_KERNELS_ONLY 1
#include "mbls_kernels.hip"
