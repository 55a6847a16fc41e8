"""dev probe: why a per-class clock model of the VALU stream over-promises what a second wave per SIMD buys. The same twelve instructions --
8 v_mad_u64_u32 (eight accumulators: a product scan's dependency distance) + 4 plain two-operand operations, the kernels' 2 : 1 proportion -- interleaved (mode 4) and
in two homogeneous blocks (mode 5), the multiply-accumulates alone (mode 0) and the plain operations alone (mode 6), at 1, 2 and 4 waves per SIMD. The figure that
matters is the MARGINAL cost of a plain operation inside the multiply-accumulate stream: (time of 8 mad + 4 plain - time of 8 mad) / 4."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from milagro_bls_amd import _native as N
ctx = N.default_context(); lib = N.lib()
per_iter = {0: 128, 4: 192, 5: 192, 6: 128}
names = {0: "8 multiply-accumulates alone", 6: "plain operations alone", 4: "8 mad + 4 plain, interleaved 2 : 1", 5: "8 mad + 4 plain, two homogeneous blocks"}
IT = 40000
for _ in range(3):                                                    # let the clocks settle under this kind of load
    ms = C.c_float(); ctx.check(lib.mbls_valu_bench(ctx.handle, 4, 4, IT, C.byref(ms)))
ns = {}
for mode in (0, 6, 4, 5):
    for w in (1, 2, 4):
        best = 1e9
        for _ in range(3):
            ms = C.c_float()
            ctx.check(lib.mbls_valu_bench(ctx.handle, mode, w, IT, C.byref(ms)))
            best = min(best, ms.value)
        ns[mode, w] = best * 1e6 / (w * IT * per_iter[mode])           # SIMD time per wave-instruction
        print("%-42s %d wave(s) per SIMD: %6.3f ns per wave-instruction" % (names[mode], w, ns[mode, w]), flush=True)
for w in (1, 2, 4):
    group_mad, group_mix = 8 * ns[0, w], 12 * ns[4, w]
    print("%d wave(s) per SIMD: a group of 8 mad %.2f ns, with 4 plain operations between them %.2f ns -> a plain operation costs %.2f ns inside the stream "
          "(%.2f ns in a stream of its own)" % (w, group_mad, group_mix, (group_mix - group_mad) / 4, ns[6, w]))
