#!/usr/bin/env python3
"""Search for the product shapes of the cooperative programs (gen_coop.SHAPES): for every block of every program, Karatsuba or schoolbook
at the Fp2 and at the Fp6 level -- whichever gives the shortest step list, one block at a time (coordinate descent, two sweeps), subject to
the program staying within SLOT_BUDGET LDS slots (eight waves per CU). Prints the table to paste into gen_coop.py. Dev tool."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_coop as G  # noqa: E402

SLOT_BUDGET = 315
DIMS = [(0, 1, 2), (0, 1), (0, 1), (0, 1), (0, 1)]      # Fp2 products, Fp6 products, Fp12 squaring, Fp12 products, line products
NDIM = len(DIMS)


def measure(name):
    comp = G.compile_program(G.PROGRAMS[name]())
    # a product step issues about 700 instructions, the others about 350 (mbls_coop.h): cost in units of 350
    cost = sum(2 if (info & 0xFF) == G.K_MUL else 1 for info, _ in comp["steps"])
    return cost, comp["n_slots"], comp["total_steps"]


def main():
    table = dict(G.SHAPES)                 # start from the installed table
    for name in (sys.argv[1:] or list(G.PROGRAMS)):
        G.SHAPES.clear(); G.SHAPES.update(table)
        m = G.PROGRAMS[name]()
        blocks = [b for b in m.blocks]
        runs = {}
        for b, rep in m.order:
            runs[b] = runs.get(b, 0) + rep
        best = measure(name)
        print(name, "start", best, flush=True)
        for sweep in range(3):
            changed = False
            for b in sorted(blocks, key=lambda x: -runs.get(x, 0)):
                if not runs.get(b):
                    continue
                cur = (tuple(G.SHAPES.get((name, b), ())) + (0,) * NDIM)[:NDIM]
                for dim, alts in enumerate(DIMS):
                    for v in alts:
                        if v == cur[dim]:
                            continue
                        shape = cur[:dim] + (v,) + cur[dim + 1:]
                        G.SHAPES[(name, b)] = shape
                        got = measure(name)
                        ok = got[1] <= max(SLOT_BUDGET, best[1])
                        if ok and (got[0] < best[0] or (got[0] == best[0] and got[1] < best[1])):
                            best, cur, changed = got, shape, True
                            print("  ", b, shape, got, flush=True)
                while cur and cur[-1] == 0:
                    cur = cur[:-1]
                if not cur:
                    G.SHAPES.pop((name, b), None)
                else:
                    G.SHAPES[(name, b)] = cur
            if not changed:
                break
        table = dict(G.SHAPES)
        print(name, "->", best, flush=True)
    print("SHAPES = {")
    for k in sorted(table):
        print("    %r: %r," % (k, table[k]))
    print("}")


if __name__ == "__main__":
    main()
