#!/bin/bash
# EXPERIMENT: plain operations of the generated routines in 4-byte VOP1 / VOP2 encodings (MBLS_GEN_E32=1) against the 8-byte VOP3 forms. Run on the GPU box; the three
# generators are re-run for each variant and the tree ends as it began.
cd "$(dirname "$0")/../.."
for spec in "base" "e32 MBLS_GEN_E32=1" "base_again"; do
    name=${spec%% *}; envs=""; [ "$spec" != "$name" ] && envs=${spec#* }
    for g in gen_fp_asm gen_fpd_asm gen_tower_d; do env $envs python3 tools/$g.py > /dev/null; done
    env $envs python3 -m milagro_bls_amd.build --force > /dev/null 2>&1
    python3 bench.py --no-variants --no-cpu-baseline --steps 10 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phase_ms'].items()}, d['bitmap_matches_expectation'])"
done
