#!/bin/bash
# A/B of generator variants on the GPU box (dev tool): for each "NAME ENV=VAL ..." line regenerate, rebuild, run the short bench, print the phase times
# usage: scripts/dbg/ab_gen.sh "base" "f2first MBLS_GEN_F2_FIRST=1" ...
for spec in "$@"; do
    name=${spec%% *}; envs=""; [ "$spec" != "$name" ] && envs=${spec#* }
    env $envs python3 tools/gen_tower_d.py > /dev/null && env $envs python3 -m milagro_bls_amd.build > /dev/null 2>&1
    python3 bench.py --no-variants --no-cpu-baseline --steps 10 2> /dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$name', round(d['value']), round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phase_ms'].items()}, d['bitmap_matches_expectation'])"
done
