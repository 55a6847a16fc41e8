#!/bin/bash
# The engine clock under the mix benchmark and under the bench's kernels (dev probe, run on the GPU box through gpurun): GRBM_GUI_ACTIVE / (8 XCDs x duration).
ROOT=$(pwd); OUT=$ROOT/gpurun_out/mix_clock; mkdir -p $OUT; export TMPDIR=/tmp
python3 scripts/dbg/gen_mixbench.py > /dev/null && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mixbench scripts/dbg/mixbench.hip || exit 1
cd /tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/m -o m -- /tmp/mixbench > $OUT/mixbench.txt 2> $OUT/m.err
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/mix_clock/m/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if dur > 2_000_000:
            acc[(r["Kernel_Name"].split("(")[0][:40], r["Grid_Size"])].append(float(r["Counter_Value"]) / 8 / dur)
for k, v in acc.items():
    print(k, "launches", len(v), "GHz min %.3f max %.3f" % (min(v), max(v)))
PY
