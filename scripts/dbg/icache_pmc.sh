#!/bin/bash
# Instruction-cache counters of the bench's kernels (dev probe, run on the GPU box through gpurun): two --pmc passes, nothing else in them.
ROOT=$(pwd); OUT=$ROOT/gpurun_out/icache_pmc; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
B="python3 $ROOT/bench.py --no-cpu-baseline --no-variants --steps 2 --warmup 1"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/a -o a -- $B > /dev/null 2> $OUT/a.err
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/b -o b -- $B > /dev/null 2> $OUT/b.err
cd $ROOT
python3 - <<'PY'
import csv, glob, collections
for tag in ("a", "b"):
    f = glob.glob("gpurun_out/icache_pmc/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not f:
        print(tag, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("(")[0]
        if not k.startswith("k_"):
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (k, r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); cnt[k] += 1
    for k in acc:
        print(tag, k, "launches", cnt[k], {c: round(v / cnt[k]) for c, v in acc[k].items()})
PY
