#!/bin/bash
# Collect the round's profiles on the GPU box (run through gpurun from the repository root):
#   bash scripts/profile_round.sh r02
# 1. rocprofv3 kernel trace + stats of a bench run, 2./3. HBM traffic counters in separate passes (FETCH_SIZE, WRITE_SIZE cannot share
# a pass), 4. SQ counters, 5. the bench line itself (with the CPU legs), 6. code-object metadata (registers, scratch).
TAG=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cp milagro_bls_amd/libmbls_hip.so.srchash $OUT/source_hash.txt      # the build every counter below belongs to
B="python3 $ROOT/bench.py --no-cpu-baseline --no-variants"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- $B --steps 5 --warmup 1 > $OUT/stats_bench.json 2> $OUT/stats.err
for CFG in 2 4 lat mid; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_c$CFG -o c$CFG -- python3 $ROOT/scripts/run_config.py $CFG > /dev/null 2> $OUT/stats_c$CFG.err
done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- $B --steps 2 --warmup 1 > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- $B --steps 2 --warmup 1 > /dev/null 2> $OUT/write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/sq -o s -- $B --steps 2 --warmup 1 > /dev/null 2> $OUT/sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/grbm -o g -- $B --steps 2 --warmup 1 > /dev/null 2> $OUT/grbm.err
cd $ROOT
# the traffic figures of THIS build into profiles/hbm_traffic.json (stamped with the library's source hash) before the bench line is taken: bench.py attaches them to
# roofline.traffic only when the hash matches the library it loaded. (The same collection is repeated in the container on the merged files, for the commit.)
echo '{}' > $OUT/bench_line.json
python3 scripts/collect_profiles.py $TAG $OUT/stats $OUT/fetch $OUT/write $OUT/sq $OUT/bench_line.json > $OUT/collect_on_box.log 2>&1
python3 bench.py --steps 10 --warmup 2 > $OUT/bench_line.json 2> $OUT/bench.err
python3 bench.py --items 131072 --steps 5 --warmup 1 --no-cpu-baseline > $OUT/bench_line_131072.json 2>> $OUT/bench.err
python3 scripts/latency.py $OUT/latency.json > $OUT/latency.log 2>&1
python3 scripts/time_keyops.py > $OUT/keyops.log 2>&1; cp gpurun_out/keyops.json $OUT/keyops.json
bash scripts/kstats.sh > $OUT/kstats.txt 2>&1
python3 scripts/throughput_vs_n.py $TAG > $OUT/sweep.log 2>&1; cp gpurun_out/${TAG}_throughput_vs_n.json $OUT/throughput_vs_n.json
# the instruction-cache probe is built from its source on the box (no binary in the repository)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/icbench scripts/dbg/icbench.hip > $OUT/icbench_build.log 2>&1 && /tmp/icbench > $OUT/icache_footprint.txt 2>&1
find $OUT -name "*.csv" -size +20M -delete
ls -la $OUT $OUT/stats $OUT/fetch 2>/dev/null | head -40
