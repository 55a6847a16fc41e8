cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06/base
for spec in "fav 1" "vm 16384" "vm 65536"; do
  set -- $spec
  rocprofv3 --kernel-trace -d gpurun_out/r06/base/tr_$1_$2 -o t --output-format csv -- python3 scripts/dbg/trace_n.py $1 $2 > gpurun_out/r06/base/tr_$1_$2.log 2>&1
  first=k_aggregate_raw_d; 
  python3 scripts/dbg/trace_show.py gpurun_out/r06/base/tr_$1_$2 $first > gpurun_out/r06/base/timeline_$1_$2.txt 2>&1
  rm -rf gpurun_out/r06/base/tr_$1_$2
done
python3 scripts/latency.py > gpurun_out/r06/base/latency.log 2>&1
tail -5 gpurun_out/r06/base/latency.log
