cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06/t1
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -8
for spec in "vm 16384"; do
  set -- $spec
  rocprofv3 --kernel-trace -d gpurun_out/r06/t1/tr_$1_$2 -o t --output-format csv -- python3 scripts/dbg/trace_n.py $1 $2 > gpurun_out/r06/t1/tr_$1_$2.log 2>&1
  python3 scripts/dbg/trace_show.py gpurun_out/r06/t1/tr_$1_$2 k_aggregate_raw_d > gpurun_out/r06/t1/timeline_$1_$2.txt 2>&1
  rm -rf gpurun_out/r06/t1/tr_$1_$2
done
head -16 gpurun_out/r06/t1/timeline_vm_16384.txt; tail -3 gpurun_out/r06/t1/timeline_vm_16384.txt
