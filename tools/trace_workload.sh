#!/bin/bash
# Kernel trace of one bench workload (eager step) -> gpurun_out/trace_<workload>.txt
# usage (on the GPU box, from the repository root): bash tools/trace_workload.sh <workload> [more workloads]
set -e
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
for wl in "$@"; do
  rm -rf $out/trw_$wl
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $out/trw_$wl -- python3 $root/bench.py --steps 3 --warmup 2 --no-knn --no-cpu-baseline --no-entity --no-configs --workload $wl > $out/trw_$wl.json 2> $out/trw_$wl.err
  python3 $root/tools/step_trace.py $(find $out/trw_$wl -name "*kernel_trace.csv" | head -1) > $out/trace_$wl.txt
  rm -rf $out/trw_$wl
  echo trace $wl done
done
