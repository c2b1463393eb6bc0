#!/bin/bash
# Collects the per-round profile set on the GPU box into gpurun_out/ (copy the results into profiles/):
#   kernel stats of the bench command, one kernel trace per workload (step traces), PMC read / write passes.
# usage (from the repository root on the box): bash tools/profile_round.sh <round tag, e.g. r02>
set -e
tag=${1:-r02}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$tag -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/stats_$tag.json 2> $out/stats_$tag.err
cp $(find $out/stats_$tag -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
for wl in streamspot_batch theia; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/tr_${tag}_$wl -- python3 $root/bench.py --steps 3 --warmup 2 --no-knn --no-cpu-baseline --no-entity --workload $wl > $out/tr_${tag}_$wl.json 2> $out/tr_${tag}_$wl.err
  python3 $root/tools/step_trace.py $(find $out/tr_${tag}_$wl -name "*kernel_trace.csv" | head -1) > $out/${tag}_step_trace_$wl.txt
done
cd $root
bash tools/pmc_pass.sh rd_ss FETCH_SIZE && bash tools/pmc_pass.sh wr_ss WRITE_SIZE
bash tools/pmc_pass.sh rd_th FETCH_SIZE --workload theia && bash tools/pmc_pass.sh wr_th WRITE_SIZE --workload theia
MAGIC_PROFILE_ROUND=$tag python3 tools/make_traffic.py gpurun_out/pmc_rd_ss gpurun_out/pmc_wr_ss streamspot_batch 96000 4 > /dev/null
MAGIC_PROFILE_ROUND=$tag python3 tools/make_traffic.py gpurun_out/pmc_rd_th gpurun_out/pmc_wr_th theia 700000 3 > /dev/null
cp profiles/traffic.json profiles/${tag}_pmc_summary_*.txt gpurun_out/
echo profile set $tag done
