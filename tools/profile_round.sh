#!/bin/bash
# Collects the per-round profile set on the GPU box into gpurun_out/ (copy the results into profiles/):
#   kernel stats of the bench command, one kernel trace per workload (step traces), PMC read / write passes
#   (sparse sweeps, projection launch shapes, KNN kernels), one matrix-pipe utilisation pass.
# usage (from the repository root on the box): bash tools/profile_round.sh <round tag, e.g. r03>
set -e
tag=${1:-r03}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$tag -- python3 $root/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/stats_$tag.json 2> $out/stats_$tag.err
cp $(find $out/stats_$tag -name "*kernel_stats.csv" | head -1) $out/${tag}_bench_kernel_stats.csv
echo stats done
for wl in streamspot_batch theia; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/tr_${tag}_$wl -- python3 $root/bench.py --steps 3 --warmup 2 --no-knn --no-cpu-baseline --no-entity --no-configs --workload $wl > $out/tr_${tag}_$wl.json 2> $out/tr_${tag}_$wl.err
  python3 $root/tools/step_trace.py $(find $out/tr_${tag}_$wl -name "*kernel_trace.csv" | head -1) > $out/${tag}_step_trace_$wl.txt
  echo trace $wl done
done
cd $root
export MAGIC_BENCH_SEQUENCE=$out/seq_ss.json
bash tools/pmc_pass.sh rd_ss FETCH_SIZE --no-configs && bash tools/pmc_pass.sh wr_ss WRITE_SIZE --no-configs
export MAGIC_BENCH_SEQUENCE=$out/seq_th.json
bash tools/pmc_pass.sh rd_th FETCH_SIZE --workload theia && bash tools/pmc_pass.sh wr_th WRITE_SIZE --workload theia
unset MAGIC_BENCH_SEQUENCE
echo gat pmc done
MAGIC_PROFILE_ROUND=$tag python3 tools/make_traffic.py gpurun_out/pmc_rd_ss gpurun_out/pmc_wr_ss streamspot_batch 96000 4 $out/seq_ss.json > /dev/null
MAGIC_PROFILE_ROUND=$tag python3 tools/make_traffic.py gpurun_out/pmc_rd_th gpurun_out/pmc_wr_th theia 700000 3 $out/seq_th.json > /dev/null
# KNN: a 131,072-query search against the full train set is one full round of the filter
bash tools/pmc_knn.sh rd_knn FETCH_SIZE && bash tools/pmc_knn.sh wr_knn WRITE_SIZE
MAGIC_PROFILE_ROUND=$tag python3 tools/make_traffic.py gpurun_out/pmc_rd_knn gpurun_out/pmc_wr_knn knn 0 0 > /dev/null
echo knn pmc done
# matrix-pipe utilisation: projections of both workloads + the KNN filter
bash tools/pmc_pass.sh mfma_ss "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" --no-configs
bash tools/pmc_pass.sh mfma_th "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" --workload theia
bash tools/pmc_knn.sh mfma_knn "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVES"
python3 tools/pmc_mfma.py gpurun_out/pmc_mfma_ss gpurun_out/pmc_mfma_th gpurun_out/pmc_mfma_knn > profiles/${tag}_pmc_mfma.txt
cp profiles/traffic.json profiles/${tag}_pmc_summary_*.txt profiles/${tag}_pmc_mfma.txt gpurun_out/
# the raw traces and counter dumps stay on the box (only 64 MiB travel back)
rm -rf $out/stats_$tag $out/tr_${tag}_* $out/pmc_rd_* $out/pmc_wr_* $out/pmc_mfma_*
echo profile set $tag done
