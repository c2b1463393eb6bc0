#!/bin/bash
# usage: pmc_knn.sh <tag> "<counters>"
# One rocprofv3 --pmc pass over the KNN leg alone (131,072 queries = one round of the filter, full train set).
tag=$1; ctr=$2
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --knn-only --knn-queries 131072 --knn-steps 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
