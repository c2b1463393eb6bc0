#!/bin/bash
# usage: pmc_pass.sh <tag> <counter> [bench.py arguments...]
# One rocprofv3 --pmc pass (a single counter) over a short bench run of one workload.
tag=$1; ctr=$2; shift 2
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-knn --no-cpu-baseline --no-entity "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
