#!/bin/bash
# usage: pmc_cmd.sh <tag> "<counters...>" <python-script> [args...]
# One rocprofv3 --pmc pass (counters must fit one pass) over a short python program.
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc $ctrs --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
