#!/bin/bash
# Runs ON THE GPU BOX: one rocprofv3 --pmc pass (its own run, --kernel-trace only) of a command, reduced to one row per
# (kernel, counter).   bash tools/pmc_pass.sh <name> "<counters>" python3 bench.py --steps 300 ...
set -o pipefail
NAME=$1; CTRS=$2; shift 2
OUT=gpurun_out/pmc_$NAME
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -- "$@" > $OUT.log 2>&1 || echo "pass failed"
f=$(find $OUT -name "*counter_collection.csv" | head -1)
[ -n "$f" ] && python3 tools/pmc_reduce.py "$f" $OUT.reduced.csv && rm -rf $OUT
