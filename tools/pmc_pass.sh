#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <counters...> -- <bench args...>
# One rocprofv3 counter pass over one frame of bench.py; the per-kernel sums go to gpurun_out/pmc_<tag>.csv.
set -e
tag=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
timeout -s KILL 300 rocprofv3 --pmc "${ctrs[@]}" -d /tmp/pmc_$tag -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > /tmp/pmc_$tag.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmc_sum.py /tmp/pmc_$tag $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.csv "rocprofv3 --pmc ${ctrs[*]} -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline $*; sums over all dispatches of each kernel in the run (column dispatches)"
