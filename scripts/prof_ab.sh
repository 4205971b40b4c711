#!/bin/bash
# usage: prof_ab.sh <tag> <script and args ...>: rocprofv3 kernel trace of a python script -> gpurun_out/r4/<tag>_stats.csv
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r4
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_$tag
rocprofv3 --kernel-trace --stats -d /tmp/pf_$tag -o $tag -- python3 "$@" > $R/gpurun_out/r4/${tag}_run.log 2>&1 || exit 1
python3 $R/scripts/rocpd_kernel_stats.py $(find /tmp/pf_$tag -name "*.db" | head -1) $R/gpurun_out/r4/${tag}_stats.csv > /dev/null
rm -rf /tmp/pf_$tag
