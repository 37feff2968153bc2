#!/bin/bash
T=gpurun_out/r4t; mkdir -p $T; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$T/prof -o host -- python $GRAFT_REPO_ROOT/tools/gpu_host_stream_trace.py > $GRAFT_REPO_ROOT/$T/run.log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT; f=$(ls $T/prof/*/*kernel_stats.csv $T/prof/*kernel_stats.csv 2>/dev/null | head -1); echo $f; head -30 $f | cut -c1-200
