#!/bin/bash
# Cost of re-targeting a four-node executable graph per launch (hipGraphExecKernelNodeSetParams) against direct forked launches.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job11
mkdir -p $O
timeout 120 tools/ubench/build/graph_setparams 2>&1 | tee $O/ubench_graph_setparams.txt
