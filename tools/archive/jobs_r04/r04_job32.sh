#!/bin/bash
# The GPU suite under the fallback settings of round 4's new paths.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
mkdir -p gpurun_out/r04_job32
i=0
for envs in "RK_GRAPH_UPDATE=0 RK_GRAPH_CACHE=2" "RK_GRAPH=0 RK_STAGE_KEEP=0"; do
  i=$((i+1))
  env $envs timeout 1500 python3 -m pytest tests -m gpu -q -x > gpurun_out/r04_job32/run$i.log 2>&1
  echo "$envs: $(tail -1 gpurun_out/r04_job32/run$i.log)"
done
