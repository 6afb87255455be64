#!/bin/bash
# Sanity sweep of the light-tail fraction at 4M on the final build.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
BENCH_ARGS="--no-pageable-leg" bash tools/ab_env.sh "-" "RK_PLAN_TAIL=0.15" "RK_PLAN_TAIL=0.35" "RK_PLAN_TAIL=0.5"
