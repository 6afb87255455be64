#!/bin/bash
# LDS padding per wavefront (fewer resident waves through the LDS budget: 5632 -> 6144 / 6656 / 7680 bytes = 26 / 24 / 21 per CU).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
BENCH_ARGS="--no-pageable-leg" bash tools/ab.sh base exp_pad512 exp_pad1024 exp_pad2048
