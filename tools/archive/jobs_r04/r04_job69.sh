#!/bin/bash
# LDS per wavefront 5632 -> 5120 bytes (stack of 384 instead of 512 entries): does the allocation granule cap the 7-waves-per-SIMD classes?
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
BENCH_ARGS="--no-pageable-leg" bash tools/ab.sh base exp_stack384
