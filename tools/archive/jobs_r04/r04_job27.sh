#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
RK_BRIDGE_TIMING=1 tests/build/cuda_bridge_driver timing 4000000 2>&1 | grep -v amdgpu.ids | head -30

RK_ALIAS_DEVICES=4 tests/build/cuda_bridge_driver | tail -2
