#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/ab.sh base minr2 tieh unr1_8 unr2_4 unr34_2 w12_8 2>&1 | grep -v amdgpu
