#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
RK_SL_PARTS_BELOW=0 bash tools/pmc_variant.sh gpurun_out/r03_job22/pmc4 4000000 4 5
python3 tools/pmc_digest.py gpurun_out/r03_job22/pmc4/summary.txt | cut -c1-170
