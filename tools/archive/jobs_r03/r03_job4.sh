#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
bash tools/pmc_variant.sh gpurun_out/r03_job4/pmc4 4000000 4 5
python3 tools/pmc_digest.py gpurun_out/r03_job4/pmc4/summary.txt
bash tools/pmc_variant.sh gpurun_out/r03_job4/pmc2 4000000 2 5
python3 tools/pmc_digest.py gpurun_out/r03_job4/pmc2/summary.txt
