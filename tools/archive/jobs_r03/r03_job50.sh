#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do
RK_ANY_FIRST=0 timeout 300 python3 tools/first_call_probe.py 2>&1 | tail -1
timeout 300 python3 tools/first_call_probe.py 2>&1 | tail -1
done
RK_ANY=3 timeout 300 python3 tools/first_call_probe.py 2>&1 | tail -1
RK_ANY=1 timeout 300 python3 tools/first_call_probe.py 2>&1 | tail -1
timeout 300 python3 tools/any_probe3.py 2>&1 | tail -1
