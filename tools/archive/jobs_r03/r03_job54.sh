#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
echo "default:        $(timeout 600 python3 tools/size_scan.py 2>&1 | tail -1)"
echo "forked graphs:  $(RK_GRAPH_FORKED=1 timeout 600 python3 tools/size_scan.py 2>&1 | tail -1)"
echo "no rev window:  $(RK_PLAN_REV_MAX_GROUPS=0 timeout 600 python3 tools/size_scan.py 1.5e6,2e6 2>&1 | tail -1)"
echo "no rev, graphs: $(RK_PLAN_REV_MAX_GROUPS=0 RK_GRAPH_FORKED=1 timeout 600 python3 tools/size_scan.py 1.5e6,2e6 2>&1 | tail -1)"
echo "serial classes: $(RK_SERIAL_CLASSES=1 timeout 600 python3 tools/size_scan.py 2.5e6,4e6 2>&1 | tail -1)"
