#!/bin/bash
# Executable-graph cache: recurring signatures (8: all replayed), more signatures than the cache holds (eviction, parked forked
# executables re-targeted), the ever-new-signature stress of round 3, the call-cache tests.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export RK_BACKTRACE=1 PYTHONFAULTHANDLER=1
echo "== 8 recurring signatures, one-launch (linear) graphs"; timeout 600 python3 tools/stress_graph_recurring.py 6000 8 2>&1 | grep -a "stress ok\|Error\|MISMATCH\|fault" | tail -3
echo "== 8 recurring signatures, forked graphs"; RK_PLAN=0 RK_ANY_FIRST=0 timeout 600 python3 tools/stress_graph_recurring.py 6000 8 2>&1 | grep -a "stress ok\|Error\|MISMATCH\|fault" | tail -3
echo "== 20 signatures, forked, cap 4 (eviction + re-targeting)"; RK_PLAN=0 RK_ANY_FIRST=0 RK_GRAPH_FORKED_MAX=4 timeout 900 python3 tools/stress_graph_recurring.py 3000 20 2>&1 | grep -a "stress ok\|Error\|MISMATCH\|fault" | tail -3
echo "== 20 signatures, forked, default cap"; RK_PLAN=0 RK_ANY_FIRST=0 timeout 900 python3 tools/stress_graph_recurring.py 3000 20 2>&1 | grep -a "stress ok\|Error\|MISMATCH\|fault" | tail -3
echo "== 20 signatures, forked, cap 4, no re-targeting"; RK_PLAN=0 RK_ANY_FIRST=0 RK_GRAPH_FORKED_MAX=4 RK_GRAPH_UPDATE=0 timeout 900 python3 tools/stress_graph_recurring.py 3000 20 2>&1 | grep -a "stress ok\|Error\|MISMATCH\|fault" | tail -3
echo "== round 3's stress (new signature every second call), forked"; RK_PLAN=0 timeout 900 python3 tools/stress_graph_capture.py 3000 2>&1 | grep -a "stress ok\|Error\|MISMATCH\|fault" | tail -3
echo "== round 3's stress, default plans"; timeout 900 python3 tools/stress_graph_capture.py 3000 2>&1 | grep -a "stress ok\|Error\|MISMATCH\|fault" | tail -3
timeout 1200 python3 -m pytest tests/test_gpu_call_caches.py -m gpu -x -q 2>&1 | tail -4
