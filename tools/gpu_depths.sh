#!/bin/bash
# loop depth A/B in fresh processes (tools/loop_run.py prints the wall time per call): headline at depths 1 2 3, 128 columns at 1 2
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
	for d in 1 2 3; do timeout -k 10 120 python tools/loop_run.py 1000 1048576 64 fp32_tc_cor 0 0 $d | sed "s/^/depth $d: /"; done
	for d in 1 2; do timeout -k 10 120 python tools/loop_run.py 600 1048576 128 fp32_tc_cor 0 0 $d | sed "s/^/depth $d: /"; done
done
