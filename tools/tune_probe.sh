#!/bin/bash
# same-box probe of the apply pass's share settings: blocking calls (loop depth 1) and the chained stream (depth 3), 3000 calls each, interleaved;
# "-" = the library's own measured choice (TSQR_MI_APPLY_SHARES unset), 1 = shares by XCD parity and round (rounds 3-4), 2 = by round only, 0 = equal
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
for v in "-" "TSQR_MI_APPLY_SHARES=1" "TSQR_MI_APPLY_SHARES=0"; do
	[ "$v" = "-" ] && e="" || e="$v"
	b=$(env $e python tools/loop_run.py 3000 1048576 64 fp32_tc_cor 0 0 1 2>/dev/null | tail -1 | sed 's/.*: \([0-9.]* us\).*/\1/')
	s=$(env $e python tools/loop_run.py 3000 1048576 64 fp32_tc_cor 0 0 3 2>/dev/null | tail -1 | sed 's/.*: \([0-9.]* us\).*/\1/')
	echo "[$v] blocking $b | chained $s"
done
done
