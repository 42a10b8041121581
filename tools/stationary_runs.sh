# -j / -gs / -sgs as solvers at full size through the host CLI: the device schedules (default) against the reference's
# kernel-by-kernel order with a blocking norm per iteration (-unfused).  BIS_TIMERS_SYNC=0 as in tools/config_runs.sh.
B=$GRAFT_REPO_ROOT/basic_iterative_solvers_amd/host/basic_iterative_solvers
O=$GRAFT_REPO_ROOT/gpurun_out/stationary.log
: > $O
run() { echo "=== $*" >> $O; ( time timeout -k 10 280 $B "$@" ) 2>&1 | grep -E "converged|did not converge|Total elapsed|Preprocessing time|Solve time|Iterate time|Sample time|real" | tail -n 8 >> $O; }
export BIS_TIMERS_SYNC=0
run hpcg:256 -j
run hpcg:256 -j -unfused
run anderson:256,shift=9 -j
run anderson:256,shift=9 -j -unfused
run anderson:256,shift=9 -gs
run anderson:256,shift=9 -gs -unfused
run anderson:256,shift=9 -sgs
run anderson:256,shift=9 -sgs -unfused
run hpcg:128 -sgs
run hpcg:128 -sgs -unfused
run hpcg:256 -cg
run hpcg:256 -cg -p j
run hpcg:256 -cg -p sgs
cat $O
