# BASELINE configs 2-5 through the host CLI (full sizes): natural order (tiled sweeps, the default on these generated matrices, and level-scheduled ones), multi-colour,
# device- against host-scalar GMRES / BiCGSTAB.  BIS_TIMERS_SYNC=0: the timer tree does not drain the stream per call.
B=$GRAFT_REPO_ROOT/basic_iterative_solvers_amd/host/basic_iterative_solvers
O=$GRAFT_REPO_ROOT/gpurun_out/configs.log
: > $O
run() { echo "=== $*" >> $O; ( time timeout -k 10 280 $B "$@" ) 2>&1 | grep -E "converged|did not converge|Total elapsed|Preprocessing time|Solve time|Iterate time|SpMV time|Precond. time|Orthog. time|Factor time|real|colours|reordering" | tail -n 12 >> $O; }
export BIS_TIMERS_SYNC=0
run anderson:256 -cg
run anderson:256,shift=9 -cg -p j
run anderson:256,shift=9 -gm -p gs
run anderson:256,shift=9 -gm -p gs -hostscalars
run anderson:256,shift=9 -gm -p gs -trsv level
run anderson:256,shift=9 -gm -p gs -perm mc
run anderson:256,shift=9 -gm -p gs -perm rcm
run fem:80,80,81 -bi -p ilu0
run fem:80,80,81 -bi -p ilu0 -hostscalars
run fem:80,80,81 -bi -p ilu0 -trsv level
run fem:80,80,81 -bi -p ilu0 -perm mc
run fem:80,80,81 -cg -p j
run hpcg:256 -cg -p sgs -perm mc
run hpcg:128 -cg -p sgs
run hpcg:128 -cg -p sgs -trsv level
run anderson:256,shift=9 -gs
run anderson:256,shift=9 -gs -trsv level
export BIS_TIMERS_SYNC=1
run anderson:256,shift=9 -gm -p gs
run fem:80,80,81 -bi -p ilu0
cat $O
