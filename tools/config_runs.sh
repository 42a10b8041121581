# BASELINE configs 2-5 through the host CLI (full sizes), natural order and multi-colour.
B=$GRAFT_REPO_ROOT/basic_iterative_solvers_amd/host/basic_iterative_solvers
O=$GRAFT_REPO_ROOT/gpurun_out/configs.log
: > $O
run() { echo "=== $*" >> $O; ( time timeout -k 10 280 $B "$@" ) 2>&1 | grep -E "converged|did not converge|Iterate time|SpMV time|Precond. time|Factor time|Init time|real|colours|levels" | tail -n 9 >> $O; }
run anderson:256 -cg
run anderson:256,shift=9 -cg -p j
run anderson:256,shift=9 -gm -p gs
run anderson:256,shift=9 -gm -p gs -perm mc
run fem:80,80,81 -bi -p ilu0
run fem:80,80,81 -bi -p ilu0 -perm mc
run fem:80,80,81 -cg -p j
run hpcg:256 -cg -p sgs -perm mc
cat $O
