# every solver x preconditioner of the CLI on one generated input: iterations and iterate time
B=$GRAFT_REPO_ROOT/basic_iterative_solvers_amd/host/basic_iterative_solvers
M=${1:-hpcg:128}
for s in cg bi gm j gs sgs; do
  for p in none j gs bgs sgs ilu0 2st s2st; do
    case $s in j|gs|sgs) [ "$p" != none ] && continue;; esac
    if [ "$p" = none ]; then pa=""; else pa="-p $p"; fi
    out=$(timeout -k 5 120 $B $M -$s $pa 2>&1)
    it=$(echo "$out" | grep -E "converged in|did not converge" | tail -n 1 | sed 's/Solver: //')
    t=$(echo "$out" | grep "Iterate time" | tail -n 1 | awk '{print $5}')
    echo "$s/$p: $t  $it"
  done
done
