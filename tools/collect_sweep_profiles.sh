# rocprofv3 kernel-trace statistics of the preconditioned BASELINE configs (host CLI).
#   bash tools/collect_sweep_profiles.sh <tag>   -> gpurun_out/sweeps_<tag>/
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/sweeps_$TAG
B=$R/basic_iterative_solvers_amd/host/basic_iterative_solvers
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for args in "anderson:256,shift=9 -gm -p gs" "anderson:256,shift=9 -gm -p gs -perm mc" "fem:80,80,81 -bi -p ilu0" "fem:80,80,81 -bi -p ilu0 -perm mc" "hpcg:256 -cg -p sgs -perm mc"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/run$i -o r -- $B $args > $O/run$i.out 2> $O/run$i.err
  echo "# $args" > $O/config${i}_kernel_stats.csv
  cat $(find $O/run$i -name "*kernel_stats.csv") >> $O/config${i}_kernel_stats.csv
  grep -E "converged|did not" $O/run$i.out | tail -n 1
  head -n 5 $O/config${i}_kernel_stats.csv | cut -c1-260
  rm -rf $O/run$i
done
