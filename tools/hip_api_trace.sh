R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --stats --output-format csv -d $O/hiptr -o r -- $R/basic_iterative_solvers_amd/host/basic_iterative_solvers hpcg:256 -cg -p sgs -perm mc > $O/hiptr.log 2>&1
f=$(find $O/hiptr -name "*hip_api_stats.csv"); cp $f $O/hiptr_api_stats.csv; rm -rf $O/hiptr
