# rocprofv3 HIP-API and kernel statistics of one host-CLI run:  bash tools/hip_api_trace.sh <tag> <cli args...>
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $O/hiptr_$TAG -o r -- $R/basic_iterative_solvers_amd/host/basic_iterative_solvers "$@" > $O/hiptr_$TAG.log 2>&1
cp $(find $O/hiptr_$TAG -name "*hip_api_stats.csv") $O/hiptr_${TAG}_api_stats.csv
cp $(find $O/hiptr_$TAG -name "*kernel_stats.csv") $O/hiptr_${TAG}_kernel_stats.csv
rm -rf $O/hiptr_$TAG
