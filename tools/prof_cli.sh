# rocprofv3 kernel stats of one host-CLI run:  bash tools/prof_cli.sh <tag> <cli args...>   -> gpurun_out/prof_cli_<tag>_kernel_stats.csv
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cli_$TAG -o r -- $R/basic_iterative_solvers_amd/host/basic_iterative_solvers "$@" > $O/prof_cli_$TAG.log 2>&1
cp $(find $O/prof_cli_$TAG -name "*kernel_stats.csv") $O/prof_cli_${TAG}_kernel_stats.csv
rm -rf $O/prof_cli_$TAG
