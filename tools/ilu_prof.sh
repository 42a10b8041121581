# kernel statistics of tools/ilu_bench.py under rocprofv3:  bash tools/ilu_prof.sh <tag> <spec> <order>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/iluprof_$TAG -o r -- python3 $R/tools/ilu_bench.py $2 $3 > $O/iluprof_$TAG.log 2>&1
grep "ilu0_persistent=" $O/iluprof_$TAG.log
grep -h "ilu0_persistent_kernel\|ilu0_level_wave_kernel" $(find $O/iluprof_$TAG -name "*kernel_stats.csv") | cut -d'"' -f2,3 | cut -c1-60,150-260
rm -rf $O/iluprof_$TAG
