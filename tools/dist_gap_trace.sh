# kernel timeline of the partitioned CG code path on one GPU (one rank's share of a strong-scaled problem, no wire):
#   bash tools/dist_gap_trace.sh <tag> <n1> <nz>   -> gpurun_out/dist_gap_<tag>.txt  (per-kernel durations and the gaps between them)
TAG=$1; N1=$2; NZ=$3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/dgap_$TAG -o r -- python3 $R/tools/dist_overhead.py $N1 $NZ 60 > $O/dgap_$TAG.log 2>&1
python3 $R/tools/dist_gap_summary.py $(find $O/dgap_$TAG -name "*kernel_trace.csv") > $O/dist_gap_$TAG.txt
cat $O/dgap_$TAG.log >> $O/dist_gap_$TAG.txt
rm -rf $O/dgap_$TAG $O/dgap_$TAG.log
