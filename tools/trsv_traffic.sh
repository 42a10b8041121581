# PMC HBM traffic of the natural-order sweeps of bench.py's `sweeps` legs: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3
# runs (MI355X_MICROARCH.md: they do not fit one pass), never combined with trace domains; the program itself follows `--`.
#   bash tools/trsv_traffic.sh <tag> [workloads...]   -> gpurun_out/trsv_traffic_<tag>/trsv_traffic.json (+ kernel stats per workload)
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trsv_traffic_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CASES=${@:-hpcg256 anderson256 fem80x80x81 unstr80_asis unstr80_rcm}
NS=10
for c in $CASES; do
  for d in forward backward; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o r -- python3 $R/tools/sweep_probe.py $c $d $NS > $O/${c}_${d}_trace.log 2>&1 || echo "trace $c $d failed"
    cp $(find $O/t -name "*kernel_stats.csv") $O/${c}_${d}_kernel_stats.csv; rm -rf $O/t
    for grp in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/p -o r -- python3 $R/tools/sweep_probe.py $c $d $NS > $O/${c}_${d}_$grp.log 2>&1 || echo "pmc $c $d $grp failed"
      f=$(find $O/p -name "*counter_collection.csv")
      [ -n "$f" ] && cp $f $O/${c}_${d}_$grp.csv
      rm -rf $O/p
    done
    echo "done $c $d"   # (progress for gpurun's hang detector)
  done
done
python3 $R/tools/trsv_traffic.py $O $NS $R/profiles/spmv_traffic.json > $O/trsv_traffic.json
cat $O/trsv_traffic.json
