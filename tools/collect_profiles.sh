# rocprofv3 evidence for bench.py (run on the GPU box through gpurun):
#   bash tools/collect_profiles.sh <tag>      -> gpurun_out/prof_<tag>/
# kernel-trace + stats in one run, each PMC group in its own run (never combined with trace domains).
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-target-512 --no-sweeps --no-configs"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r -- $B --steps 50 --warmup 5 > $O/bench_under_rocprof.json 2> $O/trace.err &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o r -- $B --steps 5 --warmup 1 > $O/fetch.json 2> $O/fetch.err &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o r -- $B --steps 5 --warmup 1 > $O/write.json 2> $O/write.err &&
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -o r -- $B --steps 5 --warmup 1 > $O/tcc.json 2> $O/tcc.err &&
timeout -k 10 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_EA0_RDREQ_sum --output-format csv -d $O/tcp -o r -- $B --steps 5 --warmup 1 > $O/tcp.json 2> $O/tcp.err
F=$(find $O/fetch -name "*counter_collection.csv"); W=$(find $O/write -name "*counter_collection.csv"); T=$(find $O/tcc -name "*counter_collection.csv"); P=$(find $O/tcp -name "*counter_collection.csv")
python3 $R/tools/pmc_summary_csv.py $F > $O/pmc_fetch_summary.csv
python3 $R/tools/pmc_summary_csv.py $W > $O/pmc_write_summary.csv
python3 $R/tools/pmc_summary_csv.py $T > $O/pmc_tcc_summary.csv
python3 $R/tools/pmc_summary_csv.py $P > $O/pmc_tcp_summary.csv
python3 $R/tools/pmc_traffic.py $F $W $T 256 > $O/spmv_traffic.json
cp $(find $O/trace -name "*kernel_stats.csv") $O/kernel_stats.csv
rm -rf $O/trace $O/fetch $O/write $O/tcc $O/tcp
cat $O/bench_under_rocprof.json | cut -c1-400; head -4 $O/kernel_stats.csv; cat $O/spmv_traffic.json
