# PMC evidence for the natural-order triangular sweeps: one counter group per run,
# never combined with trace domains; the program itself follows `--`.
#   bash tools/trsv_pmc.sh <tag>     -> gpurun_out/trsv_pmc_<tag>/
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/trsv_pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/basic_iterative_solvers_amd/host/basic_iterative_solvers
GROUPS_=(
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE"
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY"
)
run_cfg() { # name, args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_trace -o r -- $B "$@" > $O/${name}_trace.log 2>&1 || echo "trace $name failed"
  cp $(find $O/${name}_trace -name "*kernel_stats.csv") $O/${name}_kernel_stats.csv
  g=0
  for grp in "${GROUPS_[@]}"; do
    g=$((g+1))
    timeout -k 10 280 rocprofv3 --pmc $grp --output-format csv -d $O/${name}_g$g -o r -- $B "$@" > $O/${name}_g$g.log 2>&1 || echo "pmc $name group $g failed"
    f=$(find $O/${name}_g$g -name "*counter_collection.csv")
    [ -n "$f" ] && python3 $R/tools/pmc_summary_csv.py $f > $O/${name}_pmc_g$g.csv
    rm -rf $O/${name}_g$g
  done
  rm -rf $O/${name}_trace
}
run_cfg anderson256_gm_gs anderson:256,shift=9 -gm -p gs &&
run_cfg fem_bi_ilu0 fem:80,80,81 -bi -p ilu0 &&
run_cfg hpcg256_cg_sgs hpcg:256 -cg -p sgs
# one file per configuration: kernel statistics and the counters of the sweep kernels
for n in anderson256_gm_gs fem_bi_ilu0 hpcg256_cg_sgs; do
  cat $O/${n}_pmc_g*.csv | grep -i "Kernel\|trsv\|sptrsv" > $O/${n}_pmc.csv
done
ls -la $O
