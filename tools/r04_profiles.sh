# Round-4 profile evidence (kernel statistics + counter groups, one group per run, never with trace domains).
#   bash tools/r04_profiles.sh <tag>   -> gpurun_out/r04_<tag>/
TAG=$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B=$R/basic_iterative_solvers_amd/host/basic_iterative_solvers
GROUPS_=(
 "FETCH_SIZE"
 "WRITE_SIZE"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY"
)
run_cfg() { # name, pmc (0/1), args...
  name=$1; pmc=$2; shift; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_trace -o r -- $B "$@" > $O/${name}.out 2> $O/${name}.err || echo "trace $name failed"
  echo "# $*" > $O/${name}_kernel_stats.csv
  cat $(find $O/${name}_trace -name "*kernel_stats.csv") >> $O/${name}_kernel_stats.csv
  grep -E "converged|did not|Iterate time|Factor time|Preprocessing" $O/${name}.out | tail -n 5
  head -n 6 $O/${name}_kernel_stats.csv | cut -c1-200
  rm -rf $O/${name}_trace
  if [ "$pmc" = "1" ]; then
    g=0
    for grp in "${GROUPS_[@]}"; do
      g=$((g+1))
      timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/${name}_g$g -o r -- $B "$@" > $O/${name}_g$g.log 2>&1 || echo "pmc $name group $g failed"
      f=$(find $O/${name}_g$g -name "*counter_collection.csv")
      [ -n "$f" ] && python3 $R/tools/pmc_summary_csv.py $f > $O/${name}_pmc_g$g.csv
      rm -rf $O/${name}_g$g $O/${name}_g$g.log
    done
    cat $O/${name}_pmc_g*.csv | grep -i "^kernel\|trsv\|sptrsv\|ilu0" > $O/${name}_pmc.csv
    rm -f $O/${name}_pmc_g*.csv
  fi
}
# config 5 as named: an unstructured matrix, BiCGSTAB + ILU(0); RCM-ordered (the realistic pipeline), as generated, and with the round-1 kernels
run_cfg unstr_rcm_bi_ilu0 1 unstr:80,80,80 -bi -p ilu0 -perm rcm &&
run_cfg unstr_rcm_bi_ilu0_wave 1 unstr:80,80,80 -bi -p ilu0 -perm rcm -trsv wave &&
run_cfg unstr_asis_bi_ilu0 0 unstr:80,80,80 -bi -p ilu0 &&
run_cfg unstr_rcm_gm_gs 0 unstr:80,80,80 -gm -p gs -perm rcm &&
run_cfg fem_bi_ilu0 0 fem:80,80,81 -bi -p ilu0 &&
run_cfg fem_bi_ilu0_chain 0 fem:80,80,81 -bi -p ilu0 -trsv level
ls -la $O
