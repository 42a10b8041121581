# PMC passes (one counter group per run) for a list of SpMV configurations.
#   bash tools/pmc_sweep.sh <outdir-name> "<cfg1>" "<cfg2>" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
shift
mkdir -p $O
GROUPS_=(
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"
 "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCR_TCP_STALL_CYCLES_sum"
 "TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
 "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"
 "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS"
 "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD"
 "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
 "TA_FLAT_READ_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum"
)
i=0
for cfg in "$@"; do
  i=$((i+1)); g=0
  for grp in "${GROUPS_[@]}"; do
    g=$((g+1))
    rocprofv3 --pmc $grp -d $O/c${i}_g${g} -o r --output-format csv -- python3 $R/tools/spmv_pmc_run.py hpcg 256 "$cfg" 3 > $O/c${i}_g${g}.log 2>&1 || echo "cfg $i group $g failed"
  done
  echo "c$i = $cfg" >> $O/summary.txt
done
python3 $R/tools/pmc_summarise.py $O >> $O/summary.txt 2>&1
cat $O/summary.txt
find $O -name "*.csv" ! -name "*counter_collection.csv" -delete
