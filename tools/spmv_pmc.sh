# Counter passes for the SpMV kernel (one group per run, never combined with trace domains; the program itself follows `--`).
#   bash tools/spmv_pmc.sh <tag> [valdict] [hpcg|anderson|fem] [size]   -> gpurun_out/spmv_pmc_<tag>/
TAG=$1
VD=${2:--1}
KIND=${3:-hpcg}
SIZE=${4:-256}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/spmv_pmc_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
GROUPS_=(
 "FETCH_SIZE"
 "WRITE_SIZE"
 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
 "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU"
 "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES"
 "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
 "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_SMEM SQ_INSTS_FLAT"
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE"
 "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o r -- python3 $R/tools/spmv_probe.py $KIND $SIZE 20 $VD > $O/trace.log 2>&1 || echo "trace failed"
cp $(find $O/trace -name "*kernel_stats.csv") $O/kernel_stats.csv
rm -rf $O/trace
g=0
for grp in "${GROUPS_[@]}"; do
  g=$((g+1))
  # PMC_GROUPS="1 2 10": only those counter groups (default: all)
  if [ -n "$PMC_GROUPS" ] && ! echo " $PMC_GROUPS " | grep -q " $g "; then continue; fi
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/g$g -o r -- python3 $R/tools/spmv_probe.py $KIND $SIZE 20 $VD > $O/g$g.log 2>&1 || echo "pmc group $g failed: $grp"
  f=$(find $O/g$g -name "*counter_collection.csv")
  [ -n "$f" ] && python3 $R/tools/pmc_summary_csv.py $f > $O/pmc_g$g.csv
  if [ -n "$f" ] && { [ $g -le 2 ] || [ $g -eq 10 ]; }; then cp $f $O/raw_g$g.csv; fi
  rm -rf $O/g$g
done
cat $O/pmc_g*.csv | grep -i "spmv_\|Kernel" > $O/summary.txt
# per-launch HBM bytes of the SpMV kernels of this run (FETCH_SIZE correction from the bench passes' calibration)
CF=${TRAFFIC_CAL:-$R/profiles/spmv_traffic.json}
python3 $R/tools/pmc_traffic.py $O/raw_g1.csv $O/raw_g2.csv $O/raw_g10.csv $SIZE --correction-from $CF > $O/spmv_traffic.json
rm -f $O/raw_g*.csv
ls -la $O
