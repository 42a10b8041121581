# counters of the win8 kernel per dispatch, K matrices in one process (11 dispatches each): slow against fast allocations
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/win8_place_pmc
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/win8_place_pmc.py 8 > $O/plain.log 2>&1; cat $O/plain.log
i=0
for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUBBLE_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_HIT_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $O/p$i -o r -- python3 $R/tools/win8_place_pmc.py 8 > $O/g$i.log 2>&1 || echo "group $i failed"
  f=$(find $O/p$i -name "*counter_collection.csv")
  [ -n "$f" ] && python3 - "$f" "$O/g$i.log" <<'PY'
import csv, sys, collections, re
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'spmv_win8_kernel' in r['Kernel_Name']]
by=collections.defaultdict(dict)
for r in rows: by[int(r['Dispatch_Id'])][r['Counter_Name']]=float(r['Counter_Value'])
ids=sorted(by)
times=[float(m) for m in re.findall(r"matrix \d+: ([0-9.]+) ms", open(sys.argv[2]).read())]
per=len(ids)//max(len(times),1)
for k,t in enumerate(times):
    chunk=ids[k*per:(k+1)*per]
    if not chunk: continue
    names=sorted(by[chunk[0]])
    avg={n: sum(by[i][n] for i in chunk[1:])/max(len(chunk)-1,1) for n in names}
    print(f"matrix {k}: {t:.4f} ms  " + "  ".join(f"{n}={avg[n]:.4g}" for n in names))
PY
  rm -rf $O/p$i
done
