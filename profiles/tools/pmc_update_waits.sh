export TMPDIR=/tmp
W=/tmp/pmcu; rm -rf $W; mkdir -p $W
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_FLAT GRBM_GUI_ACTIVE SQ_WAIT_INST_VMEM"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $W/p$i -o p -- python3 profiles/diag/upd_ramp.py > /dev/null 2> $W/p$i.err
  f=$(find $W/p$i -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && (head -1 $f; grep -E "k_gxt_dma|k_apply_dma" $f) > $W/p$i.csv || tail -3 $W/p$i.err
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("/tmp/pmcu/p*.csv")):
    for r in csv.DictReader(open(f)):
        k = "gxt" if "gxt" in r["Kernel_Name"] else "apply"
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:30s} n={len(v):4d} median={sorted(v)[len(v)//2]:.4g}")
PY
