#!/bin/bash
# SQ / LDS / HBM counters of k_bottleneck alone at layer1's 1080p shape (run on the GPU box through gpurun).
# Separate --pmc passes, kernel-trace only (TCC: FETCH_SIZE costs 3 slots, WRITE_SIZE 2).
export TMPDIR=/tmp
OUT=gpurun_out/pmc_bn
rm -rf $OUT; mkdir -p $OUT
ARGS="tools/bench_bottleneck.py --reps 3"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p3 -- python3 $ARGS > $OUT/p3.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p4 -- python3 $ARGS > $OUT/p4.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for f in glob.glob('gpurun_out/pmc_bn/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_bottleneck' not in r['Kernel_Name']: continue
        k=r['Kernel_Name'][r['Kernel_Name'].index('k_bottleneck'):][:40]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
out={}
for k,v in sorted(agg.items()):
    wc=v['SQ_WAVE_CYCLES']/max(1,n[k]['SQ_WAVE_CYCLES'])
    print(k)
    row={}
    for c,val in sorted(v.items()):
        per=val/max(1,n[k][c]); row[c]=per
        extra = "  %6.3f of WAVE_CYCLES" % (per/wc) if c.startswith('SQ_') and wc else ("  (x2 for wide reads: %.1f MB)" % (2*per/1024) if c=='FETCH_SIZE' else ("  %.1f MB" % (per/1024) if c=='WRITE_SIZE' else ""))
        print("   %-28s %16.0f%s" % (c, per, extra))
    out[k]=row
json.dump(out, open('gpurun_out/pmc_bn/summary.json','w'), indent=1)
PY
