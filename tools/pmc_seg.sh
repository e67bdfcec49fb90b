#!/bin/bash
# HBM-traffic and SQ counters per kernel for one segmentation forward (run on the GPU box via gpurun).
# Separate --pmc passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), kernel-trace only.
export TMPDIR=/tmp
OUT=gpurun_out/pmc_seg
rm -rf $OUT; mkdir -p $OUT
ARGS="tools/profile_seg.py --reps 1 --top 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/p3 -- python3 $ARGS > $OUT/p3.log 2>&1
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter(); dur=collections.defaultdict(float)
for f in glob.glob('gpurun_out/pmc_seg/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name'] in ('FETCH_SIZE',): n[k]+=1
print("%-42s %5s %10s %10s %8s %8s %8s %8s" % ("kernel","calls","fetchMB*2","writeMB","L2hit","waitany","waitinst","active"))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get('FETCH_SIZE',0)):
    if n[k]==0: continue
    c=n[k]
    hit=v['TCC_HIT_sum']/max(1,(v['TCC_HIT_sum']+v['TCC_MISS_sum']))
    wc=max(1,v['SQ_WAVE_CYCLES'])
    # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 FETCH_SIZE reads 1/2 of wide streaming reads (MI355X_MICROARCH.md HBM) -> doubled
    print("%-42s %5d %10.1f %10.1f %8.3f %8.3f %8.3f %8.3f" % (k, c, 2*v['FETCH_SIZE']/1024/c* (c/ c), v['WRITE_SIZE']/1024/c, hit, v['SQ_WAIT_ANY']/wc, v['SQ_WAIT_INST_ANY']/wc, v['SQ_ACTIVE_INST_ANY']/wc))
PY
