#!/bin/bash
# HBM-traffic and SQ counters per kernel for one segmentation forward (run on the GPU box via gpurun).
# Separate --pmc passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), kernel-trace only.
# usage: bash tools/pmc_seg.sh [precision]   (default mixed)  ->  gpurun_out/pmc_seg/summary.json
export TMPDIR=/tmp
PREC=${1:-mixed}
OUT=gpurun_out/pmc_seg
rm -rf $OUT; mkdir -p $OUT
# usage: bash tools/pmc_seg.sh [precision] [mixed options, e.g. full_split=1]
ARGS="tools/profile_seg.py --reps 1 --top 1 --precision $PREC"
if [ -n "$2" ]; then ARGS="$ARGS --mixed-opts $2"; fi
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- python3 $ARGS > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p2 -- python3 $ARGS > $OUT/p2.log 2>&1
# (the matrix cores' busy cycles; GRBM_GUI_ACTIVE = kernel cycles summed over the 8 XCDs, its own counter block)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- python3 $ARGS > $OUT/p3.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json, re
def fam(name):
    m = re.search(r"k_[a-z0-9_]+", name)
    base = m.group(0) if m else name[:30]
    t = re.search(r"k_gemm_ring<[^,]*, *(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)", name) or re.search(r"k_gemm_ringI[^L]*Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)", name)
    if t and "ring_mx" not in name: base += "<%s,%s,%s,%s,probe %s,nsub %s>" % t.groups()
    return base
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for f in glob.glob('gpurun_out/pmc_seg/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=fam(r['Kernel_Name'])
        agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
        if r['Counter_Name']=='FETCH_SIZE': n[k]+=1
out=[]
print("%-30s %6s %12s %10s %7s %8s %8s %8s %9s" % ("kernel","calls","fetchMB(x2)","writeMB","L2hit","waitany","waitinst","active","mfma_busy"))
for k,v in sorted(agg.items(), key=lambda kv:-kv[1].get('FETCH_SIZE',0)):
    c=n[k]
    if c==0: continue
    hit=v['TCC_HIT_sum']/max(1,(v['TCC_HIT_sum']+v['TCC_MISS_sum']))
    wc=max(1,v['SQ_WAVE_CYCLES'])
    # FETCH_SIZE / WRITE_SIZE are in KB.  gfx950: FETCH_SIZE counts 1/2 of the bytes of wide (16 B/lane) coalesced reads
    # (MI355X_MICROARCH.md, HBM): doubled here; WRITE_SIZE is exact for 16 B/lane stores.
    # mfma_busy: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the chip's 1024 SIMDs) / (kernel cycles x 1024); kernel cycles =
    # GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs; MI355X_MICROARCH.md, DVFS give-back)
    gui=v.get('GRBM_GUI_ACTIVE',0.0)
    row=dict(kernel=k, launches_profiled=c, fetch_MB_per_launch=2*v['FETCH_SIZE']/1024/c, write_MB_per_launch=v['WRITE_SIZE']/1024/c,
             l2_hit=hit, wait_any=v['SQ_WAIT_ANY']/wc, wait_inst=v['SQ_WAIT_INST_ANY']/wc, active=v['SQ_ACTIVE_INST_ANY']/wc,
             mfma_busy=(v.get('SQ_VALU_MFMA_BUSY_CYCLES',0.0)/(gui/8*1024)) if gui else None,
             mfma_insts_per_launch=v.get('SQ_INSTS_MFMA',0.0)/c)
    out.append(row)
    print("%-30s %6d %12.1f %10.1f %7.3f %8.3f %8.3f %8.3f %9s" % (k, c, row['fetch_MB_per_launch'], row['write_MB_per_launch'], hit, row['wait_any'], row['wait_inst'], row['active'], "-" if row['mfma_busy'] is None else "%.3f" % row['mfma_busy']))
json.dump(out, open('gpurun_out/pmc_seg/summary.json','w'), indent=1)
PY
