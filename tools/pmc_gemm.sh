#!/bin/bash
# PMC counters for the GEMM variants on one layer shape (run on the GPU box through gpurun).
set -e
export TMPDIR=/tmp
OUT=gpurun_out/pmc_gemm
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/sq -- python3 tools/bench_gemm.py --variants ${1:-0,2} --reps 3 --no-check > $OUT/run_sq.log 2>&1
python3 - <<'PY'
import csv, glob, collections
rows=[]
for f in glob.glob('gpurun_out/pmc_gemm/sq/**/*counter_collection.csv', recursive=True):
    rows += list(csv.DictReader(open(f)))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=(r['Kernel_Name'][:60], r.get('Grid_Size'), r.get('LDS_Block_Size'))
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if 'gemm' not in k[0]: continue
    n=sum(1 for r in rows if (r['Kernel_Name'][:60], r.get('Grid_Size'), r.get('LDS_Block_Size'))==k and r['Counter_Name']=='SQ_WAVE_CYCLES')
    print(k, 'dispatches', n)
    wc=v['SQ_WAVE_CYCLES']
    for c,val in sorted(v.items()):
        print('   %-28s %14.0f  %6.3f of WAVE_CYCLES' % (c, val/n, val/wc if wc else 0))
PY
