#!/bin/bash
# rocprofv3 kernel stats + HBM-traffic counters of the mapping kernels alone (configs C and E), run on the GPU box via gpurun:
#   gpurun -- 'bash tools/prof_mapping.sh'   ->  gpurun_out/prof_mapping/{stats.csv,summary.json}
# Separate --pmc passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), kernel-trace only.
export TMPDIR=/tmp
OUT=gpurun_out/prof_mapping
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 tools/bench_mapping.py > $OUT/s.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- python3 tools/bench_mapping.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p2 -- python3 tools/bench_mapping.py > $OUT/p2.log 2>&1
grep "us/frame" $OUT/s.log
cp $(find $OUT/s -name "*kernel_stats.csv" | head -1) $OUT/stats.csv
python3 - <<'PY'
import csv, glob, collections, json, re
def fam(name):
    m = re.search(r"k_[a-z0-9_]+(<[^>]*>)?", name)
    return m.group(0) if m else name[:40]
# per kernel AND grid size (config C and E launch the same kernels with different grids)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob('gpurun_out/prof_mapping/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_fused_vote' not in r['Kernel_Name'] and 'k_grid_' not in r['Kernel_Name']:
            continue
        k = (fam(r['Kernel_Name']), int(r['Grid_Size']))
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'FETCH_SIZE': n[k] += 1
dur = collections.defaultdict(list)
for f in glob.glob('gpurun_out/prof_mapping/s/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_fused_vote' in r['Kernel_Name'] or 'k_grid_' in r['Kernel_Name']:
            dur[(fam(r['Kernel_Name']), int(r.get('Grid_Size') or r['Grid_Size_X']))].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
out = []
for k, v in sorted(agg.items()):
    c = max(1, n[k]); d = dur.get(k, [0])
    # FETCH_SIZE / WRITE_SIZE are in KB.  FETCH_SIZE counts 1/2 of the bytes of wide coalesced reads on gfx950 (MI355X_MICROARCH.md):
    # reported raw AND doubled; these kernels mix 16-B point loads (x2 applies) with scattered 4-B accesses (uncalibrated).
    row = dict(kernel=k[0], grid=k[1], launches=c, avg_us=sum(d) / len(d) / 1e3, fetch_MB_raw=v['FETCH_SIZE'] / 1024 / c,
               fetch_MB_x2=2 * v['FETCH_SIZE'] / 1024 / c, write_MB=v['WRITE_SIZE'] / 1024 / c,
               l2_hit=v['TCC_HIT_sum'] / max(1.0, v['TCC_HIT_sum'] + v['TCC_MISS_sum']))
    out.append(row)
    print("%-40s grid %9d  %7.2f us  fetch %7.2f MB (x2 %7.2f)  write %7.2f MB  L2 hit %.3f" % (k[0], k[1], row['avg_us'], row['fetch_MB_raw'], row['fetch_MB_x2'], row['write_MB'], row['l2_hit']))
json.dump(out, open('gpurun_out/prof_mapping/summary.json', 'w'), indent=1)
PY
