import re,collections,sys
agg=collections.defaultdict(lambda:[0,0,0,0])
for l in open(sys.argv[1]):
    m=re.search(r'M (\d+) N (\d+) K (\d+) nmx (\d) (\w+) \(.*sub-step (\d+) .*epilogue (\d+) \(x ([\d.]+)',l)
    if m:
        k=(int(m.group(2)),int(m.group(3)),m.group(5)); a=agg[k]; a[0]+=int(m.group(6)); a[1]+=int(m.group(7)); a[2]+=float(m.group(8)); a[3]+=1
for k,a in sorted(agg.items()):
    print('N %4d K %4d %-5s: sub-step %5.0f cycles, epilogue %6.0f cycles, tiles per wave %.1f  (n=%d)'%(k[0],k[1],k[2],a[0]/a[3],a[1]/a[3],a[2]/a[3],a[3]))
