#!/bin/bash
# Development: FETCH_SIZE (L2 read misses, 64-byte units ... as rocprofv3 reports it) of the conv kernels of a few IResNet-100
# forwards, for the library named in DIF_LIB.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=$1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/fetch_$tag -o p -- python3 tools/f32_ab.py iresnet100 512 > gpurun_out/fetch_$tag.log 2>&1
python3 - <<PY
import csv, collections
f='gpurun_out/fetch_$tag/p_counter_collection.csv'
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    if r['Counter_Name']=='FETCH_SIZE':
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('dif::','')[:60]
        agg[k]+=float(r['Counter_Value']); n[k]+=1
tot=sum(agg.values())
print('$tag total FETCH_SIZE %.3e' % tot)
for k,v in sorted(agg.items(), key=lambda kv:-kv[1])[:5]: print('   %-60s %5d launches %.3e' % (k, n[k], v))
PY
