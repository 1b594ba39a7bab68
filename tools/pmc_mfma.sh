#!/bin/bash
# matrix-core counters of the Gram build (rocprofv3 --pmc, own pass, no tracing besides the kernel trace)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d /root/repo/gpurun_out/pmc_mfma -- \
  python3 /root/repo/tools/gram_time.py > /root/repo/gpurun_out/pmc_mfma.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob('/root/repo/gpurun_out/pmc_mfma/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'gram' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0]][r['Counter_Name']] += float(r['Counter_Value']); cnt[(r['Kernel_Name'].split('(')[0], r['Counter_Name'])] += 1
for k, v in agg.items():
    print(k)
    for c, x in sorted(v.items()):
        print('   %-32s %.5g per launch' % (c, x / cnt[(k, c)]))
PY
