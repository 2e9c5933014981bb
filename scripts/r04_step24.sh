#!/bin/bash
set -u
mkdir -p gpurun_out
O=gpurun_out/r04_s24.txt; : > $O
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o pp -- python $OLDPWD/scripts/gpu_k2_uniform.py 13 > /dev/null 2>&1
python - <<'PY' >> $OLDPWD/$O
import csv,glob
f=glob.glob('/tmp/pp/**/pp_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r['Name'][:70].ljust(70), r['Calls'], r['AverageNs'], r['MinNs'])
PY
cat $OLDPWD/$O
