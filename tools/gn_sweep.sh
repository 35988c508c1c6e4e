#!/bin/bash
# Developer tool: HBM fetch (FETCH_SIZE) per Winograd layer and layer times for the unit-order group sizes (wf_unit):
#   bash tools/gn_sweep.sh "0 1 2 4"
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp APR_DEV_KNOBS=1
cd $R
for gn in $1; do
  export APR_WINO_GN=$gn APR_WINO_S2_GN=$gn
  rm -rf gpurun_out/gn_f
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/gn_f -o p -- python3 bench.py --steps 1 --warmup 0 --no-extras --no-profile --files 2 --seconds 2049 > gpurun_out/gn_f.out 2>&1 || exit 1
  timeout -k 10 200 python3 bench.py --no-extras --steps 2 > gpurun_out/gn_t.json 2>/dev/null || exit 1
  python3 - <<PY
import csv, collections, json
f=collections.OrderedDict()
for r in csv.DictReader(open('gpurun_out/gn_f/p_counter_collection.csv')):
    k=(int(r['Dispatch_Id']), r['Kernel_Name'])
    f[k]=f.get(k,0.0)+float(r['Counter_Value'])
import re
ks=sorted(k for k in f if 'wino' in k[1])[:15]    # the first pass: layers 3..12 in dispatch order (input transform + fused kernel on the stride-1 layers)
d=json.loads([l for l in open('gpurun_out/gn_t.json').read().splitlines() if l.startswith('{')][-1])
ms=[(l['layer'], l['kernel'], l['ms_per_step']) for l in d['layers'] if l['layer'] >= 3]
short=lambda n: re.search(r'apr::([A-Za-z0-9_]+)', n).group(1)
print('gn=$gn', round(d['ms_per_step'],1), ' '.join(f"L{l}:{short(k[1])}:{f[k]*4096/1e9:.2f}GB/{t:.1f}ms" for k,(l,kn,t) in zip(ks,ms)))
PY
done
