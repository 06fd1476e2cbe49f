#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_occ
mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT -o occ -- python3 bench.py --steps 4 --warmup 6 --no-cpu-baseline > $OUT/occ.log 2>&1 || { tail -5 $OUT/occ.log; exit 1; }
python3 - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/pmc_occ/occ_counter_collection.csv')))
acc=collections.defaultdict(list)
for r in rows:
    if 'nuts2_kernel' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
        lds=r.get('LDS_Block_Size'); vg=r.get('VGPR_Count'); ag=r.get('Accum_VGPR_Count'); sg=r.get('SGPR_Count'); grid=r.get('Grid_Size'); wg=r.get('Workgroup_Size')
for k,v in acc.items(): print(k, sum(v[-4:])/4)
print('LDS',lds,'VGPR',vg,'AGPR',ag,'SGPR',sg,'grid',grid,'wg',wg)
PY
