#!/bin/bash
# HBM traffic of the NUTS kernel: FETCH_SIZE and WRITE_SIZE in separate passes
# (MI355X_MICROARCH.md: TCC slots; FETCH_SIZE reads 1/2 of a wide coalesced stream on gfx950).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_traffic
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT -o $n -- python3 bench.py --steps 4 --warmup 6 --fuse-max 1 --no-cpu-baseline > $OUT/$n.log 2>&1 || { tail -3 $OUT/$n.log; }
done
python3 - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc_traffic/*_counter_collection.csv')):
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-70:]
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,d in acc.items():
        if 'nuts2_kernel' in k or 'prep' in k or 'post' in k:
            for cn,v in d.items(): print(f"{k:42s} {cn:20s} mean_last4={sum(v[-4:])/4:.4e}")
PY
