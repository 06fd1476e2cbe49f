cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in v1 p0r0g; do
  export SMCN_LIB=smcnuts_amd/variants/libsmcnuts_w_$v.so
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_$v -o kt -- python3 bench.py --config c5 --steps 2 --warmup 1 --step-size 0.1 --repeats 1 --no-peaks > gpurun_out/kt_$v.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/kt_$v/*kernel_trace.csv")[0]
rows=[r for r in csv.DictReader(open(f)) if "nuts_wave" in r["Kernel_Name"]]
r=rows[-1]
print("$v", {k:r[k] for k in r if k in ("LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Workgroup_Size","Grid_Size")}, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6,"ms")
PY
done
