# which runtime call holds the first device-to-host copy of a sampler built after another's teardown?
# HIP + HSA API trace of bench.py's sequence repeated 10 times; calls longer than 5 ms are listed.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/hsa_stall; rm -rf $OUT; mkdir -p $OUT
python3 tools/dbg/stall_repro.py 10 > $OUT/plain.txt 2>&1; cat $OUT/plain.txt
rocprofv3 --hip-trace --hsa-trace --output-format csv -d $OUT -o t -- python3 tools/dbg/stall_repro.py 10 > $OUT/traced.txt 2> $OUT/traced.err; cat $OUT/traced.txt
python3 - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/*_api_trace.csv"):
    for r in csv.DictReader(open(f)):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if d > 5.0:
            rows.append((int(r["Start_Timestamp"]), d, r["Domain"], r["Function"], r.get("Thread_Id")))
rows.sort()
t0 = rows[0][0] if rows else 0
with open(out + "/long_calls.txt", "w") as fh:
    for s, d, dom, fn, th in rows:
        print(f"{(s - t0) / 1e6:10.2f} ms  {d:8.2f} ms  {dom:20s} {fn}  (thread {th})", file=fh)
print(open(out + "/long_calls.txt").read()[-3000:])
# inside the stalled copies: every API call nested in a hipMemcpyAsync of more than 10 ms (after start-up)
allrows = []
for f in glob.glob(out + "/*_api_trace.csv"):
    for r in csv.DictReader(open(f)):
        allrows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Domain"], r["Function"], r.get("Thread_Id")))
allrows.sort()
with open(out + "/stalled_copy_calls.txt", "w") as fh:
    for s, e, dom, fn, th in allrows:
        if fn == "hipMemcpyAsync" and (e - s) / 1e6 > 10.0 and (s - t0) / 1e6 > 200.0:
            print(f"== hipMemcpyAsync at {(s - t0) / 1e6:.2f} ms, {(e - s) / 1e6:.2f} ms, thread {th}; calls inside it:", file=fh)
            inner = [(s2, e2, d2, f2) for s2, e2, d2, f2, t2 in allrows if s2 >= s and e2 <= e and t2 == th and f2 != "hipMemcpyAsync"]
            for s2, e2, d2, f2 in inner:
                print(f"     +{(s2 - s) / 1e6:8.3f} ms  {(e2 - s2) / 1e6:8.3f} ms  {d2:18s} {f2}", file=fh)
            print(f"     ({len(inner)} calls, {sum(e2 - s2 for s2, e2, _, _ in inner) / 1e6:.3f} ms inside HSA/HIP calls of {(e - s) / 1e6:.2f} ms)", file=fh)
# ... and what the process did in the 12 ms before each of them (HIP calls of every thread, longer than 20 us)
with open(out + "/before_stalled_copy.txt", "w") as fh:
    for s, e, dom, fn, th in allrows:
        if fn == "hipMemcpyAsync" and (e - s) / 1e6 > 10.0 and (s - t0) / 1e6 > 200.0:
            print(f"== before the hipMemcpyAsync at {(s - t0) / 1e6:.2f} ms:", file=fh)
            for s2, e2, d2, f2, t2 in allrows:
                if s - 12e6 <= s2 < s and d2.startswith("HIP") and (e2 - s2) > 20e3:
                    print(f"     {(s2 - s) / 1e6:9.3f} ms  {(e2 - s2) / 1e6:8.3f} ms  thread {t2}  {f2}", file=fh)
print(open(out + "/stalled_copy_calls.txt").read()[-2500:])
print(open(out + "/before_stalled_copy.txt").read()[-5000:])
PY
rm -f $OUT/*_api_trace.csv
