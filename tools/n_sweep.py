"""Population-size sweep of the headline kernel: leapfrog/s against N across the
one-wavefront-per-SIMD point (N = 65 536 = 1 024 wavefronts = the chip's 1 024 SIMDs).
    python tools/n_sweep.py [steps] [warmup]   ->  one line per N (JSON), then a table"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
K = sys.argv[1] if len(sys.argv) > 1 else "20"
W = sys.argv[2] if len(sys.argv) > 2 else "5"
rows = []
for n in [int(v) for v in os.environ.get("SWEEP_N", "32768,65536,65600,73728,98304,131072,196608,262144").split(",")]:
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", K, "--warmup", W, "--particles", str(n),
                        "--no-cpu-baseline", "--no-end-to-end", "--repeats", "3", "--settle-ms", "150"]
                       + (["--lane-segments", os.environ["SWEEP_SEGS"]] if "SWEEP_SEGS" in os.environ else []),
                       capture_output=True, text=True)
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    if p.returncode != 0 or not line:
        print(f"N={n}: bench failed: {p.stderr[-400:]}")
        continue
    d = json.loads(line[-1])
    rows.append((n, d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["leapfrogs_per_particle_step"]))
    print(json.dumps({"N": n, "value": d["value"], "ms_per_step": d["ms_per_step"], "nuts_launch_ms": d["roofline"]["avg_launch_ms"],
                      "leapfrogs_per_particle_step": d["leapfrogs_per_particle_step"]}))
print(f"\n{'N':>8} {'wavefronts':>10} {'G leapfrog/s':>13} {'ms/step':>9} {'NUTS launch ms':>15} {'ns per particle-leapfrog':>25}")
for n, v, ms, lm, lpp in rows:
    print(f"{n:8d} {(n + 63) // 64:10d} {v / 1e9:13.3f} {ms:9.4f} {lm:15.3f} {1e9 / v:25.4f}")
