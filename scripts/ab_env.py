"""Same-box A/B of one library under two environments (a diagnostic switch read at launch time):
    python scripts/ab_env.py GSM_PROPOSE_DBG=0 GSM_PROPOSE_DBG=16384
Prints the median bench value of AB_REPS short bench.py runs per setting, interleaved (see ab_lib.py for the box-to-box caveat)."""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
settings = sys.argv[1:]
reps = int(os.environ.get("AB_REPS", "3"))
extra = os.environ.get("AB_BENCH_ARGS", "--steps 6 --warmup 1 --no-cpu-baseline --no-extras").split()
vals = {s: [] for s in settings}
for r in range(reps):
    for s in settings:
        env = dict(os.environ)
        for kv in s.split(","):
            k, v = kv.split("=", 1); env[k] = v
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), *extra], env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        vals[s].append(json.loads(out)["value"])
base = sorted(vals[settings[0]])[reps // 2]
for s in settings:
    v = sorted(vals[s])[reps // 2]
    print(f"{s:28s} {v / 1e6:8.3f} M  ({100 * (v / base - 1):+.2f} % vs {settings[0]})  runs: {[round(x / 1e6, 3) for x in vals[s]]}", flush=True)
