"""Same-box A/B of library builds (box-to-box variation is 1-2 %, run-to-run on one box about 0.2 %):
    python scripts/ab_lib.py v0 kb2 ...      # names of mcmc_gpu_amd/libgsm_<name>.so, each timed in its own process
Prints the median bench value of `reps` short bench.py runs per build, interleaved."""
import json, os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
names = sys.argv[1:]
reps = int(os.environ.get("AB_REPS", "3"))
vals = {n: [] for n in names}
for r in range(reps):
    for n in names:
        env = dict(os.environ, GSM_LIB=str(ROOT / "mcmc_gpu_amd" / f"libgsm_{n}.so"))
        out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "6", "--warmup", "1", "--inner", "1024", "--no-cpu-baseline", "--no-extras"],
                             env=env, capture_output=True, text=True).stdout.strip().splitlines()[-1]
        vals[n].append(json.loads(out)["value"])
base = sorted(vals[names[0]])[reps // 2]
for n in names:
    v = sorted(vals[n])[reps // 2]
    print(f"{n:14s} {v / 1e6:8.3f} M chain-steps/s  ({100 * (v / base - 1):+.2f} % vs {names[0]})  runs: {[round(x / 1e6, 3) for x in vals[n]]}")
