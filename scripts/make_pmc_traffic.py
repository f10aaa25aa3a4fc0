"""profiles/pmc_traffic.json from two rocprofv3 --pmc passes of scripts/pmc_run.py (FETCH_SIZE, WRITE_SIZE).

Correction per MI355X_MICROARCH.md (HBM section): the counters are in KiB; FETCH_SIZE under-reports wide coalesced
reads on gfx950, so it is calibrated on gsm::stream_copy_kernel (same 8-byte-per-lane access shape as the step kernel,
exactly 512 MiB read and 512 MiB written); WRITE_SIZE is checked on the same kernel."""
import csv, glob, json, sys
fetch_dir, write_dir, log, out = sys.argv[1:5]
def mean_per_kernel(d, counter):
    acc = {}
    for f in glob.glob(f'{d}/**/*_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                acc.setdefault(r['Kernel_Name'].split('(')[0], []).append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}
F, Wr = mean_per_kernel(fetch_dir, 'FETCH_SIZE'), mean_per_kernel(write_dir, 'WRITE_SIZE')
work = [json.loads(l) for l in open(log) if l.startswith('{')]
plane = work[0]['plane_bytes']
copy = [k for k in F if 'stream_copy' in k][0]
step = [k for k in F if 'step_kernel' in k][0]
prop = [k for k in F if k.endswith('propose_kernel')][0]
f_fetch = plane / (F[copy] * 1024)
f_write = plane / (Wr[copy] * 1024)
alg = sum(w['algorithmic_bytes_step_launch'] for w in work) / len(work)
res = {
    "grid": 256, "chains": 1024, "steps_per_launch": 32,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over scripts/pmc_run.py; KiB units; "
              "read side multiplied by the factor measured on gsm::stream_copy_kernel (known 512 MiB)",
    "calibration": {"kernel": copy, "known_bytes": plane, "FETCH_SIZE_KiB": F[copy], "WRITE_SIZE_KiB": Wr[copy],
                    "fetch_factor": f_fetch, "write_factor": f_write},
    "step_kernel": step,
    "step_kernel_FETCH_SIZE_KiB": F[step], "step_kernel_WRITE_SIZE_KiB": Wr[step],
    "step_kernel_hbm_bytes_per_launch": (F[step] * f_fetch + Wr[step] * f_write) * 1024,
    "step_kernel_algorithmic_bytes_per_launch": alg,
    "propose_kernel_hbm_bytes_per_launch": (F[prop] * f_fetch + Wr[prop] * f_write) * 1024,
    "note": "the read side counts L2 -> fabric requests, Infinity-Cache hits included: re-fetches of the shared static "
            "fields evicted from the 4 MiB L2 by the per-chain streams show up here although they rarely reach HBM",
}
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
