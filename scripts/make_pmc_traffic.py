"""profiles/pmc_traffic.json from rocprofv3 --pmc passes of scripts/pmc_fused.py (FETCH_SIZE, WRITE_SIZE in separate
passes; optionally a third directory with SQ counters).

Correction per MI355X_MICROARCH.md (HBM section): the counters are in KiB; FETCH_SIZE under-reports wide coalesced
reads on gfx950, so it is calibrated on gsm::stream_copy_kernel (8 bytes per lane, exactly 512 MiB read and 512 MiB
written in the same run); WRITE_SIZE is checked on the same kernel."""
import csv, glob, json, sys
fetch_dir, write_dir, log, out = sys.argv[1:5]
sq_dirs = sys.argv[5:]
def mean_per_kernel(dirs, counter):
    acc = {}
    for d in dirs:
        for f in glob.glob(f'{d}/**/*_counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                if r['Counter_Name'] == counter:
                    acc.setdefault(r['Kernel_Name'].split('(')[0], []).append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in acc.items()}
F, Wr = mean_per_kernel([fetch_dir], 'FETCH_SIZE'), mean_per_kernel([write_dir], 'WRITE_SIZE')
work = [json.loads(l) for l in open(log) if l.startswith('{')]
plane = work[0]['plane_bytes']
copy = [k for k in F if 'stream_copy' in k][0]
fused = [k for k in F if 'chain_fused_kernel' in k or 'chain_strip_kernel' in k][0]      # whichever chain kernel the handle ran
f_fetch = plane / (F[copy] * 1024)
f_write = plane / (Wr[copy] * 1024)
alg = sum(w['algorithmic_bytes_launch'] for w in work) / len(work)
import subprocess
try:
    commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:
    commit = None
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
try:                                   # the GPU box has no .git: the library's source hash identifies the build there
    from mcmc_gpu_amd import _lib
    src_hash = _lib.source_hash()
except Exception:
    src_hash = None
res = {
    "grid": 256, "chains": 1024, "state": "f64", "steps_per_launch": work[0]['steps'],
    "commit": os.environ.get("GSM_COMMIT") or commit or (("src:" + src_hash) if src_hash else None),
    "source_hash": src_hash,
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over scripts/pmc_fused.py; KiB units; "
              "read side multiplied by the factor measured on gsm::stream_copy_kernel (known 512 MiB)",
    "calibration": {"kernel": copy, "known_bytes": plane, "FETCH_SIZE_KiB": F[copy], "WRITE_SIZE_KiB": Wr[copy],
                    "fetch_factor": f_fetch, "write_factor": f_write},
    "kernel": fused,
    "FETCH_SIZE_KiB": F[fused], "WRITE_SIZE_KiB": Wr[fused],
    "hbm_bytes_per_launch": (F[fused] * f_fetch + Wr[fused] * f_write) * 1024,
    "algorithmic_bytes_per_launch": alg,
    "note": "the read side counts L2 -> fabric requests, Infinity-Cache hits included",
}
if sq_dirs:
    sq = {}
    for c in ("SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
              "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU",
              "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        m = mean_per_kernel(sq_dirs, c)
        if fused in m:
            sq[c] = m[fused]
    res["sq_counters_per_launch"] = sq
    if "SQ_WAVE_CYCLES" in sq and "SQ_ACTIVE_INST_VALU" in sq:
        waves_per_simd = 4   # one 1024-thread workgroup or two 512-thread workgroups per CU
        # SQ_WAVE_CYCLES and SQ_ACTIVE_INST_* count quad-cycles per wave; a SIMD executes one VALU instruction at a time
        res["valu_busy_frac_per_simd"] = waves_per_simd * sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]
        res["mfma_busy_frac_per_simd"] = sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (sq["SQ_WAVE_CYCLES"] * 4 / waves_per_simd)
        chain_steps = res["chains"] * res["steps_per_launch"]
        res["valu_wave_instructions_per_chain_step"] = sq.get("SQ_INSTS_VALU", 0.0) / chain_steps
        res["mfma_instructions_per_chain_step"] = sq.get("SQ_INSTS_MFMA", 0.0) / chain_steps
        res["salu_instructions_per_chain_step"] = sq.get("SQ_INSTS_SALU", 0.0) / chain_steps
        if "SQ_WAIT_ANY" in sq:
            res["wait_any_frac_of_wave_cycles"] = sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"]
        if "SQ_LDS_BANK_CONFLICT" in sq and sq.get("SQ_LDS_IDX_ACTIVE"):
            res["lds_bank_conflict_frac"] = sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]
json.dump(res, open(out, 'w'), indent=1)
print(json.dumps(res, indent=1))
