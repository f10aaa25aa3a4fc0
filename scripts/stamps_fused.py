"""Diagnostic: where a step of the fused chain kernel spends its cycles (stamped build, GPU box):
    python scripts/stamps_fused.py [--chains 1024] [--steps 64]"""
import argparse, ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ.setdefault("GSM_STAMPS", "1")
import numpy as np, torch
from mcmc_gpu_amd import _lib
_lib.build(force=True)
from mcmc_gpu_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument('--chains', type=int, default=1024); ap.add_argument('--grid', type=int, default=256)
ap.add_argument('--steps', type=int, default=64)
a = ap.parse_args()
prob, ch, rf = synthetic.template(a.grid)
eng = ch._make_engine(rf, a.chains, 0)
eng.set_state(synthetic.initial_beds(prob, a.chains))
p = eng.rf_struct(rf)
seeds = list(range(7, 7 + a.chains))
eng.enable_timing(True)
for r in range(2):
    loss, acc, blk = eng.run_philox(a.steps, r * a.steps, seeds, p, batch=a.steps)
tm = eng.last_timing()
out = np.zeros((a.chains, 16), dtype=np.uint64)
eng.lib.gsm_debug_stamps_fused.argtypes = [C.c_void_p, C.c_int32]
rc = eng.lib.gsm_debug_stamps_fused(out.ctypes.data, a.chains)
assert rc == 0, rc
names = {15: "loop top", 10: "P coefficients (Philox, Box-Muller, amplitude) + table DMA issue", 0: "window, fence, P0 issue state loads",
         2: "barrier (slowest wave's coefficients, table DMA)", 1: "stage 1 MFMA", 13: "stage-1 barrier",
         11: "T^T write + barrier + stage 2 MFMA + mask loads", 14: "standardise (reduction + barrier)", 12: "emit -> LDS",
         3: "emit barrier", 4: "A statics+flux", 5: "A barrier", 6: "D stencil", 7: "R wave-reduce", 8: "R barrier",
         9: "decide + E commit"}
order = [15, 10, 0, 2, 1, 13, 11, 14, 12, 3, 4, 5, 6, 7, 8, 9]
per = out.astype(np.float64).mean(axis=0) / a.steps
print(f"fused launch {tm['step_ms']:.3f} ms for {a.steps} steps x {a.chains} chains; accept {np.mean(acc):.3f}")
for k in order:
    print(f"  {names[k]:72s} {per[k]:9.0f} cycles/step  ({100 * per[k] / per.sum():5.1f} %)")
print(f"  total                              {per.sum():9.0f} cycles/step")
