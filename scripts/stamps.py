"""Diagnostic: where a step of the flux step kernel spends its cycles.  Run on the GPU box with a stamped build:
    GSM_STAMPS=1 GSM_FORCE_BUILD=1 python scripts/stamps.py [--chains 1024] [--steps 32]
Prints, per phase, the mean shader cycles per step of wave 0 (s_memtime), over all chains of the last launch."""
import argparse, ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ.setdefault("GSM_STAMPS", "1")
import numpy as np, torch
from mcmc_gpu_amd import _lib
_lib.build(force=True)
from mcmc_gpu_amd import synthetic
from mcmc_gpu_amd.engine import _ptr

ap = argparse.ArgumentParser()
ap.add_argument('--chains', type=int, default=1024); ap.add_argument('--grid', type=int, default=256)
ap.add_argument('--steps', type=int, default=32); ap.add_argument('--reps', type=int, default=3)
a = ap.parse_args()
prob, ch, rf = synthetic.template(a.grid)
eng = ch._make_engine(rf, a.chains, 0)
eng.set_state(synthetic.initial_beds(prob, a.chains))
p = eng.rf_struct(rf)
seeds = eng._seeds(list(range(7, 7 + a.chains)))
n = a.chains * a.steps
si = torch.empty(n, dtype=torch.int32, device='cuda'); ce = torch.empty(2 * n, dtype=torch.int32, device='cuda')
u = torch.empty(n, dtype=torch.float64, device='cuda')
fl = torch.empty((n, eng.field_stride), dtype=torch.float64, device='cuda')
loss = torch.empty(n, dtype=torch.float64, device='cuda'); acc = torch.empty(n, dtype=torch.uint8, device='cuda')
st = eng._stream()
eng.lib.gsm_debug_stamps.argtypes = [C.c_void_p, C.c_int32]
for r in range(a.reps):
    eng._check(eng.lib.gsm_propose_philox(eng.h, a.steps, r * a.steps, _ptr(seeds), C.byref(p), _ptr(si), _ptr(ce), _ptr(u), _ptr(fl), eng.field_stride, None, st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    eng._check(eng.lib.gsm_run_replay(eng.h, a.steps, _ptr(eng.beds), _ptr(eng.energy), _ptr(eng.resampled), _ptr(eng.loss_sum), _ptr(si), _ptr(ce), _ptr(u), _ptr(fl), eng.field_stride, _ptr(loss), _ptr(acc), st))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
out = np.zeros((a.chains, 8), dtype=np.uint64)
rc = eng.lib.gsm_debug_stamps(out.ctypes.data, a.chains)
assert rc == 0, rc
names = ["scalars+window", "A loads+flux", "A barrier", "D stencil", "R wave-reduce", "R barrier", "decide+E commit", "loop top/fence"]
per = out.astype(np.float64).mean(axis=0) / a.steps
print(f"step launch {ms:.3f} ms for {a.steps} steps x {a.chains} chains; accept {acc.float().mean().item():.3f}")
for nm, v in zip(names, per):
    print(f"  {nm:18s} {v:9.0f} cycles/step  ({100 * v / per.sum():5.1f} %)")
print(f"  total              {per.sum():9.0f} cycles/step = {per.sum() / 100:.1f} us at 100 MHz memtime" )
