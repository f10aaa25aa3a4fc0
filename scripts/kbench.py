"""Kernel-level timing on the GPU box: proposal kernel and step kernel in isolation (torch events on the
current stream).  Usage: python scripts/kbench.py [--chains 1024] [--grid 256] [--steps 8] [--reps 10]"""
import argparse, ctypes as C, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from mcmc_gpu_amd import synthetic
from mcmc_gpu_amd.engine import _ptr

ap = argparse.ArgumentParser()
ap.add_argument('--chains', type=int, default=1024); ap.add_argument('--grid', type=int, default=256)
ap.add_argument('--steps', type=int, default=8); ap.add_argument('--reps', type=int, default=10)
ap.add_argument('--generator', default='spectral'); ap.add_argument('--classes', type=int, default=2)
a = ap.parse_args()
prob, ch, rf = synthetic.template(a.grid)
eng = ch._make_engine(rf, a.chains, 0)
eng.set_state(synthetic.initial_beds(prob, a.chains))
if a.generator == 'cholesky':
    from mcmc_gpu_amd import cholesky as chol
    rf.generator = 'cholesky'
    chol.build_factors(eng, rf, n_classes=a.classes)
p = eng.rf_struct(rf)
seeds = eng._seeds(list(range(7, 7 + a.chains)))
n = a.chains * a.steps
si = torch.empty(n, dtype=torch.int32, device='cuda'); ce = torch.empty(2 * n, dtype=torch.int32, device='cuda')
u = torch.empty(n, dtype=torch.float64, device='cuda')
fl = torch.empty((n, eng.field_stride), dtype=torch.float64, device='cuda')
loss = torch.empty(n, dtype=torch.float64, device='cuda'); acc = torch.empty(n, dtype=torch.uint8, device='cuda')
st = eng._stream()
def ev(): return torch.cuda.Event(enable_timing=True)
tp, ts = [], []
for r in range(a.reps):
    e0, e1, e2 = ev(), ev(), ev()
    e0.record()
    eng._check(eng.lib.gsm_propose_philox(eng.h, a.steps, r * a.steps, _ptr(seeds), C.byref(p), _ptr(si), _ptr(ce), _ptr(u), _ptr(fl), eng.field_stride, None, st))
    e1.record()
    eng._check(eng.lib.gsm_run_replay(eng.h, a.steps, _ptr(eng.beds), _ptr(eng.energy), _ptr(eng.resampled), _ptr(eng.loss_sum), _ptr(si), _ptr(ce), _ptr(u), _ptr(fl), eng.field_stride, _ptr(loss), _ptr(acc), st))
    e2.record(); torch.cuda.synchronize()
    tp.append(e0.elapsed_time(e1)); ts.append(e1.elapsed_time(e2))
cs = a.chains * a.steps
print(f"propose: median {np.median(tp):.3f} ms  min {min(tp):.3f}  -> {cs / np.median(tp) / 1e3:.2f} M proposals/s")
print(f"step   : median {np.median(ts):.3f} ms  min {min(ts):.3f}  -> {cs / np.median(ts) / 1e3:.2f} M chain-steps/s   accept {acc.float().mean().item():.3f}")
