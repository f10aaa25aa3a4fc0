"""Workload for rocprofv3 --pmc passes: calibration kernels with known byte counts (residual_kernel / init_loss_kernel:
8 B/lane coalesced reads and writes of 512 MiB each) followed by one proposal launch and one step launch of the
headline geometry (256x256, 1024 chains, 8 steps).  Prints the algorithmic bytes of the step launch."""
import ctypes as C, sys, json
sys.path.insert(0, '.')
import numpy as np, torch
from mcmc_gpu_amd import synthetic
from mcmc_gpu_amd.engine import _ptr
import bench

chains, steps = 1024, 32
prob, ch, rf = synthetic.template(256)
eng = ch._make_engine(rf, chains, 0)
eng.set_state(synthetic.initial_beds(prob, chains))            # init_loss_kernel: read beds, write energy
r = eng.residual(eng.beds); del r                              # residual_kernel: read beds, write residual
dst = torch.empty_like(eng.energy)
eng._check(eng.lib.gsm_debug_stream_copy(_ptr(eng.energy), _ptr(dst), eng.energy.numel(), eng._stream()))   # calibration: 512 MiB in, 512 MiB out
del dst
p = eng.rf_struct(rf)
seeds = eng._seeds(list(range(7, 7 + chains)))
n = chains * steps
si = torch.empty(n, dtype=torch.int32, device='cuda'); ce = torch.empty(2 * n, dtype=torch.int32, device='cuda')
u = torch.empty(n, dtype=torch.float64, device='cuda')
fl = torch.empty((n, eng.field_stride), dtype=torch.float64, device='cuda')
loss = torch.empty(n, dtype=torch.float64, device='cuda'); acc = torch.empty(n, dtype=torch.uint8, device='cuda')
st = eng._stream()
for rep in range(3):
    eng._check(eng.lib.gsm_propose_philox(eng.h, steps, rep * steps, _ptr(seeds), C.byref(p), _ptr(si), _ptr(ce), _ptr(u), _ptr(fl), eng.field_stride, None, st))
    eng._check(eng.lib.gsm_run_replay(eng.h, steps, _ptr(eng.beds), _ptr(eng.energy), _ptr(eng.resampled), _ptr(eng.loss_sum), _ptr(si), _ptr(ce), _ptr(u), _ptr(fl), eng.field_stride, _ptr(loss), _ptr(acc), st))
    torch.cuda.synchronize()
    blocks = np.concatenate([ce.view(chains, steps, 2).cpu().numpy(), eng.bh[si.cpu().numpy()].reshape(chains, steps, 1), eng.bw[si.cpu().numpy()].reshape(chains, steps, 1)], axis=2)
    print(json.dumps({"rep": rep, "algorithmic_bytes_step_launch": bench.algorithmic_bytes(blocks, acc.view(chains, steps).cpu().numpy(), 256, 256),
                      "accept": float(acc.float().mean()), "plane_bytes": chains * 256 * 256 * 8}))
