"""Workload for rocprofv3 --pmc passes over the fused chain kernel: a calibration stream copy of known size (512 MiB in,
512 MiB out, 8 B per lane) followed by three gsm_run_philox calls of the headline geometry (256x256, 1024 chains,
`steps` Metropolis steps each = one chain_fused_kernel launch).  Prints the algorithmic bytes of each launch."""
import json, sys
sys.path.insert(0, '.')
import numpy as np, torch
from mcmc_gpu_amd import synthetic
from mcmc_gpu_amd.engine import _ptr
import bench

chains, steps = 1024, int(sys.argv[1]) if len(sys.argv) > 1 else 64
prob, ch, rf = synthetic.template(256)
eng = ch._make_engine(rf, chains, 0)
eng.set_state(synthetic.initial_beds(prob, chains))
dst = torch.empty_like(eng.energy)
eng._check(eng.lib.gsm_debug_stream_copy(_ptr(eng.energy), _ptr(dst), eng.energy.numel(), eng._stream()))
del dst
p = eng.rf_struct(rf)
seeds = list(range(7, 7 + chains))
for rep in range(3):
    loss, acc, blk = eng.run_philox(steps, rep * steps, seeds, p, batch=steps)
    print(json.dumps({"rep": rep, "steps": steps, "algorithmic_bytes_launch": bench.algorithmic_bytes(blk, acc, 256, 256),
                      "accept": float(np.mean(acc)), "plane_bytes": chains * 256 * 256 * 8}))
