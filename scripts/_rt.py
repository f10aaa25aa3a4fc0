import sys, time
sys.path.insert(0,'.')
import numpy as np
from mcmc_gpu_amd import MCMC_gpu, synthetic
if __name__=="__main__":
    import os
    print("cpus", os.cpu_count(), len(os.sched_getaffinity(0)))
    prob, ch, rf = synthetic.template(256)
    n=64
    st=[np.random.default_rng(seed=s).bit_generator.state for s in range(n)]
    t0=time.time()
    out=MCMC_gpu.run_many_replay(ch, rf, synthetic.initial_beds(prob,n), st, st, 1000, progress=True)
    print("total", time.time()-t0)
