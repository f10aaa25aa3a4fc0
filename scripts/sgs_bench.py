"""Small-scale chain: device wall time (all chains of a call in one handle) beside the oracle's CPU loop on this host.
    python scripts/sgs_bench.py [--grid 64] [--chains 4] [--iters 1000]"""
import argparse, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import numpy as np
import sgs_common as sc
import sgs_oracle as so
from mcmc_gpu_amd import sgs

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument('--grid', type=int, default=64); ap.add_argument('--chains', type=int, default=4)
    ap.add_argument('--iters', type=int, default=1000); ap.add_argument('--cpu-iters', type=int, default=30)
    ap.add_argument('--philox', action='store_true', help='Philox mode: the draws are made on the device')
    ap.add_argument('--transform', action='store_true', help="attach scikit-learn's QuantileTransformer(n_quantiles=1000, normal), as the reference's drivers do")
    a = ap.parse_args()
    H = a.grid
    prob = sc.problem(H)
    ch = sgs.chain_sgs_gpu(prob["xx"], prob["yy"], prob["bed"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                           prob["cond_bed"], prob["data_mask"], np.ones((H, H), dtype=int), prob["resolution"])
    ch.set_update_region(True, prob["region_mask"]); ch.set_loss_type(sigma_mc=60.0, massConvInRegion=True)
    nst = None
    if a.transform:
        from sklearn.preprocessing import QuantileTransformer
        nst = QuantileTransformer(n_quantiles=1000, output_distribution='normal', random_state=0, subsample=None).fit(prob['cond_bed'][prob['data_mask']].reshape(-1, 1))
    ch.set_normal_transformation(nst, do_transform=a.transform); ch.set_trend(None, detrend_map=False)
    sill = float(np.var(prob["bed"]))
    ch.set_variogram("Exponential", 6000.0, sill, 0.0, isotropic=True); ch.set_sgs_param(16, 4000.0); ch.set_block_sizes(3, 8, 3, 8)
    beds = [prob["bed"] + np.random.default_rng(40 + i).normal(0, 3, prob["bed"].shape) for i in range(a.chains)]
    rngs = [np.random.default_rng(900 + i) for i in range(a.chains)]
    sgs.run_many_sgs(ch, beds[:1], [np.random.default_rng(1)], 20)          # warm-up (library load, first launches)
    t0 = time.time()
    out, _ = sgs.run_many_sgs(ch, beds, rngs, a.iters, philox_seeds=[7000 + i for i in range(a.chains)] if a.philox else None)
    t_dev = time.time() - t0
    cfg = so.SgsConfig(prob["xx"], prob["yy"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"], prob["cond_bed"],
                       prob["data_mask"], np.ones((H, H), dtype=int), prob["region_mask"], prob["resolution"], 60.0,
                       [0, 0.0, 6000.0, 6000.0, sill, "Exponential", None], [16, 4000.0, False, 0], 3, 8, 3, 8, nst_trans=nst)
    t0 = time.time()
    ref = so.run_chain_sgs(cfg, beds[0], a.cpu_iters, np.random.default_rng(900))
    t_cpu = time.time() - t0
    same = a.philox or (np.array_equal(ref[4], out[0][4][:a.cpu_iters]) and np.allclose(ref[3], out[0][3][:a.cpu_iters], rtol=1e-9))
    print(f"small-scale chain {H}x{H}, {a.chains} chains x {a.iters} iterations on the device: {t_dev:.2f} s = "
          f"{a.chains * a.iters / t_dev:.0f} chain-iterations/s ({a.iters / t_dev:.0f} it/s per chain, accept {np.mean([o[4].mean() for o in out]):.2f}); "
          f"oracle (reference's NumPy loop) on one host core: {a.cpu_iters / t_cpu:.1f} it/s; first {a.cpu_iters} iterations of chain 0 "
          f"agree with it: {same}")
