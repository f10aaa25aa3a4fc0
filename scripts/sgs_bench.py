"""Small-scale chain: device wall time (all chains of a call in one handle) beside the oracle's CPU loop on this host.
    python scripts/sgs_bench.py [--grid 64] [--chains 4] [--iters 1000] [--philox] [--light] [--no-transform]
Default = the reference driver's own configuration (synthetic.sgs_template: 48 neighbours within 30 km, blocks 5-20, Matern,
QuantileTransformer(1000), trend); --light = the small configuration of rounds 1-2 (16 neighbours within 4 km, blocks 3-8)."""
import argparse, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import numpy as np

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument('--grid', type=int, default=64); ap.add_argument('--chains', type=int, default=4)
    ap.add_argument('--iters', type=int, default=1000); ap.add_argument('--cpu-iters', type=int, default=0)
    ap.add_argument('--philox', action='store_true', help='Philox mode: the draws are made on the device')
    ap.add_argument('--light', action='store_true'); ap.add_argument('--no-transform', action='store_true')
    ap.add_argument('--cells', action='store_true', help='also print the number of simulated cells (block cells without conditioning data)')
    ap.add_argument('--pcg64', action='store_true', help="the chains' own NumPy generator streams advanced on the device (replay mode's chain, no host draws)")
    a = ap.parse_args()
    from mcmc_gpu_amd import sgs, synthetic
    H = a.grid
    prob, ch = synthetic.sgs_template(H, transform=not a.no_transform, light=a.light)
    beds = [prob["bed"] + np.random.default_rng(40 + i).normal(0, 3, prob["bed"].shape) for i in range(a.chains)]
    rngs = [np.random.default_rng(900 + i) for i in range(a.chains)]
    sgs.run_many_sgs(ch, beds[:1], [np.random.default_rng(1)], 20)          # warm-up (library load, first launches)
    t0 = time.time()
    out, _ = sgs.run_many_sgs(ch, beds, rngs, a.iters, philox_seeds=[7000 + i for i in range(a.chains)] if a.philox else None, pcg64=a.pcg64)
    t_dev = time.time() - t0
    msg = (f"small-scale chain {H}x{H} ({'light' if a.light else 'driver'} config, transform {not a.no_transform}, "
           f"{'philox' if a.philox else 'pcg64' if a.pcg64 else 'replay'}), {a.chains} chains x {a.iters} iterations on the device: {t_dev:.2f} s = "
           f"{a.chains * a.iters / t_dev:.0f} chain-iterations/s ({a.iters / t_dev:.0f} it/s per chain, accept {np.mean([o[4].mean() for o in out]):.3f})")
    if a.cells:
        sys.path.insert(0, '.')
        import bench
        is_data = ~np.isnan(np.asarray(prob["cond_bed"]))
        msg += f"; simulated cells of all chains and iterations: {bench._simulated_cells(out, is_data)}"
    if a.cpu_iters:
        import sgs_oracle as so
        v = ch.vario_param
        cfg = so.SgsConfig(prob["xx"], prob["yy"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"], prob["cond_bed"],
                           prob["data_mask"], np.ones((H, H), dtype=int), prob["region_mask"], prob["resolution"], ch.sigma_mc,
                           list(v), list(ch.sgs_param), ch.block_min_x, ch.block_max_x, ch.block_min_y, ch.block_max_y,
                           trend=ch.trend if ch.detrend_map else None, nst_trans=ch.nst_trans if ch.do_transform else None)
        t0 = time.time()
        so.STABLE_TIES = True
        ref = so.run_chain_sgs(cfg, beds[0], a.cpu_iters, np.random.default_rng(900))
        t_cpu = time.time() - t0
        same = a.philox or (np.array_equal(ref[4], out[0][4][:a.cpu_iters]) and np.allclose(ref[3], out[0][3][:a.cpu_iters], rtol=1e-8))
        msg += f"; oracle (reference's NumPy loop) on one host core: {a.cpu_iters / t_cpu:.2f} it/s; first {a.cpu_iters} iterations of chain 0 agree with it: {same}"
    print(msg)
