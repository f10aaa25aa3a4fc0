"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz (build container only).

Runs the imported upstream reference (oracle/ref_loader.py) and the clean-room restatement
(oracle/mcmc_oracle.py) on identical seeded synthetic inputs, asserts that every output is
bit-identical, then writes the golden vectors.  The vectors hold inputs' parameters and
expected outputs only -- no reference source text.  Re-run with:

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_fixtures.py

Fixture set: F1 standard 64x64 chain, F2 variant chain (RF block type,
whole-map updates, anisotropic Gaussian model, nugget), F3 per-step draws + first fields,
F4 edge masks + conditioning weight, F5 residual stencil with NaNs, F6 covariance assembly +
Cholesky draw, F7 two-segment wrapper run with RNG-state JSON, F8 256x256 anchors.
"""
import contextlib
import hashlib
import io
import json
import os
import sys
import tempfile
from copy import deepcopy
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import mcmc_oracle as orc  # noqa: E402
import ref_loader  # noqa: E402

GOLD = HERE.parent / "tests" / "golden"


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or not np.array_equal(a, b, equal_nan=True):
        raise AssertionError(f"oracle differs from reference: {what}")


def quiet():
    return contextlib.redirect_stdout(io.StringIO())


@contextlib.contextmanager
def warnings_off():
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield


def build_reference_chain(M, prob, cfg_kw, rfp: orc.RFParams, block_min, block_max, sigma=5.0):
    """Template chain + RandField through the reference's own public API."""
    with quiet():
        ch = M.chain_crf(prob["xx"], prob["yy"], prob["bed"], prob["surf"], prob["velx"], prob["vely"],
                         prob["dhdt"], prob["smb"], prob["cond_bed"], prob["data_mask"],
                         prob["grounded_ice_mask"], prob["resolution"])
        if cfg_kw["update_in_region"]:
            ch.set_update_region(True, prob["region_mask"])
        else:
            ch.set_update_region(False)
        ch.set_loss_type(sigma_mc=sigma, massConvInRegion=True)
        rf = M.RandField(rfp.range_min_x, rfp.range_max_x, rfp.range_min_y, rfp.range_max_y,
                         rfp.scale_min, rfp.scale_max, rfp.nugget_max, rfp.model_name, rfp.isotropic,
                         smoothness=rfp.smoothness)
        rf.set_block_sizes(block_min, block_max, block_min, block_max)
        rf.set_weight_param(2, 0, 6, 1, 49900.0, prob["resolution"])
        rf.set_generation_method(True)
        ch.set_crf_data_weight(rf)
        ch.set_update_type(cfg_kw["block_type"])
    return ch, rf


def rehydrate(M, ch, rf, seed, initial_bed):
    cp = deepcopy(ch.__dict__)
    cp["rng_seed"] = seed
    cp["initial_bed"] = initial_bed
    rp = deepcopy(rf.__dict__)
    rp["rng_seed"] = seed
    with quiet():
        c = M.init_lsc_chain_by_instance(cp)
        r = M.initiate_RF_by_instance(rp)
    return c, r, cp, rp


def run_pair(M, H, n_iter, chain_index, rfp, block_type, update_in_region, tag):
    prob, cfg, pairs, masks, _ = orc.standard_setup(H, block_type=block_type,
                                                   update_in_region=update_in_region, rf_params=rfp)
    bmin, bmax = (8, 16) if H < 128 else (50, 80)
    ch, rf = build_reference_chain(M, prob, dict(update_in_region=update_in_region, block_type=block_type),
                                   rfp, bmin, bmax)
    # setup-time parity
    same(rf.pairs, pairs, f"{tag}: pairs")
    for a, b in zip(rf.edge_masks, masks):
        same(a, b, f"{tag}: edge mask")
    same(ch.crf_data_weight, cfg.crf_data_weight, f"{tag}: crf weight")
    seed = 7 + chain_index
    bed0 = orc.chain_initial_bed(prob, chain_index)
    c, r, _, _ = rehydrate(M, ch, rf, seed, bed0.copy())
    with quiet():
        ref_out = c.run(n_iter, r, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=False)
    orf = orc.OracleRandField(rfp, seed, pairs, masks, prob["resolution"])
    orng = np.random.default_rng(seed=seed)
    out = orc.run_chain(cfg, bed0.copy(), n_iter, orf, orng, record=True)
    names = ["bed", "loss_mc", "loss_data", "loss", "steps", "resampled", "blocks"]
    for n, a, b in zip(names, ref_out, out[:7]):
        same(a, b, f"{tag}: {n}")
    assert c.rng.bit_generator.state == orng.bit_generator.state, f"{tag}: chain rng state"
    assert r.rng.bit_generator.state == orf.rng.bit_generator.state, f"{tag}: rf rng state"
    return prob, cfg, pairs, masks, out


def sgs_problem(H=32, res=500.0):
    """Synthetic inputs of the small-scale-chain fixture: the standard problem on a small grid, conditioning data on
    every 4th row and every 8th column (so that octant searches find neighbours everywhere), a smooth trend."""
    prob = orc.synthetic_problem(H, res=res)
    data_mask = np.zeros((H, H), dtype=bool)
    data_mask[::4, :] = True
    data_mask[:, ::8] = True
    prob["data_mask"] = data_mask
    prob["cond_bed"] = np.where(data_mask, prob["bed"], np.nan)
    region = np.zeros((H, H), dtype=int)
    region[H // 8: 7 * H // 8, H // 8: 7 * H // 8] = 1
    prob["region_mask"] = region
    Lx = H * res
    prob["trend"] = prob["surf"] - 1000.0 - 150.0 * np.cos(4 * np.pi * prob["xx"] / Lx)
    return prob


def make_f10_sgs(M):
    """F10: chain_sgs.run of the reference vs oracle/sgs_oracle.py, two variants:
    (a) exponential variogram, no transform, no trend; (b) Matern variogram, detrended, normal-score transform."""
    import sgs_oracle as so
    from sklearn.preprocessing import QuantileTransformer
    H, n_iter = 32, 14
    prob = sgs_problem(H)
    out = {}
    for tag, vtype, smooth, use_trend, use_nst, seed, sigma in (("a", "Exponential", None, False, False, 11, 60.0),
                                                                ("b", "Matern", 1.5, True, True, 12, 5.0)):
        trend = prob["trend"] if use_trend else None
        base = prob["bed"] - trend if use_trend else prob["bed"]
        nst = None
        if use_nst:
            data = (prob["cond_bed"] - trend if use_trend else prob["cond_bed"])[prob["data_mask"]].reshape(-1, 1)
            nst = QuantileTransformer(n_quantiles=200, output_distribution="normal", random_state=152).fit(data)
        sill = float(np.var((nst.transform(base.reshape(-1, 1)) if use_nst else base.reshape(-1, 1))))
        rng_range, nug = 6000.0, 0.0
        with quiet():
            ch = M.chain_sgs(prob["xx"], prob["yy"], prob["bed"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"],
                             prob["smb"], prob["cond_bed"], prob["data_mask"], np.ones((H, H), dtype=int), prob["resolution"])
            ch.set_update_region(True, prob["region_mask"])
            ch.set_loss_type(sigma_mc=sigma, massConvInRegion=True)
            ch.set_normal_transformation(nst, do_transform=use_nst)
            ch.set_trend(trend, detrend_map=use_trend)
            ch.set_variogram(vtype, rng_range, sill, nug, isotropic=True, vario_smoothness=smooth)
            ch.set_sgs_param(16, 4000.0)
            ch.set_block_sizes(3, 8, 3, 8)
            ch.set_random_generator(rng_seed=seed)
            with warnings_off():
                ref = ch.run(n_iter, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=False)
        ref_state = ch.rng.bit_generator.state
        cfg = so.SgsConfig(prob["xx"], prob["yy"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                           prob["cond_bed"], prob["data_mask"], np.ones((H, H), dtype=int), prob["region_mask"],
                           prob["resolution"], sigma, [0, nug, rng_range, rng_range, sill, vtype, smooth], [16, 4000.0, False, 0],
                           3, 8, 3, 8, trend=trend, nst_trans=nst)
        rng = np.random.default_rng(seed=seed)
        trace = []
        with warnings_off():
            mine = so.run_chain_sgs(cfg, prob["bed"], n_iter, rng, trace=trace)
        for k, name in enumerate(("bed", "loss_mc", "loss_data", "loss", "steps", "resampled", "blocks")):
            same(ref[k], mine[k], f"F10{tag} {name}")
        assert rng.bit_generator.state == ref_state, f"F10{tag}: final RNG state differs"
        tr = np.array(trace)
        out.update({f"{tag}_bed": mine[0], f"{tag}_loss": mine[3], f"{tag}_steps": mine[4], f"{tag}_resampled": mine[5],
                    f"{tag}_blocks": mine[6], f"{tag}_sill": sill, f"{tag}_seed": seed, f"{tag}_sigma_mc": sigma,
                    f"{tag}_trace_head": tr[:40], f"{tag}_trace_sha": sha(tr), f"{tag}_n_sim": len(trace),
                    f"{tag}_rng_state": json.dumps(ref_state)})
        print(f"F10{tag}: oracle == reference over {n_iter} iterations ({len(trace)} simulated cells, "
              f"accept {mine[4].mean():.2f})")
    np.savez_compressed(GOLD / "f10_sgs_chain32.npz", H=H, n_iter=n_iter, range=6000.0, radius=4000.0, num_points=16,
                        **out)


def main():
    M, T, G, C = ref_loader.load_reference()
    GOLD.mkdir(parents=True, exist_ok=True)

    # ---------------- F1 + F3 + F4: standard 64x64 chain --------------------------------
    rfp = orc.standard_rf_params()
    prob, cfg, pairs, masks, out = run_pair(M, 64, 300, 0, rfp, "CRF_weight", True, "F1")
    tr = out[7]
    np.savez_compressed(
        GOLD / "f1_chain64_standard.npz",
        H=64, n_iter=300, seed=7, chain_index=0,
        bed=out[0], loss=out[3], steps=out[4], resampled=out[5], blocks=out[6],
        size_idx=np.array(tr.size_idx), centre=np.array(tr.centre), u=np.array(tr.u),
        rf_scalars=np.array(tr.rf_scalars), tries=np.array(tr.tries))
    nf = 5
    np.savez_compressed(
        GOLD / "f3_fields64.npz",
        **{f"field{i}": tr.fields[i] for i in range(nf)},
        field_sha=np.array([sha(f) for f in tr.fields]))
    np.savez_compressed(
        GOLD / "f4_setup64.npz", pairs=pairs, crf_weight=cfg.crf_data_weight,
        **{f"mask{i}": m for i, m in enumerate(masks)})

    # a second standard chain (perturbed initial bed, other seed)
    _, _, _, _, out_b = run_pair(M, 64, 200, 3, rfp, "CRF_weight", True, "F1b")
    np.savez_compressed(GOLD / "f1b_chain64_seed10.npz", H=64, n_iter=200, seed=10, chain_index=3,
                        bed=out_b[0], loss=out_b[3], steps=out_b[4], resampled=out_b[5], blocks=out_b[6])

    # ---------------- F2: variant chain ---------------------------------------------------
    rfp2 = orc.RFParams(8e3, 30e3, 12e3, 40e3, 30, 90, 4.0, "Gaussian", False, None)
    _, _, _, _, out2 = run_pair(M, 64, 300, 1, rfp2, "RF", False, "F2")
    np.savez_compressed(GOLD / "f2_chain64_variant.npz", H=64, n_iter=300, seed=8, chain_index=1,
                        bed=out2[0], loss=out2[3], steps=out2[4], resampled=out2[5], blocks=out2[6],
                        rf_params=np.array([8e3, 30e3, 12e3, 40e3, 30, 90, 4.0]))
    rfp2e = orc.RFParams(10e3, 50e3, 10e3, 50e3, 50, 150, 0.0, "Exponential", True, None)
    _, _, _, _, out2e = run_pair(M, 64, 150, 2, rfp2e, "CRF_weight", True, "F2e")
    np.savez_compressed(GOLD / "f2e_chain64_exponential.npz", H=64, n_iter=150, seed=9, chain_index=2,
                        bed=out2e[0], loss=out2e[3], steps=out2e[4], resampled=out2e[5], blocks=out2e[6])

    # ---------------- F5: residual stencil with NaNs -------------------------------------
    g = np.random.default_rng(55)
    shp = (37, 41)
    arrs = {k: g.normal(0, 1, shp) for k in ("velx", "vely", "dhdt", "smb")}
    surf5 = 1500 + g.normal(0, 30, shp)
    bed5 = 400 + g.normal(0, 30, shp)
    bed5[5, 7] = np.nan
    arrs["velx"][20, 3] = np.nan
    ref5 = T.get_mass_conservation_residual(bed5, surf5, arrs["velx"], arrs["vely"], arrs["dhdt"], arrs["smb"], 437.5)
    mine5 = orc.mc_residual(bed5, surf5, arrs["velx"], arrs["vely"], arrs["dhdt"], arrs["smb"], 437.5)
    same(ref5, mine5, "F5 residual")
    # thin windows (2 rows / 2 cols) exercise the one-sided edges only
    thin = T.get_mass_conservation_residual(bed5[:2, :2], surf5[:2, :2], arrs["velx"][:2, :2], arrs["vely"][:2, :2],
                                            arrs["dhdt"][:2, :2], arrs["smb"][:2, :2], 437.5)
    same(thin, orc.mc_residual(bed5[:2, :2], surf5[:2, :2], arrs["velx"][:2, :2], arrs["vely"][:2, :2],
                               arrs["dhdt"][:2, :2], arrs["smb"][:2, :2], 437.5), "F5 thin")
    np.savez_compressed(GOLD / "f5_residual.npz", bed=bed5, surf=surf5, resolution=437.5, residual=ref5,
                        residual_thin=thin, **arrs)

    # ---------------- F6: covariance assembly + Cholesky draw ----------------------------
    jj, ii = np.meshgrid(np.arange(5), np.arange(6))
    coord = np.stack([jj.ravel() * 500.0, ii.ravel() * 500.0], axis=1)
    f6 = {"coord": coord}
    z = np.random.default_rng(66).normal(size=coord.shape[0])
    f6["z"] = z
    for vt, extra in (("Exponential", {}), ("Gaussian", {}), ("Spherical", {}), ("Matern", {"s": 0.9125})):
        vario = dict(azimuth=30.0, nugget=0.0, major_range=4000.0, minor_range=2500.0, sill=1.0, vtype=vt, **extra)
        R = C._krige.make_rotation_matrix(vario["azimuth"], vario["major_range"], vario["minor_range"])
        sig_ref = C._krige.make_sigma(coord, R, vario)
        sig = orc.cov_matrix(coord, vario)
        same(sig_ref, sig, f"F6 sigma {vt}")
        f6[f"sigma_{vt.lower()}"] = sig_ref
        if vt != "Spherical":
            L = np.linalg.cholesky(sig_ref + 1e-10 * np.eye(sig_ref.shape[0]))
            f6[f"draw_{vt.lower()}"] = L @ z
    np.savez_compressed(GOLD / "f6_covariance.npz", **f6)

    # ---------------- F7: two-segment wrapper run (checkpoint files) ----------------------
    import types
    sys.modules.setdefault("config", types.ModuleType("config"))
    import largeScaleChain_multiprocessing as drv  # reference CPU driver; wrapper identical to the GPU one
    prob, cfg, pairs, masks, _ = orc.standard_setup(64)
    ch, rf = build_reference_chain(M, prob, dict(update_in_region=True, block_type="CRF_weight"), rfp, 8, 16)
    seed = 123456789
    with tempfile.TemporaryDirectory() as td:
        outdir = Path(td) / "LargeScaleChain"
        (outdir / str(seed)[:6]).mkdir(parents=True)
        seg = []
        for n_it in (1000, 1000):
            cp = deepcopy(ch.__dict__); cp["rng_seed"] = seed; cp["initial_bed"] = prob["bed"].copy()
            rp = deepcopy(rf.__dict__); rp["rng_seed"] = seed
            runp = dict(n_iter=n_it, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=False,
                        chain_id=0, tqdm_position=1, seed=seed, output_path=str(outdir))
            with quiet():
                seg.append(drv.lsc_run_wrapper(cp, rp, runp))
        folder = outdir / str(seed)[:6]
        files = sorted(p.name for p in folder.iterdir())
        bed2k = np.load(folder / "bed_2k.npy")
        with np.load(folder / "results_2k.npz") as r:
            res = {k: r[k] for k in r.files}
        st_rf = json.load(open(folder / "RNGState_RandField.txt"))
        st_ch = json.load(open(folder / "RNGState_chain.txt"))
        cur = int(np.loadtxt(folder / "current_iter.txt"))
    np.savez_compressed(GOLD / "f7_wrapper_two_segments.npz", seed=seed, files=np.array(files), bed_2k=bed2k,
                        current_iter=cur, rng_state_randfield=json.dumps(st_rf), rng_state_chain=json.dumps(st_ch),
                        **{f"res_{k}": v for k, v in res.items()})

    # ---------------- F8: 256x256 anchors --------------------------------------------------
    _, _, _, _, out8 = run_pair(M, 256, 120, 0, rfp, "CRF_weight", True, "F8")
    np.savez_compressed(GOLD / "f8_chain256_anchor.npz", H=256, n_iter=120, seed=7, chain_index=0,
                        loss=out8[3], steps=out8[4], blocks=out8[6], bed_sha=sha(out8[0]),
                        resampled_sha=sha(out8[5]), bed_row128=out8[0][128])
    # ---------------- F9: Topography.get_highvel_boundary -----------------
    g9 = np.random.default_rng(99)
    H9, W9 = 36, 44
    xx9, yy9 = np.meshgrid(np.arange(W9) * 450.0 + 1.2e5, np.arange(H9) * 450.0 - 3.0e4)
    velx9 = 120.0 * np.exp(-((yy9 - yy9.mean()) / 4000.0) ** 2) + g9.normal(0, 5, (H9, W9))
    vely9 = g9.normal(0, 8, (H9, W9))
    grounded9 = np.ones((H9, W9), dtype=bool); grounded9[:, :6] = False
    ocean9 = ~grounded9
    with warnings_off():
        hv = T.get_highvel_boundary(velx9, vely9, 60.0, grounded9, ocean9, 2500.0, xx9, yy9, smooth_mode=5)
    np.savez_compressed(GOLD / "f9_highvel_boundary.npz", velx=velx9, vely=vely9, grounded=grounded9, ocean=ocean9,
                        xx=xx9, yy=yy9, threshold=60.0, distance_max=2500.0, smooth_mode=5, mask_final=hv)

    make_f10_sgs(M)

    print("fixtures written to", GOLD)
    for p in sorted(GOLD.iterdir()):
        print(f"  {p.name:40s} {p.stat().st_size:9d} B")


if __name__ == "__main__":
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    main()
