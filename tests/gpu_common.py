"""Helpers shared by the -m gpu tests: build an engine for the oracle's standard setups."""
import numpy as np

import mcmc_oracle as orc


def make_engine(H, n_chains, block_type="CRF_weight", update_in_region=True, rf_params=None, W=None, state_dtype="f64"):
    from mcmc_gpu_amd.engine import GsmEngine
    prob, cfg, pairs, masks, rfp = orc.standard_setup(H, W, block_type=block_type,
                                                      update_in_region=update_in_region, rf_params=rf_params)
    Hh, Ww = prob["bed"].shape
    eng = GsmEngine(Hh, Ww, n_chains, state_dtype=state_dtype)
    upd = cfg.region_mask if update_in_region else cfg.grounded_ice_mask
    eng.set_static(cfg.surf, cfg.velx, cfg.vely, cfg.dhdt, cfg.smb,
                   cfg.crf_data_weight if block_type == "CRF_weight" else None,
                   upd, cfg.mc_region_mask, cfg.resolution, cfg.sigma_mc)
    eng.set_blocks(pairs, masks)
    eng.set_centres(cfg.region_mask)
    return eng, prob, cfg, pairs, masks, rfp


def oracle_chains(prob, cfg, pairs, masks, rfp, n_chains, n_iter, seed0=7, state_f32=False):
    cfg.state_f32 = state_f32
    outs = []
    for c in range(n_chains):
        rf = orc.OracleRandField(rfp, seed0 + c, pairs, masks, prob["resolution"])
        rng = np.random.default_rng(seed=seed0 + c)
        outs.append(orc.run_chain(cfg, orc.chain_initial_bed(prob, c), n_iter, rf, rng, record=True))
    return outs


def replay_inputs(eng, outs):
    n_chains, n_steps = len(outs), len(outs[0][7].u)
    size_idx = np.array([o[7].size_idx for o in outs])
    centre = np.array([o[7].centre for o in outs])
    u = np.array([o[7].u for o in outs])
    fields = eng.pack_fields([o[7].fields for o in outs])
    return size_idx, centre, u, fields
