#!/usr/bin/env python3
"""Throughput of the large-scale-chain Metropolis hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 256x256 synthetic grid, 1024 chains per GPU, fp64,
Matern(0.9125) spectral proposals, blocks 50-80 cells, sigma_mc = 5, Philox draws generated on the device.
One bench "step" = `--inner` Metropolis steps of every chain = one gsm_run_philox call = one launch of the fused chain
kernel (proposal + Metropolis step per chain-step inside it).  Weak scaling: every rank runs its own 1024 chains
(different seeds), no collective inside the step loop, one all-gather of the per-chain caches and an all-reduce of the
posterior-mean field at the end of the timed region.

Ranks.  Under torchrun (RANK / WORLD_SIZE in the environment) this process is one rank.  Called directly with
--gpus N > 1 it starts its N ranks ITSELF -- N children of this script, one per GPU, rendezvous on 127.0.0.1 -- before
anything touches the GPU in the parent, waits for them and exits non-zero if any child failed (the reference's driver
starts its own workers too, largeScaleChain_multiprocessing_GPU.py:47, :84-85).  Backend "nccl" (= RCCL);
GSM_DIST_BACKEND=gloo rehearses the same code path with the ranks sharing the visible GPUs.

Prints ONE JSON line on rank 0 (see the driver contract): metric chain-steps/s, plus
  roofline      -- algorithmic HBM bytes (SURVEY.md 8d formula, summed exactly from the recorded blocks/accepts)
                   per launch of the dominant kernel / its average duration (HIP events on its own stream)
  cpu_baseline  -- the NumPy oracle (bit-validated restatement of the reference's CPU loop) under the reference's
                   multiprocessing.Pool pattern on this host's cores, bounded sample, rank 0 at N=1 only
  other_configs -- (N=1, default workload only) short measurements of BASELINE configs[3] (512x512, precomputed-Cholesky
                   generator, MFMA roofline), of one GPU's shard of configs[4] (1024x1024, fp32 state) and of configs[0] (the
                   small-scale chain, 64x64 grid x 4 chains) on the device, in the same run.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F64_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak


def pmc_traffic(H, n_chains, state, kernel="chain_fused_kernel"):
    """HBM-side bytes per chain-step of the fused chain kernel from the COMMITTED rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of scripts/pmc_fused.py,
    calibrated on a stream copy of known size, MI355X_MICROARCH.md HBM section), plus the SQ counters of the same
    workload.  Not collected in this run: the returned source string says where the numbers come from.
    (None, None, None) when no measurement matches this config."""
    try:
        d = json.load(open(ROOT / "profiles" / "pmc_traffic.json"))
        if (d["grid"], d["chains"], d.get("state", "f64")) == (H, n_chains, state) and kernel in d.get("kernel", "chain_fused_kernel"):
            sq = {k: d[k] for k in ("valu_busy_frac_per_simd", "mfma_busy_frac_per_simd",
                                    "valu_wave_instructions_per_chain_step", "mfma_instructions_per_chain_step",
                                    "salu_instructions_per_chain_step") if k in d}
            per_step = d["hbm_bytes_per_launch"] / (d["chains"] * d["steps_per_launch"])
            src = f"profiles/pmc_traffic.json @ {d.get('commit', 'round 1')} (rocprofv3 --pmc passes, not collected in this run)"
            return per_step, sq, src
    except Exception:
        pass
    return None, None, None


def algorithmic_bytes(blocks, accept, H, W, state_bytes=8):
    """s*2*B_eff + a*(s*2*B_eff + 8*B_eff) summed over every chain-step (B_eff = clipped window).
    NumPy arrays or torch tensors."""
    if isinstance(blocks, torch.Tensor):
        row, col, bh, bw = (blocks[..., i].long() for i in range(4))
        r0 = (row - bh // 2).clamp(min=0); r1 = (row + bh // 2).clamp(max=H)
        c0 = (col - bw // 2).clamp(min=0); c1 = (col + bw // 2).clamp(max=W)
        beff = (r1 - r0) * (c1 - c0)
        a = accept.long()
        return int((state_bytes * 2 * beff + a * (state_bytes * 2 * beff + 8 * beff)).sum().item())
    row, col, bh, bw = (blocks[..., i].astype(np.int64) for i in range(4))
    r0 = np.maximum(0, row - bh // 2); r1 = np.minimum(H, row + bh // 2)
    c0 = np.maximum(0, col - bw // 2); c1 = np.minimum(W, col + bw // 2)
    beff = (r1 - r0) * (c1 - c0)
    a = accept.astype(np.int64)
    return int((state_bytes * 2 * beff + a * (state_bytes * 2 * beff + 8 * beff)).sum())  # s = 8 (f64) or 4 (f32)


def _cpu_worker(args):
    """One chain on one core with the oracle's reference-faithful loop (test infrastructure, timed as baseline)."""
    sys.path.insert(0, str(ROOT / "oracle"))
    import mcmc_oracle as orc
    cfg, pairs, masks, rfp, res, bed0, seed, n_iter = args
    rf = orc.OracleRandField(rfp, seed, pairs, masks, res)
    rng = np.random.default_rng(seed=seed)
    t0 = time.perf_counter()
    out = orc.run_chain(cfg, bed0, n_iter, rf, rng)
    return n_iter - 1, time.perf_counter() - t0, float(out[4].mean())


def cpu_baseline(H, budget_s=15.0):
    """largeScaleChain_multiprocessing.py pattern: Pool(n_workers), one chain per task,
    n_workers = physical cores - 1 (:464), capped to the CPUs this process may use."""
    import multiprocessing as mp
    sys.path.insert(0, str(ROOT / "oracle"))
    import mcmc_oracle as orc
    try:
        import psutil
        phys = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        phys = os.cpu_count()
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else phys
    n_workers = max(1, min(phys - 1, allowed - 1, 15))
    prob, cfg, pairs, masks, rfp = orc.standard_setup(H)
    # calibrate on one short chain, then size the sample to ~budget_s of wall time
    n_cal = 120
    _, t_cal, _ = _cpu_worker((cfg, pairs, masks, rfp, prob["resolution"], orc.chain_initial_bed(prob, 0), 7, n_cal))
    rate1 = (n_cal - 1) / t_cal
    n_iter = int(max(200, min(20000, rate1 * budget_s)))
    tasks = [(cfg, pairs, masks, rfp, prob["resolution"], orc.chain_initial_bed(prob, i), 7 + i, n_iter)
             for i in range(n_workers)]
    t0 = time.perf_counter()
    with mp.get_context("fork").Pool(n_workers) as pool:
        res = pool.map(_cpu_worker, tasks)
    wall = time.perf_counter() - t0
    steps = sum(r[0] for r in res)
    return {"value": steps / wall, "unit": "chain-steps/s", "cores": n_workers, "kind": "port",
            "sample": f"{n_workers} chains x {n_iter - 1} steps, {H}x{H} grid, blocks 50-80, NumPy oracle of MCMC.py:1247-1360 "
                      f"under multiprocessing.Pool({n_workers}); {wall:.1f} s wall incl. pool start; "
                      f"{steps / sum(r[1] for r in res):.0f} steps/s/core in-loop",
            "accept_rate": float(np.mean([r[2] for r in res])),
            # the port avoids some of the reference's full-grid copies: measured in the build container (same core, 256x256,
            # blocks 50-80): oracle 850 steps/s, imported reference chain_crf.run 672 steps/s
            "port_speed_vs_reference": 1.26,
            "note": "the NumPy port runs ~1.26x faster than the reference's own loop on the same core: divide `value` by 1.26 for an "
                    "estimate of the reference itself"}


# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """Start n ranks of this script (children, one per GPU) and wait.  Runs before anything in this process touches the
    GPU (torch.cuda.device_count() does not initialise it).  Returns the exit code: 0 only if every rank exited 0."""
    backend = os.environ.get("GSM_DIST_BACKEND") or ("nccl" if torch.cuda.device_count() > 0 else "gloo")
    if backend == "nccl" and torch.cuda.device_count() < n:
        print(f"bench.py: --gpus {n} but only {torch.cuda.device_count()} GPU(s) visible; RCCL needs one GPU per rank "
              "(GSM_DIST_BACKEND=gloo rehearses the multi-rank path on fewer)", file=sys.stderr)
        return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GSM_DIST_BACKEND=backend, GSM_BENCH_SELF_LAUNCHED="1")
        env.setdefault("OMP_NUM_THREADS", "1")      # as torchrun does: N ranks must not each start a thread per host core
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *argv], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                for q in alive:          # a dead rank leaves the others waiting in a collective: stop them (exact PIDs)
                    q.terminate()
        time.sleep(0.05)
    return rc


def measure(args, H, n_local, generator, state, inner, batch, steps, warmup, rank, world, dev, classes=2, headline=True):
    """One workload, timed as the driver contract says; returns the result dict (rank 0) or None."""
    from mcmc_gpu_amd import parallel, synthetic
    dev_index = dev.index
    n_total = n_local * world
    prob, ch, rf = synthetic.template(H)
    ch.state_dtype = state
    sbytes = 8 if state == "f64" else 4
    if headline:
        beds0 = synthetic.initial_beds(prob, n_local, first=rank * n_local)   # SURVEY 8d initial beds, chain by chain
    else:
        g = torch.Generator(device=dev); g.manual_seed(1000 + rank)
        beds0 = torch.as_tensor(prob["bed"], device=dev)[None] + 5.0 * torch.randn((n_local, H, H), dtype=torch.float64, device=dev, generator=g)
    seeds = [7 + rank * n_local + i for i in range(n_local)]
    eng = ch._make_engine(rf, n_local, dev_index)
    eng.set_state(beds0)
    del beds0
    eng.enable_timing(True)
    t_setup = time.perf_counter()
    if generator == "cholesky":
        from mcmc_gpu_amd import cholesky as chol
        rf.generator = "cholesky"
        chol.build_factors(eng, rf, n_classes=classes)
    t_setup = time.perf_counter() - t_setup
    p = eng.rf_struct(rf)
    n_timed = steps * inner
    loss = torch.empty((n_local, inner), dtype=torch.float64, device=dev)
    acc = torch.empty((n_local, inner), dtype=torch.uint8, device=dev)
    blk = torch.empty((n_local, inner, 4), dtype=torch.int32, device=dev)
    acc_all = torch.empty((n_local, n_timed), dtype=torch.uint8, device=dev)
    loss_all = torch.empty((n_local, n_timed), dtype=torch.float64, device=dev)
    blk_all = torch.empty((n_local, n_timed, 4), dtype=torch.int32, device=dev) if rank == 0 else None

    def segment_end():
        """Per-chain caches to every rank, posterior-mean field, small results to the host."""
        g_loss = parallel.all_gather_chains(loss_all, n_total)
        g_acc = parallel.all_gather_chains(acc_all, n_total)
        mean_field = parallel.all_reduce_mean_field(eng.beds.sum(dim=0, dtype=torch.float64), n_total)
        if args.gather_beds:
            parallel.all_gather_chains(eng.beds, n_total)
        return g_loss[:, -1].cpu(), float(g_acc.float().mean().item()), mean_field.cpu()

    step0 = 0
    for _ in range(warmup):
        eng.run_philox(inner, step0, seeds, p, batch=batch, out=(loss, acc, blk), to_host=False)
        step0 += inner
    if warmup > 0:
        acc_all.zero_(); loss_all.zero_()
        segment_end()          # loads torch's reduce/copy kernels and RCCL channels outside the timed region
        algorithmic_bytes(blk, acc, H, H, sbytes)
    torch.cuda.synchronize(dev)
    parallel.barrier()
    torch.cuda.synchronize(dev)
    t_step = t_prop = 0.0
    n_step_l = n_prop_l = 0
    t0 = time.perf_counter()
    for k in range(steps):
        eng.run_philox(inner, step0, seeds, p, batch=batch, out=(loss, acc, blk), to_host=False)
        step0 += inner
        sl = slice(k * inner, (k + 1) * inner)
        acc_all[:, sl] = acc; loss_all[:, sl] = loss
        if rank == 0:
            blk_all[:, sl] = blk     # the roofline numerator is summed from these records after the timed region
        tm = eng.last_timing()
        t_step += tm["step_ms"] * tm["step_launches"]; n_step_l += tm["step_launches"]
        t_prop += tm["proposal_ms"] * tm["proposal_launches"]; n_prop_l += tm["proposal_launches"]
    h_loss, h_acc_rate, h_mean = segment_end()
    torch.cuda.synchronize(dev)
    parallel.barrier()
    torch.cuda.synchronize(dev)
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, dev)
    fused = bool(eng.last_run_fused())
    strip = bool(eng.strip_active())          # chain_strip_kernel (two chains per CU) or chain_fused_kernel (one)
    eng.close()
    del eng
    if rank != 0:
        return None

    chain_steps = n_total * n_timed
    bytes_total, flops_total = 0, 0.0
    for k in range(steps):       # device-side sums, one bench step at a time (bounded temporaries)
        sl = slice(k * inner, (k + 1) * inner)
        bytes_total += algorithmic_bytes(blk_all[:, sl], acc_all[:, sl], H, H, sbytes)
        if generator == "cholesky":
            flops_total += float(((blk_all[:, sl, 2].double() * blk_all[:, sl, 3].double()) ** 2).sum().item())
    del blk_all
    bytes_per_launch = bytes_total / max(n_step_l, 1)
    step_ms = t_step / max(n_step_l, 1); prop_ms = t_prop / max(n_prop_l, 1)
    steps_per_launch = n_timed / max(n_step_l, 1)
    if generator == "cholesky":
        # algorithmic flops per chain-step = (bh*bw)^2 (lower-triangular L z)
        flops_per_launch = flops_total / max(n_prop_l, 1)
        ach = flops_per_launch / (prop_ms * 1e-3) / 1e12 if prop_ms > 0 else 0.0
        roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / MFMA_F64_PEAK_TFLOPS, "traffic": None, "kernel": "cz_* proposal pipeline (zgen + gemm)",
                "algorithmic_flops_per_chain_step": flops_total / (n_local * n_timed), "flops_per_launch": flops_per_launch,
                "step_kernel_ms": step_ms, "propose_kernel_ms": prop_ms, "launches": n_prop_l}
    else:
        dom = (("chain_strip_kernel" if strip else "chain_fused_kernel") if fused else
               ("step_kernel" if t_step >= t_prop else "propose_kernel"))
        dom_ms = prop_ms if dom == "propose_kernel" else step_ms
        achieved = bytes_per_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        per_step, sq, src = pmc_traffic(H, n_local, state, dom) if fused else (None, None, None)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": per_step * n_local * steps_per_launch if per_step else None,
                "kernel": dom, "algorithmic_bytes_per_chain_step": bytes_total / (n_local * n_timed),
                "bytes_per_launch": bytes_per_launch, "kernel_ms": dom_ms, "launches": n_step_l}
        if per_step:
            roof["traffic_source"] = src
        if sq:
            # what actually limits the fused kernel: vector-instruction issue (rocprofv3 SQ counters of the same workload)
            roof["issue_limits"] = dict(sq, source=src)
        if not fused:
            roof["step_kernel_ms"] = step_ms; roof["propose_kernel_ms"] = prop_ms
    gen_txt = ("spectral (Matern 0.9125) proposals" if generator == "spectral" else
               f"precomputed-Cholesky (Matern 0.9125, {classes} range classes) proposals")
    tag = {("spectral", "f64", 256): "BASELINE configs[1]" if world == 1 else "BASELINE configs[2] at 8 GPUs: 1024 chains/GPU",
           ("cholesky", "f64", 512): "BASELINE configs[3]",
           ("spectral", "f32", 1024): "BASELINE configs[4], one GPU's shard"}.get((generator, state, H), "")
    return {
        "metric": f"chain-steps/sec on {H}x{H} grid x {n_local} chains{' per GPU' if world > 1 else ''}; accept-rate parity",
        "value": chain_steps / elapsed, "unit": "chain-steps/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64" if state == "f64" else "f64 arithmetic on f32 state", "data": "synthetic",
        "config": {"workload": f"largeScaleChain {H}x{H} grid, {n_local} chains/GPU, "
                               f"{'fp64' if state == 'f64' else 'fp32 state / fp64 arithmetic'}, Philox {gen_txt}, blocks 50-80, "
                               f"sigma_mc 5" + (f" ({tag})" if tag else ""),
                   "chains_total": n_total, "mh_steps_per_bench_step": inner, "steps_per_launch": steps_per_launch,
                   "timed_seconds": elapsed, "setup_seconds_factors": t_setup if generator == "cholesky" else None},
        "accept_rate": h_acc_rate, "final_loss_mean": float(h_loss.mean()),
        "roofline": roof,
    }


# sgs_weights_kernel: vector wave-instructions per simulated cell at the driver configuration (48 neighbours: ring search + 49-step
# Gauss-Jordan in registers): rocprofv3 SQ_INSTS_VALU summed over a run / the run's simulated cells (profiles/r04_sgs_weights_valu_per_cell.txt)
SGS_VALU_PER_CELL = 6.16e3
VALU_ISSUE_PEAK = 1024 * 2.4e9 / 4.25      # wave-instructions/s of the chip: 1024 SIMDs, one fp64-class instruction per 4.25 cycles
                                            # with four waves per SIMD (profiles/r02_valu_issue_rates.txt)


def _simulated_cells(outs, is_data):
    """Cells simulated by a set of chains: block cells without conditioning data, summed over all iterations (from the block records)."""
    import numpy as np
    H, W = is_data.shape
    free = np.zeros((H + 1, W + 1), dtype=np.int64)
    free[1:, 1:] = np.cumsum(np.cumsum(~is_data, axis=0), axis=1)
    tot = 0
    for o in outs:
        b = o[6]
        r0 = np.maximum(0, (b[:, 0] - b[:, 2] / 2).astype(int)); r1 = np.minimum(H, (b[:, 0] + b[:, 2] / 2).astype(int))
        c0 = np.maximum(0, (b[:, 1] - b[:, 3] / 2).astype(int)); c1 = np.minimum(W, (b[:, 1] + b[:, 3] / 2).astype(int))
        tot += int((free[r1, c1] - free[r0, c1] - free[r1, c0] + free[r0, c0]).sum())
    return tot


def measure_small_scale(H=64, n_chains=4, n_iter=400, cpu_iters=20):
    """BASELINE configs[0] (smallScaleChain 64x64 grid, 4 chains -- the reference's CPU-runnable case) through the product's
    small-scale chain on the device, with the reference DRIVER's own parameters (smallScaleChain_multiprocessing.py:489-556:
    set_sgs_param(48, 30e3) at 500 m = 60-cell search half-width, blocks 5-20, Matern, QuantileTransformer(1000), trend):
    SGS-block Metropolis iterations per second, all chains in one handle; a 256-chain throughput line beside it."""
    import numpy as np
    from mcmc_gpu_amd import sgs, synthetic
    prob, ch = synthetic.sgs_template(H, transform=True)
    is_data = ~np.isnan(prob["cond_bed"])
    beds = [prob["bed"] + np.random.default_rng(40 + i).normal(0, 3, prob["bed"].shape) for i in range(256)]
    sgs.run_many_sgs(ch, beds[:1], [np.random.default_rng(1)], 20)           # warm-up
    rngs = [np.random.default_rng(900 + i) for i in range(n_chains)]
    t0 = time.perf_counter()
    out, _ = sgs.run_many_sgs(ch, beds[:n_chains], rngs, n_iter)             # replay mode: the drop-in's default (host draws)
    dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    out_p, _ = sgs.run_many_sgs(ch, beds[:n_chains], rngs, n_iter, philox_seeds=[7000 + i for i in range(n_chains)])
    dt_p = time.perf_counter() - t0
    t0 = time.perf_counter()
    out_g, _ = sgs.run_many_sgs(ch, beds[:n_chains], [np.random.default_rng(900 + i) for i in range(n_chains)], n_iter, pcg64=True)
    dt_g = time.perf_counter() - t0
    n_big, it_big = 256, 100
    t0 = time.perf_counter()
    out_b, _ = sgs.run_many_sgs(ch, beds[:n_big], [None] * n_big, it_big, philox_seeds=[7000 + i for i in range(n_big)])
    dt_b = time.perf_counter() - t0
    cells_b = _simulated_cells(out_b, is_data)
    cpu = None
    if cpu_iters > 0:          # the oracle's restatement of chain_sgs.run (MCMC.py:1599-1911), one chain on one host core
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle"))
        import sgs_oracle as so
        cfg = so.SgsConfig(prob["xx"], prob["yy"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"], prob["cond_bed"],
                           prob["data_mask"], np.ones((H, H), dtype=int), prob["region_mask"], prob["resolution"], ch.sigma_mc,
                           list(ch.vario_param), list(ch.sgs_param), ch.block_min_x, ch.block_max_x, ch.block_min_y, ch.block_max_y,
                           trend=ch.trend if ch.detrend_map else None, nst_trans=ch.nst_trans if ch.do_transform else None)
        so.STABLE_TIES = True           # equidistant neighbours as the device takes them
        t1 = time.perf_counter()
        ref = so.run_chain_sgs(cfg, beds[0], cpu_iters, np.random.default_rng(900))
        cpu = {"value": cpu_iters / (time.perf_counter() - t1), "unit": "chain-iterations/s", "cores": 1, "kind": "port",
               "sample": f"1 chain x {cpu_iters} iterations, NumPy oracle of MCMC.py:1599-1911 at the same (driver) parameters",
               "first_iterations_equal_device": bool(np.array_equal(ref[4], out[0][4][:cpu_iters]) and np.array_equal(ref[6], out[0][6][:cpu_iters]))}
    ach = cells_b * SGS_VALU_PER_CELL / dt_b
    return {"cpu_baseline": cpu,
            "metric": f"chain-iterations/sec of the small-scale (SGS block) chain on a {H}x{H} grid x {n_chains} chains",
            "value": n_chains * n_iter / dt, "unit": "chain-iterations/s", "n_gpus": 1, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"smallScaleChain {H}x{H} grid, {n_chains} chains, the reference driver's parameters: Matern variogram "
                                   "(range 9932.5 m, nu 1.226), 48 neighbours within 30 km (search half-width 60 cells), blocks 5-20, sigma_mc 5, "
                                   f"detrended, QuantileTransformer(1000) normal-score transform {'on the device' if ch.do_transform else 'absent (scikit-learn not importable)'} "
                                   "(BASELINE configs[0], here on the GPU; replay mode: host draws with the reference's NumPy generator calls)",
                       "iterations": n_iter},
            "accept_rate": float(np.mean([o[4].mean() for o in out])), "timed_seconds": dt,
            "pcg64_mode": {"value": n_chains * n_iter / dt_g, "unit": "chain-iterations/s", "timed_seconds": dt_g,
                           "same_chain_as_replay_mode": bool(all(np.array_equal(x[4], y[4]) and np.array_equal(x[6], y[6]) for x, y in zip(out, out_g))),
                           "note": "the chains' own NumPy PCG64 streams advanced on the device (gsm_sgs_draw_pcg64): replay mode's chain without host draws"},
            "philox_mode": {"value": n_chains * n_iter / dt_p, "unit": "chain-iterations/s", "timed_seconds": dt_p,
                            "accept_rate": float(np.mean([o[4].mean() for o in out_p]))},
            "throughput_256_chains": {
                "value": n_big * it_big / dt_b, "unit": "chain-iterations/s", "mode": "philox", "timed_seconds": dt_b,
                "simulated_cells_per_s": cells_b / dt_b,
                "roofline": {"bound": "valu_issue", "kernel": "sgs_weights_kernel", "achieved": ach / 1e9, "peak": VALU_ISSUE_PEAK / 1e9,
                             "unit": "G wave-instructions/s", "frac": ach / VALU_ISSUE_PEAK,
                             "model": "simulated cells/s x 6.16 k vector wave-instructions per cell (ring search + 49-step Gauss-Jordan in "
                                      "registers; rocprofv3 SQ_INSTS_VALU, profiles/r04_sgs_weights_valu_per_cell.txt) against 1024 SIMDs x 2.4 GHz / 4.25 "
                                      "cycles per fp64-class instruction; wall time of the whole iteration (ranks + weights on the record streams; value pass and one launch for transforms, loss, decision and commit on the main stream), not of the kernel alone"}}}


def measure_pcg64_mode(H=256, n_chains=1024, n_steps=2048):
    """The headline geometry in the 'pcg64' draw mode: the reference's two NumPy PCG64 generator streams per chain advanced on the
    device bit for bit (gsm_draw_pcg64), device spectral synthesis, replay step kernel -- the reference-faithful chain without
    host draws.  Wall time of MCMC_gpu.run_many_pcg64 incl. engine setup and the upload / download of the beds."""
    import numpy as np
    from mcmc_gpu_amd import MCMC_gpu, synthetic
    prob, ch, rf = synthetic.template(H)
    beds = np.stack(list(synthetic.initial_beds(prob, n_chains)))
    st = [np.random.default_rng(seed=7 + i).bit_generator.state for i in range(n_chains)]
    MCMC_gpu.run_many_pcg64(ch, rf, beds[:8], st[:8], st[:8], 9)                    # warm-up
    t0 = time.perf_counter()
    tm = {}
    out, _, _ = MCMC_gpu.run_many_pcg64(ch, rf, beds, st, st, n_steps + 1, timing=tm)
    dt = time.perf_counter() - t0
    loop = tm["loop_seconds"]
    blocks = np.stack([o[6][1:] for o in out]).astype(np.int64)          # (chains, steps, 4): row, col, bh, bw
    acc = np.stack([o[4][1:] for o in out]).astype(np.int64)
    alg = algorithmic_bytes(blocks, acc, H, H)
    rate = n_chains * n_steps / loop
    return {"metric": f"chain-steps/sec on {H}x{H} grid x {n_chains} chains in the 'pcg64' draw mode",
            "value": rate, "unit": "chain-steps/s", "n_gpus": 1, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"largeScaleChain {H}x{H} grid, {n_chains} chains, fp64, NumPy PCG64 streams of the reference advanced on the "
                                   "device (draws, block records, accept masks and generator states of the CPU reference on the same seeds), "
                                   "blocks 50-80, sigma_mc 5", "steps": n_steps, "steps_per_batch": tm.get("batch"),
                       "synthesis_and_step_fused": tm.get("fused")},
            "accept_rate": float(acc.mean()), "timed_seconds": loop,
            "wall_seconds_incl_engine_setup_and_bed_transfers": dt, "rate_incl_engine_setup_and_bed_transfers": n_chains * n_steps / dt,
            # same numerator as the headline line (the chain state a step must move), over the whole batch loop: the
            # white-noise planes that gsm_draw_pcg64 hands to the chain kernel through HBM are traffic of this mode's own making
            "roofline": {"bound": "hbm", "achieved": alg / loop / 1e9, "peak": 8000.0, "unit": "GB/s", "frac": alg / loop / 8e12, "traffic": None,
                         "kernel": "pcg64_draw_kernel + chain_strip_kernel<NOISE> (gsm_run_noise), one after the other",
                         "algorithmic_bytes_per_chain_step": alg / (n_chains * n_steps)},
            "note": "value = the draw / synthesis / step batches with the chain state resident in HBM (synchronised before and after); "
                    "host-drawn replay mode on the same box: 25-41 k chain-steps/s with 15 draw workers (scripts/replay_batch_bench.py)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--chains", type=int, default=1024, help="chains per GPU")
    ap.add_argument("--inner", type=int, default=2048,
                    help="Metropolis steps per chain in one bench step (2048: --steps 20 times several seconds)")
    ap.add_argument("--batch", type=int, default=32, help="steps per kernel launch of the two-kernel pipeline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--no-extras", action="store_true", help="skip the configs[3] / configs[4]-shard lines")
    ap.add_argument("--gather-beds", action="store_true", help="also all-gather the final beds (512 MiB per GPU at 256^2)")
    ap.add_argument("--generator", choices=["spectral", "cholesky"], default="spectral",
                    help="proposal generator: the reference's spectral synthesis (headline) or precomputed Cholesky factors (BASELINE configs[3])")
    ap.add_argument("--classes", type=int, default=2, help="range classes of the Cholesky generator")
    ap.add_argument("--state", choices=["f64", "f32"], default="f64",
                    help="per-chain state storage: f64 (headline) or f32 with f64 arithmetic (BASELINE configs[4])")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous, barrier and max-over-ranks only (no GPU work): checks that --gpus N starts N ranks")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))       # nothing above touched the GPU

    from mcmc_gpu_amd import parallel
    rank, local_rank, world = parallel.dist_env()
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
              "(or without a launcher: bench.py starts its own ranks)", file=sys.stderr)
        sys.exit(2)
    if args.launch_check:
        parallel.init_distributed()
        parallel.barrier()
        nccl = world > 1 and torch.distributed.get_backend() == "nccl"
        tmax = parallel.max_over_ranks(float(rank), torch.device("cuda", local_rank) if nccl else torch.device("cpu"))
        ws = torch.distributed.get_world_size() if world > 1 else 1
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": world, "world_size": ws, "max_rank_seen": tmax,
                              "backend": torch.distributed.get_backend() if world > 1 else None}))
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    default_workload = (args.grid, args.chains, args.generator, args.state) == (256, 1024, "spectral", "f64")
    cpu_res = None
    if world == 1 and not args.no_cpu_baseline:
        cpu_res = cpu_baseline(args.grid, args.cpu_budget)   # before the GPU is initialised: the pool forks
    rank, local_rank, world = parallel.init_distributed()
    dev_index = local_rank if torch.cuda.device_count() > local_rank else 0
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    inner = args.inner if args.generator == "spectral" else min(args.inner, 256)
    out = measure(args, args.grid, args.chains, args.generator, args.state, inner, args.batch, args.steps, args.warmup,
                  rank, world, dev, classes=args.classes)
    if rank == 0:
        out["world_size"] = torch.distributed.get_world_size() if world > 1 else 1
        out["launcher"] = "self" if os.environ.get("GSM_BENCH_SELF_LAUNCHED") else ("torchrun" if world > 1 else "single process")
        if world > 1:
            out["backend"] = torch.distributed.get_backend()
        if cpu_res is not None:
            out["cpu_baseline"] = cpu_res
    if world == 1 and default_workload and not args.no_extras:
        extras = {}
        for name, kw in (("configs[3]", dict(H=512, n_local=1024, generator="cholesky", state="f64", inner=256, batch=128, steps=3, warmup=1)),
                         ("configs[4] one GPU's shard", dict(H=1024, n_local=512, generator="spectral", state="f32", inner=512, batch=32, steps=3, warmup=1))):
            t0 = time.perf_counter()
            try:
                r = measure(args, rank=rank, world=world, dev=dev, classes=2, headline=False, **kw)
                r["wall_seconds_incl_setup"] = time.perf_counter() - t0
                extras[name] = r
            except Exception as e:      # the headline line must not be lost to a failure here; the failure is reported
                extras[name] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
        t0 = time.perf_counter()
        try:
            extras["configs[1] in the pcg64 draw mode"] = measure_pcg64_mode()
            extras["configs[1] in the pcg64 draw mode"]["wall_seconds_incl_setup"] = time.perf_counter() - t0
        except Exception as e:
            extras["configs[1] in the pcg64 draw mode"] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
        t0 = time.perf_counter()
        try:
            extras["configs[0] on the device"] = measure_small_scale(cpu_iters=0 if args.no_cpu_baseline else 20)
            extras["configs[0] on the device"]["wall_seconds_incl_setup"] = time.perf_counter() - t0
        except Exception as e:
            extras["configs[0] on the device"] = {"error": f"{type(e).__name__}: {e}"}
        out["other_configs"] = extras
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
