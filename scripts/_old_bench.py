#!/usr/bin/env python3
"""Throughput of the large-scale-chain Metropolis hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 256x256 synthetic grid, 1024 chains per GPU, fp64,
Matern(0.9125) spectral proposals, blocks 50-80 cells, sigma_mc = 5, Philox draws generated on the device.
One bench "step" = `--inner` Metropolis steps of every chain = one gsm_run_philox call = one launch of the fused chain
kernel (proposal + Metropolis step per chain-step inside it; GSM_FUSED=0 selects the older two-kernel pipeline).  Weak scaling: every
rank runs its own 1024 chains (different seeds), no collective inside the step loop, one all-gather of the
per-chain caches and an all-reduce of the posterior-mean field at the end of the timed region.

Prints ONE JSON line on rank 0 (see the driver contract): metric chain-steps/s, plus
  roofline     -- algorithmic HBM bytes (SURVEY.md 8d formula, summed exactly from the recorded blocks/accepts)
                  per launch of the dominant kernel / its average duration (HIP events on its own stream)
  cpu_baseline -- the NumPy oracle (bit-validated restatement of the reference's CPU loop) under the reference's
                  multiprocessing.Pool pattern on this host's cores, bounded sample, rank 0 at N=1 only.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F64_PEAK_TFLOPS = 78.6   # MI355X fp64 matrix peak


def pmc_traffic(H, n_chains, steps_per_launch):
    """HBM-side bytes per launch of the fused chain kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of scripts/pmc_fused.py,
    calibrated on a stream copy of known size, MI355X_MICROARCH.md HBM section), plus the SQ counters of the same
    workload.  (None, None) when no measurement matches this config."""
    try:
        d = json.load(open(ROOT / "profiles" / "pmc_traffic.json"))
        if (d["grid"], d["chains"], d["steps_per_launch"]) == (H, n_chains, steps_per_launch):
            sq = {k: d[k] for k in ("valu_busy_frac_per_simd", "mfma_busy_frac_per_simd",
                                    "valu_wave_instructions_per_chain_step", "mfma_instructions_per_chain_step") if k in d}
            return d["hbm_bytes_per_launch"], sq
    except Exception:
        pass
    return None, None


def algorithmic_bytes(blocks, accept, H, W, state_bytes=8):
    """s*2*B_eff + a*(s*2*B_eff + 8*B_eff) summed over every chain-step (B_eff = clipped window)."""
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
            "accept_rate": float(np.mean([r[2] for r in res]))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--grid", type=int, default=256)
    ap.add_argument("--chains", type=int, default=1024, help="chains per GPU")
    ap.add_argument("--inner", type=int, default=128, help="Metropolis steps per chain in one bench step")
    ap.add_argument("--batch", type=int, default=32, help="steps per kernel launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--gather-beds", action="store_true", help="also all-gather the final beds (512 MiB per GPU at 256^2)")
    ap.add_argument("--generator", choices=["spectral", "cholesky"], default="spectral",
                    help="proposal generator: the reference's spectral synthesis (headline) or precomputed Cholesky factors (BASELINE configs[3])")
    ap.add_argument("--classes", type=int, default=2, help="range classes of the Cholesky generator")
    ap.add_argument("--state", choices=["f64", "f32"], default="f64",
                    help="per-chain state storage: f64 (headline) or f32 with f64 arithmetic (BASELINE configs[4])")
    args = ap.parse_args()

    from mcmc_gpu_amd import parallel, synthetic
    rank, local_rank, world = parallel.dist_env()
    cpu_res = None
    if world == 1 and not args.no_cpu_baseline:
        cpu_res = cpu_baseline(args.grid, args.cpu_budget)   # before the GPU is initialised: the pool forks
    rank, local_rank, world = parallel.init_distributed()
    if world != args.gpus and rank == 0:
        print(f"# note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    dev_index = local_rank if torch.cuda.device_count() > local_rank else 0
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    H = args.grid
    n_local = args.chains
    n_total = n_local * world
    prob, ch, rf = synthetic.template(H)
    ch.state_dtype = args.state
    beds0 = synthetic.initial_beds(prob, n_local, first=rank * n_local)
    seeds = [7 + rank * n_local + i for i in range(n_local)]
    eng = ch._make_engine(rf, n_local, dev_index)
    eng.set_state(beds0)
    del beds0
    eng.enable_timing(True)
    if args.generator == "cholesky":
        from mcmc_gpu_amd import cholesky as chol
        rf.generator = "cholesky"
        chol.build_factors(eng, rf, n_classes=args.classes)
    p = eng.rf_struct(rf)
    inner, batch = args.inner, args.batch
    n_timed = args.steps * inner
    loss = torch.empty((n_local, inner), dtype=torch.float64, device=dev)
    acc = torch.empty((n_local, inner), dtype=torch.uint8, device=dev)
    blk = torch.empty((n_local, inner, 4), dtype=torch.int32, device=dev)
    acc_all = torch.empty((n_local, n_timed), dtype=torch.uint8, device=dev)
    loss_all = torch.empty((n_local, n_timed), dtype=torch.float64, device=dev)
    blk_all = torch.empty((n_local, n_timed, 4), dtype=torch.int32, device=dev)

    def segment_end():
        """Per-chain caches to every rank, posterior-mean field, small results to the host."""
        g_loss = parallel.all_gather_chains(loss_all, n_total)
        g_acc = parallel.all_gather_chains(acc_all, n_total)
        mean_field = parallel.all_reduce_mean_field(eng.beds.sum(dim=0, dtype=torch.float64), n_total)
        if args.gather_beds:
            parallel.all_gather_chains(eng.beds, n_total)
        return g_loss[:, -1].cpu(), float(g_acc.float().mean().item()), mean_field.cpu()

    step0 = 0
    for _ in range(args.warmup):
        eng.run_philox(inner, step0, seeds, p, batch=batch, out=(loss, acc, blk), to_host=False)
        step0 += inner
    if args.warmup > 0:
        acc_all.zero_(); loss_all.zero_()
        segment_end()          # loads torch's reduce/copy kernels and RCCL channels outside the timed region
    torch.cuda.synchronize(dev)
    parallel.barrier()
    torch.cuda.synchronize(dev)
    t_step = t_prop = 0.0
    n_step_l = n_prop_l = 0
    t0 = time.perf_counter()
    for k in range(args.steps):
        eng.run_philox(inner, step0, seeds, p, batch=batch, out=(loss, acc, blk), to_host=False)
        step0 += inner
        sl = slice(k * inner, (k + 1) * inner)
        acc_all[:, sl] = acc; loss_all[:, sl] = loss; blk_all[:, sl] = blk
        tm = eng.last_timing()
        t_step += tm["step_ms"] * tm["step_launches"]; n_step_l += tm["step_launches"]
        t_prop += tm["proposal_ms"] * tm["proposal_launches"]; n_prop_l += tm["proposal_launches"]
    h_loss, h_acc_rate, h_mean = segment_end()
    torch.cuda.synchronize(dev)
    parallel.barrier()
    torch.cuda.synchronize(dev)
    elapsed = parallel.max_over_ranks(time.perf_counter() - t0, dev)

    # two-kernel pipeline only (GSM_FUSED=0): the kernels co-run on two streams in the timed region, which stretches each
    # one's duration; time one launch of each ALONE as well (outside the timed region) so the roofline can be read both ways
    iso = None
    fused = args.generator == "spectral" and os.environ.get("GSM_FUSED", "1") != "0"
    if args.generator == "spectral" and not fused:
        import ctypes as C
        from mcmc_gpu_amd.engine import _ptr
        nrec = n_local * batch
        t_si = torch.empty(nrec, dtype=torch.int32, device=dev); t_ce = torch.empty(2 * nrec, dtype=torch.int32, device=dev)
        t_u = torch.empty(nrec, dtype=torch.float64, device=dev)
        t_f = torch.empty((nrec, eng.field_stride), dtype=torch.float64, device=dev)
        t_l = torch.empty(nrec, dtype=torch.float64, device=dev); t_a = torch.empty(nrec, dtype=torch.uint8, device=dev)
        d_seeds = eng._seeds(seeds)
        tp, ts, ab = [], [], []
        for r in range(3):
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            eng._check(eng.lib.gsm_propose_philox(eng.h, batch, step0 + r * batch, _ptr(d_seeds), C.byref(p), _ptr(t_si), _ptr(t_ce),
                                                  _ptr(t_u), _ptr(t_f), eng.field_stride, None, eng._stream()))
            e1.record()
            eng._check(eng.lib.gsm_run_replay(eng.h, batch, _ptr(eng.beds), _ptr(eng.energy), _ptr(eng.resampled), _ptr(eng.loss_sum),
                                              _ptr(t_si), _ptr(t_ce), _ptr(t_u), _ptr(t_f), eng.field_stride, _ptr(t_l), _ptr(t_a),
                                              eng._stream()))
            e2.record(); torch.cuda.synchronize(dev)
            tp.append(e0.elapsed_time(e1)); ts.append(e1.elapsed_time(e2))
            si_h = t_si.cpu().numpy()
            b_h = np.concatenate([t_ce.view(n_local, batch, 2).cpu().numpy(), eng.bh[si_h].reshape(n_local, batch, 1),
                                  eng.bw[si_h].reshape(n_local, batch, 1)], axis=2)
            ab.append(algorithmic_bytes(b_h, t_a.view(n_local, batch).cpu().numpy(), H, H, 8 if args.state == "f64" else 4))
        iso = {"step_kernel_ms": float(np.median(ts)), "propose_kernel_ms": float(np.median(tp)), "bytes_per_launch": float(np.mean(ab))}
        del t_f

    chain_steps = n_total * n_timed
    value = chain_steps / elapsed
    if rank == 0:
        blocks_h = blk_all.cpu().numpy(); acc_h = acc_all.cpu().numpy()
        bytes_total = algorithmic_bytes(blocks_h, acc_h, H, H, 8 if args.state == "f64" else 4)
        bytes_per_launch = bytes_total / max(n_step_l, 1)
        step_ms = t_step / max(n_step_l, 1); prop_ms = t_prop / max(n_prop_l, 1)
        dom = "step_kernel" if t_step >= t_prop else "propose_kernel"
        if fused:
            dom = "chain_fused_kernel"
        dom_ms = prop_ms if dom == "propose_kernel" else step_ms
        achieved = bytes_per_launch / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic, sq = pmc_traffic(H, n_local, inner) if fused else (None, None)
        roof = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": dom,
                "algorithmic_bytes_per_chain_step": bytes_total / (n_local * n_timed),
                "bytes_per_launch": bytes_per_launch, "kernel_ms": dom_ms, "launches": n_step_l}
        if fused:
            # what actually limits the fused kernel: fp64 VALU issue (Philox + Box-Muller + spectral amplitude + index
            # math) with the fp64 matrix pipe on the same datapath -- rocprofv3 SQ counters of the same workload
            if sq:
                roof["issue_limits"] = sq
        else:
            roof["step_kernel_ms"] = step_ms; roof["propose_kernel_ms"] = prop_ms
        if iso is not None:
            for k in ("step_kernel", "propose_kernel"):
                iso[k + "_frac"] = iso["bytes_per_launch"] / (iso[k + "_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
            iso["note"] = "one launch of each kernel alone on an idle GPU (torch events), outside the timed region"
            roof["isolated"] = iso
        if args.generator == "cholesky":
            # algorithmic flops per chain-step = (bh*bw)^2 (lower-triangular L z)
            nn = (blocks_h[..., 2].astype(np.float64) * blocks_h[..., 3]) ** 2
            flops_per_launch = float(nn.sum()) / max(n_prop_l, 1)
            ach = flops_per_launch / (prop_ms * 1e-3) / 1e12 if prop_ms > 0 else 0.0
            roof = {"bound": "mfma", "achieved": ach, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ach / MFMA_F64_PEAK_TFLOPS, "traffic": None, "kernel": "cz_* proposal pipeline (zgen + gemm)",
                    "algorithmic_flops_per_chain_step": float(nn.mean()), "flops_per_launch": flops_per_launch,
                    "step_kernel_ms": step_ms, "propose_kernel_ms": prop_ms, "launches": n_prop_l}
        out = {
            "metric": "chain-steps/sec on 256x256 grid x 1024 chains; accept-rate parity",
            "value": value, "unit": "chain-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if args.state == "f64" else "f64 arithmetic on f32 state", "data": "synthetic",
            "config": {"workload": f"largeScaleChain {H}x{H} grid, {n_local} chains/GPU, {'fp64' if args.state == 'f64' else 'fp32 state / fp64 arithmetic'}, Philox "
                                   + ("spectral (Matern 0.9125) proposals, blocks 50-80, sigma_mc 5 "
                                      + ("(BASELINE configs[1])" if args.state == "f64" else "(BASELINE configs[4], one GPU's shard)")
                                      if args.generator == "spectral" else
                                      f"precomputed-Cholesky (Matern 0.9125, {args.classes} range classes) proposals, "
                                      "blocks 50-80, sigma_mc 5 (BASELINE configs[3])"),
                       "chains_total": n_total, "mh_steps_per_bench_step": inner,
                       "steps_per_launch": inner if fused else batch},
            "accept_rate": h_acc_rate, "final_loss_mean": float(h_loss.mean()),
            "roofline": roof,
        }
        if cpu_res is not None:
            out["cpu_baseline"] = cpu_res
        print(json.dumps(out))
    eng.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
