"""Soak run on the GPU box: many steps at the headline size, then the full-size invariants of
tests/test_gpu_fullsize.py (carried energy == from-scratch recompute bit for bit, carried sum == sum of energy,
resampled counts == accepted windows).  Usage: python scripts/soak.py [n_steps] [n_chains]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from mcmc_gpu_amd import synthetic

n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n_chains = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
H = 256
prob, ch, rf = synthetic.template(H)
eng = ch._make_engine(rf, n_chains, 0)
beds0 = synthetic.initial_beds(prob, n_chains)
loss0 = eng.set_state(beds0)
seeds = list(range(9000, 9000 + n_chains))
t0 = time.time()
loss, acc, blk = eng.run_philox(n_steps, 0, seeds, rf, batch=32)
dt = time.time() - t0
beds = eng.beds.clone(); energy = eng.energy.clone(); res = eng.resampled.clone(); lsum = eng.loss_sum.clone()
loss_re = eng.set_state(beds)
ok_energy = torch.equal(eng.energy, energy)
s_c = (lsum[:, 0] + lsum[:, 1]).cpu().numpy(); s_e = energy.sum(dim=(1, 2), dtype=torch.float64).cpu().numpy()
rel_sum = np.abs(s_c - s_e).max() / s_e.max()
rel_loss = np.abs(loss[:, -1] - loss_re).max() / loss_re.max()
resh = res.cpu().numpy(); bad = 0
for c in (0, 7, n_chains // 2, n_chains - 1):
    cnt = np.zeros((H, H), dtype=np.int64)
    for (row, col, bh, bw), a in zip(blk[c], acc[c]):
        if a:
            cnt[max(0, row - bh // 2):min(H, row + bh // 2), max(0, col - bw // 2):min(H, col + bw // 2)] += 1
    bad += int(not np.array_equal(resh[c], cnt * (prob["region_mask"] == 1)))
print(f"{n_chains} chains x {n_steps} steps in {dt:.2f} s ({n_chains * n_steps / dt / 1e6:.2f} M chain-steps/s incl. D2H); "
      f"accept {acc.mean():.4f}; energy bit-equal to recompute: {ok_energy}; |carried sum - sum(energy)| rel {rel_sum:.2e}; "
      f"|last loss - recomputed| rel {rel_loss:.2e}; resampled mismatches {bad}; loss {loss0.mean():.1f} -> {loss[:, -1].mean():.1f}")
assert ok_energy and rel_sum < 1e-12 and rel_loss < 1e-10 and bad == 0
