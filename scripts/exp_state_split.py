"""Same-box A/B of where the fused kernel issues its 2 KT state loads relative to the coefficient passes (kStateSplit0 / 1).
    python scripts/exp_state_split.py   (builds libgsm_s<a><b>.so), then on the GPU box: python scripts/ab_lib.py v0 s00 s04 ..."""
import sys
sys.path.insert(0, 'scripts')
from build_variant import build, sub
for a, b in ((0, 0), (0, 4), (0, 7), (4, 4), (4, 7), (7, 7), (2, 5)):
    build(f"s{a}{b}", lambda src, a=a, b=b: sub(src / "chain_fused_kernel.hip", "constexpr int kStateSplit0 = 0, kStateSplit1 = 0;",
                                               f"constexpr int kStateSplit0 = {a}, kStateSplit1 = {b};"))
