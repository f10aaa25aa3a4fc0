"""Experiment (timing only, results of the variants are NOT valid chains): does the fused kernel's state-load phase depend on the
NUMBER of vector-memory instructions or on the bytes?  v0 = HEAD (14 loads of 8 B per wave and step); h8 = 8 loads of 8 B (the
other cells reuse a neighbour's value); w16 = 8 loads of 16 B (two adjacent cells per load).  python scripts/exp_wide_loads.py"""
import sys
sys.path.insert(0, 'scripts')
from build_variant import build, sub

OLD = '''        if (k * kNT + 64 * wave < G.ncell) {
          int i, lr, lc; uint32_t g; bool valid, inwin;
          cell(G, W, k, i, lr, lc, g, valid, inwin);
          vb[k] = StateIO<TS>::load(r_bed, valid ? g * (uint32_t)sizeof(TS) : kOOB);
          ve[k] = StateIO<TS>::load(r_en, inwin ? g * (uint32_t)sizeof(TS) : kOOB);
          rq[k] = (uint32_t)lr | ((uint32_t)lc << 8) | (valid ? 1u << 16 : 0u) | (inwin ? 1u << 17 : 0u);
        } else {
          rq[k] = 0;
          vb[k] = StateIO<TS>::load(r_bed, kOOB);
          ve[k] = StateIO<TS>::load(r_en, kOOB);
        }'''
H8 = '''        if (k * kNT + 64 * wave < G.ncell) {
          int i, lr, lc; uint32_t g; bool valid, inwin;
          cell(G, W, k, i, lr, lc, g, valid, inwin);
          if ((k & 1) == 0) {
            vb[k] = StateIO<TS>::load(r_bed, valid ? g * (uint32_t)sizeof(TS) : kOOB);
            ve[k] = StateIO<TS>::load(r_en, inwin ? g * (uint32_t)sizeof(TS) : kOOB);
          } else { vb[k] = valid ? vb[k - 1] : 0.0; ve[k] = inwin ? ve[k - 1] : 0.0; }
          rq[k] = (uint32_t)lr | ((uint32_t)lc << 8) | (valid ? 1u << 16 : 0u) | (inwin ? 1u << 17 : 0u);
        } else {
          rq[k] = 0;
          if ((k & 1) == 0) { vb[k] = StateIO<TS>::load(r_bed, kOOB); ve[k] = StateIO<TS>::load(r_en, kOOB); }
          else { vb[k] = 0.0; ve[k] = 0.0; }
        }'''
W16 = '''        if (k * kNT + 64 * wave < G.ncell) {
          int i, lr, lc; uint32_t g; bool valid, inwin;
          cell(G, W, k, i, lr, lc, g, valid, inwin);
          if ((k & 1) == 0) {
            const double2 b2 = ld_f64x2(r_bed, valid ? g * 8u : kOOB, 0u);
            const double2 e2 = ld_f64x2(r_en, inwin ? g * 8u : kOOB, 0u);
            vb[k] = b2.x; ve[k] = e2.x;
            if (k + 1 < KT) { vb[k + 1] = b2.y; ve[k + 1] = e2.y; }
          } else { vb[k] = valid ? vb[k] : 0.0; ve[k] = inwin ? ve[k] : 0.0; }
          rq[k] = (uint32_t)lr | ((uint32_t)lc << 8) | (valid ? 1u << 16 : 0u) | (inwin ? 1u << 17 : 0u);
        } else {
          rq[k] = 0;
          if ((k & 1) == 0) {
            const double2 b2 = ld_f64x2(r_bed, kOOB, 0u); const double2 e2 = ld_f64x2(r_en, kOOB, 0u);
            vb[k] = b2.x; ve[k] = e2.x;
            if (k + 1 < KT) { vb[k + 1] = b2.y; ve[k + 1] = e2.y; }
          }
        }'''
WAIT_OLD = '"n"(2 * KT)'
WAIT_NEW = '"n"(2 * ((KT + 1) / 2))'

def patch(body):
    def f(src):
        p = src / "chain_fused_kernel.hip"
        sub(p, OLD, body)
        sub(p, WAIT_OLD, WAIT_NEW)
    return f

build("v0", lambda src: None)
build("h8", patch(H8))
build("w16", patch(W16))
