"""Build a patched copy of the HIP sources into mcmc_gpu_amd/libgsm_<name>.so for same-box A/B runs (scripts/ab_lib.py).

    from build_variant import build, sub
    build("kb4", lambda src: sub(src / "chain_fused_kernel.hip", "? 2 : KT;", "? 4 : KT;"))

The copy lives under /tmp; the committed sources are not touched."""
import sys, shutil, subprocess, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mcmc_gpu_amd import _lib
def build(name, patch):
    src = Path('/tmp/gsm_variants') / name
    src.parent.mkdir(parents=True, exist_ok=True)
    if src.exists(): shutil.rmtree(src)
    shutil.copytree(_lib.CSRC, src)
    # keep the relative include of ../../include/gsm.h working
    for f in list(src.glob('*.h')) + list(src.glob('*.hip')):
        t = f.read_text().replace('"../../include/gsm.h"', '"' + str(_lib.HEADER) + '"'); f.write_text(t)
    patch(src)
    objs = []; procs = []
    for s in _lib.SOURCES:
        obj = src / (s + '.o'); objs.append(str(obj))
        cmd = ['hipcc', *_lib.HIPCC_FLAGS, *_lib.EXTRA_FLAGS.get(s, []), '-c', '-o', str(obj), str(src / s)]
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode: raise RuntimeError(s + '\n' + out)
    out = _lib.PKG_DIR / f'libgsm_{name}.so'
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', str(out), *objs], check=True)
    print('built', out)
def sub(path, old, new, count=1):
    s = path.read_text(); assert old in s, (path, old[:60]); path.write_text(s.replace(old, new, count))
