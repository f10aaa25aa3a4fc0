"""hipcc -S of one HIP source (device side) with the library's flags: python scripts/isa_dump.py <src.hip> <out.s>"""
import subprocess, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mcmc_gpu_amd import _lib
src, out = sys.argv[1], sys.argv[2]
cmd = ['hipcc', *_lib.HIPCC_FLAGS, *_lib.EXTRA_FLAGS.get(Path(src).name, []), '-S', '--cuda-device-only', '-o', out, src]
subprocess.run(cmd, check=True, capture_output=True)
