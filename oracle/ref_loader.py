"""TEST INFRASTRUCTURE ONLY -- loader for the upstream reference (this container only).

Imports the reference's ``gstatsMCMC`` package from ``/root/reference`` so that
``oracle/make_fixtures.py`` can (a) validate the clean-room NumPy restatement in
``oracle/mcmc_oracle.py`` bit-for-bit and (b) emit the golden vectors committed under
``tests/golden/``.  ``/root/reference`` does not exist on the GPU box; nothing in
``tests/``, ``bench.py`` or ``__graft_entry__`` calls this module at run time.

The reference imports eight third-party packages at module scope that are not
installed here (gstatsim, gstools, skgstat, IPython, xarray, pyproj, verde, numba --
SURVEY.md section 8c).  None of them is touched by the large-scale-chain hot path
(``chain_crf.run`` with ``spectral=True``), so empty ``types.ModuleType`` stubs are
registered for them.  Bytecode writing is disabled: the reference tree is read-only
by contract.
"""
import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "gstatsMCMC"))


def load_reference():
    """Return (MCMC, Topography, MCMC_gpu, gstatsim_custom) modules of the reference."""
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
    sys.dont_write_bytecode = True
    os.environ.setdefault("MPLBACKEND", "Agg")
    import matplotlib
    matplotlib.use("Agg")

    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    ident = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    stub("gstatsim")
    stub("gstools")
    skg = stub("skgstat")
    skg.models = stub("skgstat.models")
    ip = stub("IPython")
    ip.display = stub("IPython.display")
    stub("xarray")
    stub("pyproj", CRS=object, Transformer=object)
    stub("verde")
    stub("numba", njit=ident, prange=range, jit=ident)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from gstatsMCMC import MCMC, Topography, MCMC_gpu  # noqa: E402
    from gstatsMCMC import gstatsim_custom  # noqa: E402
    return MCMC, Topography, MCMC_gpu, gstatsim_custom
