"""ngsld_amd -- MI355X-native pairwise-LD engine, drop-in for the pair-LD path of fgvieira/ngsLD.

The product is the C-ABI library built from ``ngsld_amd/csrc`` (hand-written HIP for gfx950, declared in
``include/ngsld.h``) and the ``ngsLD`` command-line binary on top of it.  This Python package is the
thin host-side plumbing around that library: the ctypes binding, build helpers, the synthetic-input
generator and the multi-GPU sharding helpers used by ``bench.py`` and the tests.
"""
__version__ = "0.4.0"
