"""hydra_amd -- MI355X-native BayesRR single-site Gibbs hot path.

The product is the HIP library behind include/hgibbs.h (hydra_amd/csrc, built
in-tree as hydra_amd/libhgibbs.so) and the hydra-compatible CLI
hydra_amd/bin/hydra_mi355x.  The Python modules are plumbing: `capi` binds the
C ABI with ctypes, `synth` makes seeded synthetic PLINK data.
"""
from . import synth  # noqa: F401
