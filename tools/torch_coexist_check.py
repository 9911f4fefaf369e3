"""Diagnostic: the product library next to an imported torch (shared HIP/RCCL runtime)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
from hydra_amd import capi, synth
print("torch", torch.__version__, "hip", torch.version.hip)
dev = capi.Device(0)
dev.comm_init(1, 0, capi.Device.unique_id())
geno = synth.make_genotypes(100, 900, seed=1)
y, _ = synth.make_phenotype(geno, seed=2, causal_frac=0.05)
dev.load_bed(synth.pack_bed_columns(geno), 900)
dev.set_option("force_split", 1)
ch = capi.Chain(dev, y, seed=3)
for _ in range(2):
    ch.iterate()
print("ok nnz", ch.last_nnz(), "sigmaE", ch.state()["sigmaE"])
maps = open("/proc/self/maps").read()
print(sorted({l.split()[-1] for l in maps.splitlines() if "amdhip64" in l or "rccl" in l}))
