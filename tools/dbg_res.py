import sys, numpy as np, os, subprocess
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
if len(sys.argv) > 1:
    window, M, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    from hydra_amd import capi, synth
    geno = synth.make_genotypes(M, N, seed=M + N, missing_rate=0.0)
    y, _ = synth.make_phenotype(geno, seed=M + N + 1, h2=0.5, causal_frac=0.05)
    bed = synth.pack_bed_columns(geno)
    dev = capi.Device(0); dev.load_bed(bed, N); dev.set_option("engine", 2); dev.set_option("window", window); dev.set_option("res_timeout_ms", 200); dev.set_option("res_deadline_ms", 4000)
    ch = capi.Chain(dev, y, seed=1222, shuffle=1)
    for it in range(2):
        try:
            ch.iterate()
        except Exception as e:
            print("FAILED", e, flush=True); os._exit(3)
        print("window", window, "M", M, "N", N, "it", it, "rounds", dev.sweep_stats()["rounds"], flush=True)
else:
    for (w, M, N) in [(8, 40, 37)]:
        try:
            r = subprocess.run([sys.executable, __file__, str(w), str(M), str(N)], capture_output=True, text=True, timeout=15)
            print(r.stdout[-300:], r.stderr[-900:], flush=True)
        except subprocess.TimeoutExpired:
            print("TIMEOUT window", w, M, N, flush=True)
