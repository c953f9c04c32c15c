"""Dev: GPU vs oracle on every instance of the perturbed Netlib batches (config 3): exit codes and iteration counts."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../..")
import numpy as np
import eicos_amd
from eicos_amd.generate import perturbed_batch, SEED
from oracle import oracle as orc
for name in sys.argv[1:]:
    pat, sets = eicos_amd.read_problem(f"tests/golden/{name}.epb")
    B = 256
    d = perturbed_batch(pat, sets[0], 0, B, SEED)
    g = eicos_amd.BatchSolver(pat, B)
    g.update(d["Gpr"], d["Apr"], d["c"], d["h"], d["b"])
    codes = g.solve(); ia = g.info_arrays()
    r = orc.batch_solve(pat, d["Gpr"], d["Apr"], d["c"], d["h"], d["b"], 16)
    diff = np.abs(r["iters"].astype(int) - ia["iter"].astype(int))
    same_code = r["exitcodes"] == codes
    print(name, "codes gpu", dict(collections.Counter(codes.tolist())), "oracle", dict(collections.Counter(r["exitcodes"].tolist())),
          "same code", int(same_code.sum()), "| iter diff: ==0", int((diff == 0).sum()), "<=1", int((diff <= 1).sum()), "max", int(diff.max()))
    bad = np.flatnonzero(diff > 1)
    for i in bad[:10]:
        print("   inst", i, "gpu", codes[i], ia["iter"][i], "%.10e" % ia["pcost"][i], "oracle", r["exitcodes"][i], r["iters"][i], "%.10e" % r["pcost"][i])
    g.close()
