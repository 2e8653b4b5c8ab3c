import importlib, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _parity as par, _oracle as orc
nfx = importlib.import_module("nuclei-feature-extraction_amd")
set_id, ps, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
off, xy, pu8 = par.make_inputs(5, n, ps)
ctx = nfx.Context(0)
got, _ = ctx.extract([set_id], off, xy, pu8, ps, batch_size=100)
want, _ = orc.extract([set_id], off, xy, pu8, ps, batch_size=100)
np.set_printoptions(precision=5, suppress=True, linewidth=200)
for i in range(n):
    ok = np.isclose(got[i], want[i], rtol=1e-4, atol=1e-4, equal_nan=True)
    print("nucleus", i, "ok cols", int(ok.sum()), "of", len(ok), "got[:5]", got[i][:5], "want[:5]", want[i][:5])
