import importlib, sys, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import _parity as par, _oracle as orc
nfx = importlib.import_module("nuclei-feature-extraction_amd")
set_id, ps, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
off, xy, pu8 = par.make_inputs(5, n, ps)
ctx = nfx.Context(0)
got, _ = ctx.extract([set_id], off, xy, pu8, ps, batch_size=100)
want, _ = orc.extract([set_id], off, xy, pu8, ps, batch_size=100)
print("set", set_id, "ps", ps, "n", n, "worst", par.assert_rows_close(got, want, par.column_scales(nfx, [set_id], ps)))
