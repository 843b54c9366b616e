import sys, torch
sys.path.insert(0, '.')
from collections import OrderedDict
from scarches_amd import _lib
orig = _lib.check
def chk(rc, what):
    torch.cuda.synchronize()
    print(what, rc, flush=True)
    orig(rc, what)
_lib.check = chk
import scarches_amd.engine as E
from tests._util import load_case
z, cfg, W, Gr, masks = load_case("trvae_nb_ln_small")
eng = E.CVAEEngine(dict(cfg), OrderedDict((k, torch.tensor(v)) for k, v in W.items()), "cuda:0")
eng.overlap = False
eng.bind_csr(torch.tensor(z["x_indptr"]).cuda(), torch.tensor(z["x_indices"]).cuda(), torch.tensor(z["x_data"]).cuda())
eng.begin_step(0.37, 1e-3)
eng.forward_backward(None, torch.tensor(z["batch"]).to(torch.int32).cuda(), cfg["B"], eps=torch.tensor(z["eps"]).cuda())
