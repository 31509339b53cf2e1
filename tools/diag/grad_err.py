"""rel-L2 of every gradient of the C2 model on the window path (autograd entry points) against the float64 oracle,
at B = 4096 and 20000 -- the quantities tests/test_gpu_parity.py::test_window_full_batch_sizes bounds by 1e-5."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "st-dadk_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
from golden import cases
from oracle import stdadk_oracle as orc
import test_gpu_parity as T
sizes = [(int(a), 98) for a in sys.argv[1:]] or [(4096, 99), (20000, 98)]
for B, seed in sizes:
    cfg = dict(cases.MODEL_CASES["c2_b257"], B=B, seed=seed)
    X, coords, t, y = cases.make_inputs(cfg)
    d = T.dev()
    m = T.build_model(cfg)
    m.train()
    params = cases.make_state(cfg)
    yp = m(*(torch.from_numpy(a).to(d) for a in (X, coords, t)))
    loss = torch.nn.MSELoss()(yp, torch.from_numpy(y).to(d))
    loss.backward()
    yo, lo, go = orc.train_step_grads(X, coords, t, y, params, cfg)
    print(f"B={B} KROT={os.environ.get('STDADK_KROT','1')} y {np.abs(yp.detach().cpu().numpy() - yo).max():.2e} loss {abs(loss.item()-lo)/lo:.2e} " +
          " ".join(f"{k}:{T.rel_l2(p.grad.cpu().numpy(), go[k]):.2e}" for k, p in m.named_parameters()), flush=True)
