"""fc2 ring kernel against the template form: mismatch statistics (debugging aid).  python experiments/fc2_ring_check.py [T B streams delay reps]"""
import sys
import torch
sys.path[:0] = [".", "tests"]
from helpers import cfg_full, inputs
from test_gpu_round4 import _product

T, B, streams, delay, reps, dbg = (int(x) for x in (sys.argv[1:7] + ["243", "33", "1", "24", "3", "0"][len(sys.argv) - 1:]))
cfg = cfg_full(T)
_, diff = _product(cfg, 5, "f16x3", sampling=1, family="uniform")
eng = diff._engine(torch.device("cuda", 0))
eng.set_option("streams", streams)
inp = inputs(B, T, 82)
x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
eng.set_option("fc2_ring", 0)
plain = eng.ddim_sample(x2d, nz).clone()
eng.set_option("fc2_ring", 1)
eng.set_option("fc2_ring_delay", delay)
eng.set_option("fc2_ring_dbg", dbg)
for r in range(reps):
    eng.range_flags(clear=True)
    own = eng.ddim_sample(x2d, nz).clone()
    d = (own - plain).abs()
    bad = (own != plain)
    rows = bad.reshape(-1, 3).any(dim=1).nonzero().flatten()
    print(f"dbg {dbg} rep {r}: flags {eng.range_flags()} mismatching values {int(bad.sum())} of {bad.numel()}, max diff {float(d.max()):.3e}, "
          f"rows {rows.numel()} first {rows[:8].tolist()} last {rows[-4:].tolist()}")
