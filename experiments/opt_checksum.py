"""Checksum of one sampling under an engine option on / off (bit-identity of a switchable kernel form):
   python experiments/opt_checksum.py x3_proj_rows 1 0"""
import hashlib, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import cfg_full, inputs, build_product
key, vals = sys.argv[1], [int(v) for v in sys.argv[2:]]
_, diff = build_product(cfg_full(243), 5, sampling=1, precision="f16x3")
eng = diff._engine(torch.device("cuda", 0))
inp = inputs(40, 243, 78)
for v in vals:
    eng.set_option(key, v)
    y = eng.ddim_sample(inp["x2d"].cuda(), inp["noise"].cuda())
    print(key, v, hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16], bool(torch.isfinite(y).all()))
