"""Checksum of one sampling (T=27 / 243, two weight families) with the in-tree library: run it under two builds of the library to see
whether a change kept the results bit for bit (experiments/ab_libs.sh swaps the file)."""
import hashlib, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from helpers import cfg_full, inputs, build_product
for T, B in ((27, 5), (243, 3)):
    _, diff = build_product(cfg_full(T), 5, sampling=2, precision="f16x3")
    inp = inputs(B, T, 78)
    y = diff._engine(torch.device("cuda", 0)).ddim_sample(inp["x2d"].cuda(), inp["noise"].cuda())
    print(T, B, hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16])
