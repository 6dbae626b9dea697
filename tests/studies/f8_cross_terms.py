"""Numerics study (CPU, not collected by pytest): could the two CROSS terms of the F16X3 product run on the fp8 matrix pipe?

F16X3 computes a w as a_hi w_hi + a_lo w_hi + a_hi w_lo with fp16 hi / lo halves: three fp16 MFMAs per product.  The cross terms are
2^-11 of the product, so a few bits of them would do -- and `v_mfma_f32_16x16x128_f8f6f4` moves 4x the k-depth of the fp16 MFMA in
2x its time: both cross terms of 64 k-elements in one fp8 MFMA would cut the matrix-pipe time per product from 3 units to 2.
This script runs the oracle's denoiser / DDIM loop (the reference's op sequence) with the block GEMMs and the two attention
products replaced by emulations of
    x3      : hi hi + lo hi + hi lo, 11-bit halves                       (what the engine computes)
    x1f8e4  : hi hi + r4(lo) r4(hi) + r4(hi) r4(lo), 4 significant bits  (e4m3 cross terms, per-tensor power-of-two scale = exact)
    x1f8e3  : the same with 3 significant bits                           (e5m2 cross terms)
    x1      : hi hi alone                                                (fp16 operands, one MFMA)
all with exact (float64) accumulation, against the fp32 oracle -- the gate is 1e-4 max-abs (north_star).

    python tests/studies/f8_cross_terms.py            (about a minute on 8 cores)
Result on file: experiments/NOTES.md section 0.13."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import d3d_oracle as orc          # noqa: E402  (test infrastructure: this file lives under tests/)
from helpers import cfg_full, inputs           # noqa: E402
from diff3dhpe_amd.synth import synth_state_dict   # noqa: E402


def rbits(x, bits):
    """x rounded to `bits` significant bits (round to nearest even), unbounded exponent."""
    m, e = torch.frexp(x.double())
    return torch.ldexp(torch.round(m * (1 << bits)) / (1 << bits), e)


def split(x):
    hi = rbits(x, 11)
    return hi, rbits(x.double() - hi, 11)


MODE = "x3"


def product(a, b_t):
    """a (.., M, K) @ b_t (.., K, N) under MODE; float64 accumulation, fp32 result."""
    ah, al = split(a)
    bh, bl = split(b_t)
    out = ah @ bh
    if MODE == "x3":
        out = out + al @ bh + ah @ bl
    elif MODE in ("x1f8e4", "x1f8e3"):
        nb = 4 if MODE == "x1f8e4" else 3
        out = out + rbits(al, nb) @ rbits(bh, nb) + rbits(ah, nb) @ rbits(bl, nb)
    elif MODE != "x1":
        raise ValueError(MODE)
    return out.float()


def block_linear(x, w, b):
    y = product(x, w.t())
    return y if b is None else y + b


def grand_attention(sd, p, x, heads):
    G, N, C = x.shape
    dh = C // heads
    qkv = block_linear(x, sd[p + ".qkv.weight"], sd.get(p + ".qkv.bias")).reshape(G, N, 3, heads, dh).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    a = product(q, k.transpose(-2, -1)) * (dh ** -0.5)
    a = a.softmax(dim=-1)
    eye = torch.eye(N, dtype=a.dtype).view(1, 1, N, N)
    o = product(a - eye, v).transpose(1, 2).reshape(G, N, C)
    return block_linear(o, sd[p + ".proj.weight"], sd[p + ".proj.bias"])


def main():
    global MODE
    torch.manual_seed(0)
    torch.set_num_threads(8)
    T, B = 27, 2
    cfg = cfg_full(T)
    tabs = orc.diffusion_tables("cosine", 1000)
    plain_lin, plain_att = orc._block_linear, orc.grand_attention
    for family, seed in (("uniform", 5), ("trainedlike", 11)):
        sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, seed, family=family).items()}
        inp = inputs(B, T, 700)
        xcat = torch.cat([inp["x2d"], inp["noise"] * 0.7], dim=-1)
        t = torch.tensor([905, 17], dtype=torch.long)
        ref_f = orc.forward_denoise(sd, xcat, t, depth=cfg.depth)
        ref_s = orc.ddim_sample_loop(sd, tabs, inp["x2d"], inp["noise"], depth=cfg.depth, num_timesteps=1000, sampling_timesteps=9)
        ref_s = ref_s[0] if isinstance(ref_s, (tuple, list)) else ref_s
        for mode in ("x3", "x1f8e4", "x1f8e3", "x1"):
            MODE = mode
            orc._block_linear, orc.grand_attention = block_linear, grand_attention
            try:
                f = orc.forward_denoise(sd, xcat, t, depth=cfg.depth)
                s = orc.ddim_sample_loop(sd, tabs, inp["x2d"], inp["noise"], depth=cfg.depth, num_timesteps=1000, sampling_timesteps=9)
                s = s[0] if isinstance(s, (tuple, list)) else s
            finally:
                orc._block_linear, orc.grand_attention = plain_lin, plain_att
            print(f"{family:12s} {mode:7s}  one denoiser evaluation: max-abs {float((f - ref_f).abs().max()):.3e}   "
                  f"9-step DDIM sampling: max-abs {float((s - ref_s).abs().max()):.3e}", flush=True)


if __name__ == "__main__":
    main()
