"""bf16 engine vs the oracle's bf16-operand emulation vs the fp32 oracle, by model depth (raw denoiser output, T=27, B=2)."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import inputs, build_product, torch_sd
from diff3dhpe_amd.spec import DenoiserConfig
from oracle import d3d_oracle as orc
for depth in (1, 2, 4, 8):
    cfg = DenoiserConfig(num_frame=27, embed_dim=512, depth=depth)
    sd = torch_sd(cfg, 91)
    inp = inputs(2, 27, 910)
    xcat = torch.cat([inp["x2d"], inp["noise"] * 0.7], dim=-1)
    t = torch.tensor([77, 508])
    with orc.operand_rounding(torch.bfloat16):
        emu = orc.forward_denoise(sd, xcat, t, depth=depth)
    f32 = orc.forward_denoise(sd, xcat, t, depth=depth)
    res = {}
    for prec in ("bf16", "f16x3"):
        net, _ = build_product(cfg, 91, sampling=3, precision=prec)
        res[prec] = net.forward_denoise(xcat.cuda(), t.cuda()).cpu()
    d = lambda a, b: (a.double() - b.double()).abs().max().item()
    print(f"depth {depth}: bf16 engine vs emulation {d(res['bf16'], emu):.3e} | emulation vs fp32 {d(emu, f32):.3e} | bf16 engine vs fp32 {d(res['bf16'], f32):.3e} | f16x3 engine vs fp32 {d(res['f16x3'], f32):.3e}")
