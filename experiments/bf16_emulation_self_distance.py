"""How far apart are two CORRECT bf16-operand computations that differ only in fp32-level arithmetic?  The oracle's bf16-operand
emulation with fp32 accumulation against the same emulation with fp64 accumulation (identical rounding points, operands and
weights), beside the distance of either to the unrounded fp32 oracle.  CPU only."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import inputs, torch_sd
from diff3dhpe_amd.spec import DenoiserConfig
from oracle import d3d_oracle as orc
torch.set_num_threads(8)
mp = lambda a, b: (a.double() - b.double()).norm(dim=-1).mean().item()
mx = lambda a, b: (a.double() - b.double()).abs().max().item()
for depth in (1, 2, 4, 8):
    cfg = DenoiserConfig(num_frame=27, embed_dim=512, depth=depth)
    sd = torch_sd(cfg, 91)
    inp = inputs(2, 27, 910)
    xcat = torch.cat([inp["x2d"], inp["noise"] * 0.7], dim=-1)
    t = torch.tensor([77, 508])
    f32 = orc.forward_denoise(sd, xcat, t, depth=depth)
    with orc.operand_rounding(torch.bfloat16):
        e32 = orc.forward_denoise(sd, xcat, t, depth=depth)
    torch.set_default_dtype(torch.float64)
    with orc.operand_rounding(torch.bfloat16):
        e64 = orc.forward_denoise({k: v.double() for k, v in sd.items()}, xcat.double(), t, depth=depth)
    torch.set_default_dtype(torch.float32)
    print(f"depth {depth}: emulation(fp32 acc) vs emulation(fp64 acc): max-abs {mx(e32, e64):.3e} MPJPE {mp(e32, e64):.3e} | "
          f"emulation vs fp32 oracle: max-abs {mx(e32, f32):.3e} MPJPE {mp(e32, f32):.3e}")
