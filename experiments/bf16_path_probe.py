"""Which path of a block separates the bf16 engine from the oracle's bf16 emulation?  depth-1 model with the attention branch
(proj = 0) and / or the MLP branch (fc2 = 0) switched off."""
import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import inputs, torch_sd
import diff3dhpe_amd as d3d
from diff3dhpe_amd.spec import DenoiserConfig
from oracle import d3d_oracle as orc
T = int(sys.argv[1]) if len(sys.argv) > 1 else 27
cfg = DenoiserConfig(num_frame=T, embed_dim=512, depth=1)
inp = inputs(2, T, 910)
xcat = torch.cat([inp["x2d"], inp["noise"] * 0.7], dim=-1)
t = torch.tensor([77, 508])
for name, kill in (("LN only", ("attn.proj", "mlp.fc2")), ("attention only", ("mlp.fc2",)), ("MLP only", ("attn.proj",)), ("spatial attention only", ("mlp.fc2", "TTEblocks.0.attn.proj")),
                   ("temporal attention only", ("mlp.fc2", "STEblocks.0.attn.proj")), ("all", ())):
    sd = torch_sd(cfg, 91)
    for k in sd:
        if any(s in k for s in kill):
            sd[k] = torch.zeros_like(sd[k])
    with orc.operand_rounding(torch.bfloat16):
        emu = orc.forward_denoise(sd, xcat, t, depth=1)
    f32 = orc.forward_denoise(sd, xcat, t, depth=1)
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=T, embed_dim=512, depth=1)
    net.load_state_dict(sd)
    net.precision = "bf16"
    out = net.cuda().forward_denoise(xcat.cuda(), t.cuda()).cpu()
    d = lambda a, b: (a.double() - b.double()).abs().max().item()
    print(f"{name:24s}: bf16 engine vs emulation {d(out, emu):.3e} | emulation vs fp32 {d(emu, f32):.3e}")
