"""Round-6 evidence on the MI355X, through the product API / the C ABI:
  * fc2 + post-norm on the barrier-free k-loop (kernels_fc2_ring.hip: wave-private W slots, A through a four-slot ring with arrival
    counters in LDS) against the token GEMM's post-norm form, bit for bit."""
from functools import partial

import pytest
import torch

import diff3dhpe_amd as d3d
from conftest import gold
from diff3dhpe_amd.spec import DenoiserConfig
from diff3dhpe_amd.synth import synth_state_dict
from helpers import cfg_full, inputs, maxabs
from test_gpu_round4 import _product

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,B,family,streams,delay", [(243, 33, "trainedlike", 1, 24), (243, 32, "uniform", 1, 0), (243, 64, "uniform", 2, 24),
                                                      (81, 97, "uniform", 1, 7)])
def test_fc2_ring_kernel_is_bit_identical_to_the_template_form(T, B, family, streams, delay):
    """"fc2_ring" (default): fc2 + post-norm on its own kernel without workgroup barriers in the k-loop (128 x 512 whole-row tiles, the
    template's own post-norm epilogue function, the ragged last tile by its checked form): bit for bit the token GEMM's post-norm form,
    whatever the start delay of waves 4-7.  Token counts that are and are not multiples of 128, one and two streams, NaN-filled
    workspace; no poll of an arrival counter times out (range-guard bit 16)."""
    cfg = cfg_full(T)
    _, diff = _product(cfg, 11 if family == "trainedlike" else 5, "f16x3", sampling=2, family=family)
    eng = diff._engine(torch.device("cuda", torch.cuda.current_device()))
    eng.set_option("streams", streams)
    inp = inputs(B, T, 82)
    x2d, nz = inp["x2d"].cuda(), inp["noise"].cuda()
    eng.set_option("fc2_ring", 0)
    plain = eng.ddim_sample(x2d, nz).clone()
    eng.set_option("fc2_ring", 1)
    eng.set_option("fc2_ring_delay", delay)
    try:
        eng._workspace(B).view(torch.float32).fill_(float("nan"))
        eng.range_flags(clear=True)
        own = eng.ddim_sample(x2d, nz).clone()
        assert eng.range_flags() == 0
        assert torch.isfinite(own).all()
        assert torch.equal(own, plain)
    finally:
        eng.set_option("fc2_ring_delay", 24)


CTOR = {"nobias": (dict(qkv_bias=False), "uniform"), "qkscale": (dict(qk_scale=0.2), "uniform"),
        "eps": (dict(norm_layer=partial(torch.nn.LayerNorm, eps=1e-3)), "trainedlike")}


@pytest.mark.parametrize("prec", ["fp32", "f16x3"])
@pytest.mark.parametrize("tag", list(CTOR))
def test_ctor_arguments_golden(tag, prec):
    """qkv_bias=False, qk_scale=0.2 and norm_layer=partial(nn.LayerNorm, eps=1e-3) (S2S:140-142, 184; the engine refused them until round
    6) against raw denoiser outputs of the imported reference (oracle/gen_golden.py gen_round6: T = 27, D = 512, depth 2; the oracle equal
    to the reference bit for bit while generating), in both gated precisions, range guard silent."""
    g = gold("denoise_ctor_args_T27")
    kw, family = CTOR[tag]
    cfg = DenoiserConfig(num_frame=27, embed_dim=512, depth=2)
    args = dict(num_frame=27, num_joints=17, in_chans=2, embed_dim=512, depth=2, num_heads=8, mlp_ratio=2., qkv_bias=True, qk_scale=None,
                drop_path_rate=0.1, with_time_emb=True)
    args.update(kw)
    net = d3d.HPE_model(d3d.S2S_NAME)(**args)
    sd = {k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, int(g["seed"]), family=family).items()}
    if tag == "nobias":
        sd = {k: v for k, v in sd.items() if not k.endswith("attn.qkv.bias")}
    net.load_state_dict(sd, strict=True)
    net.precision = prec
    net = net.cuda()
    inp = inputs(2, 27, int(g["input_seed"]))
    xcat = torch.cat([inp["x2d"], inp["noise"] * float(g["y_scale"])], dim=-1).cuda()
    worst = 0.0
    for t in (999, 17):
        out = net.forward_denoise(xcat, torch.full((2,), t, dtype=torch.long, device="cuda"))
        worst = max(worst, maxabs(out, g[f"{tag}_t{t}"]))
    assert worst < 1e-4, (tag, prec, worst)
    eng = net.engine_for(torch.device("cuda", torch.cuda.current_device()))
    assert eng.range_flags() == 0
