"""Round-6 evidence on the MI355X, through the product API / the C ABI:
  * fc2 + post-norm on the barrier-free k-loop (kernels_fc2_ring.hip: wave-private W slots, A through a four-slot ring with arrival
    counters in LDS) against the token GEMM's post-norm form, bit for bit."""
import pytest
import torch

from helpers import cfg_full, inputs
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
