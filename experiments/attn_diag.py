"""Step timing of the staggered temporal attention kernel (k_attn_temporal_x3s) from in-kernel clock stamps.
Needs a library built with -DD3D_ATTN_DIAG_BUILD:
    bash experiments/build_variant.sh adiag "-DD3D_ATTN_DIAG_BUILD" kernels_attn_x3
    cp experiments/_libs/libd3d_adiag.so diff3dhpe_amd/libd3d_hip.so        (on the GPU box; restore afterwards)
    python experiments/attn_diag.py [B]          (switches the "attn_diag" option of d3d_engine_set_option on for the last launch)
Prints, for workgroups 0, 3, 6 and both wave halves, per unit: cycles spent waiting at each of the three step barriers and
cycles of work behind each (score / softmax / PV step of that half)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import ctypes as C
from diff3dhpe_amd import _lib
from diff3dhpe_amd.engine import op_attention

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T, J, D, H = 243, 17, 512, 8
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B * T * J, 3 * D, generator=g).cuda()
for _ in range(3):
    op_attention(qkv, B, T, J, H, True, "f16x3")
torch.cuda.synchronize()
_lib.check(_lib.lib().d3d_engine_set_option(None, b"attn_diag", 1))
op_attention(qkv, B, T, J, H, True, "f16x3")
