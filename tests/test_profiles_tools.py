"""profiles/summarize.py names every product kernel in the rocprofv3 summaries (a kernel whose row came out unnamed once hid two GEMMs
from the round-4 roofline table), and profiles/roofline_table.py prices each of them."""
import importlib.util
import os

from conftest import ROOT


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "profiles", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


KERNELS = {
    "void d3d::(anonymous namespace)::k_fc1_x3(d3d::(anonymous namespace)::F1Args)": "(fc1)",
    "void d3d::(anonymous namespace)::k_proj_x3(d3d::(anonymous namespace)::PjArgs)": "(proj)",
    "void d3d::(anonymous namespace)::k_qkv_sattn(d3d::(anonymous namespace)::QsArgs)": "(spatial blocks)",
    "void d3d::(anonymous namespace)::k_qkv_tattn<false>(d3d::(anonymous namespace)::QtArgs)": "the frames of one joint",
    "void d3d::(anonymous namespace)::k_qkv_tattn<true>(d3d::(anonymous namespace)::QtArgs)": "255 / T joints",
    "void d3d::(anonymous namespace)::k_linear_x3q_persist<8, 1, 8, 2, 2, 10>(_Float16 const*, _Float16 const*)": "fc2 + post-norm",
    "void d3d::(anonymous namespace)::k_linear_x3q_persist<8, 2, 4, 0, 1, 1>(_Float16 const*, _Float16 const*)": "(qkv)",
    "void d3d::k_head<2>(d3d::HeadArgs)": "k_head<2>",
    "void d3d::(anonymous namespace)::k_gemm_bf16q<0>(d3d::(anonymous namespace)::BqArgs)": "(qkv)",
    "void d3d::(anonymous namespace)::k_gemm_bf16q<1>(d3d::(anonymous namespace)::BqArgs)": "(fc1)",
}


def test_every_product_kernel_gets_a_name_and_a_price():
    summ = _load("summarize")
    for raw, part in KERNELS.items():
        name = summ.short(raw)
        assert name.strip() and part in name, (raw, name)
    src = open(os.path.join(ROOT, "profiles", "roofline_table.py")).read()
    for key in ("k_qkv_sattn", "k_qkv_tattn", "k_fc1_x3", "k_proj_x3", "k_gemm_bf16q", "k_linear", "k_head", "k_embed"):
        assert key in src
