"""CPU-side checks: the C-ABI library loads and exports every symbol include/d3d.h declares, its host-only entry
points (integer DDIM schedule, engine bookkeeping, argument validation) behave, the host mirrors expose the
reference's API surface, and the product path fails loudly without a HIP device.  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import gold, ROOT
import diff3dhpe_amd as d3d
from diff3dhpe_amd import _lib
from diff3dhpe_amd.spec import DenoiserConfig, denoiser_param_spec, param_count
from diff3dhpe_amd.synth import synth_state_dict, hash_uniform, synth_inputs

NO_GPU = not torch.cuda.is_available()


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "d3d.h")).read()
    declared = sorted(set(re.findall(r"\b(d3d_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for sym in declared:
        assert hasattr(L, sym), f"libd3d_hip.so does not export {sym}"
    assert sorted(_lib.ABI_SYMBOLS) == declared
    assert L.d3d_version() >= 100


def test_allgather_entry_rejects_bad_arguments_without_touching_rccl():
    """d3d_allgather_pred (RUN:216-218 exchange for non-torch hosts): argument checks come first, no collective is attempted."""
    L = _lib.lib()
    assert L.d3d_allgather_pred(None, None, None, 0, None) == -1       # D3D_EINVAL
    assert b"bad argument" in L.d3d_last_error()


def test_ddim_times_bit_exact_for_every_S():
    g = gold("ddim_times_N1000")
    flat, offs = g["flat"], g["offsets"]
    for S in range(1, 1001):
        assert d3d.ddim_times(1000, S) == flat[offs[S - 1]:offs[S]].tolist(), S
    assert d3d.ddim_times(1000, 5) == [999, 799, 599, 399, 199, -1]
    assert d3d.ddim_times(1000, 9)[0] == 999 and d3d.ddim_times(1000, 9)[-1] == -1
    for key in ("N100_S20", "N100_S100", "N50_S7", "N200_S33"):
        N, S = (int(x[1:]) for x in key.split("_"))
        assert d3d.ddim_times(N, S) == g[key].tolist()
    with pytest.raises(d3d.D3DError):
        d3d.ddim_times(0, 5)


def _create(cfg, precision=0):
    c = _lib.Config(cfg.num_frame, cfg.num_joints, cfg.in_chans, cfg.embed_dim, cfg.depth, cfg.num_heads, cfg.mlp_hidden,
                    int(cfg.with_time_emb), int(cfg.seq2frame), precision)
    h = C.c_void_p()
    rc = _lib.lib().d3d_engine_create(C.byref(c), C.byref(h))
    return rc, h


@pytest.mark.parametrize("cfg", [DenoiserConfig(num_frame=81, embed_dim=512, depth=8),
                                 DenoiserConfig(num_frame=27, embed_dim=512, depth=8, seq2frame=True),
                                 DenoiserConfig(num_frame=27, embed_dim=32, depth=4, with_time_emb=False)])
def test_engine_weight_inventory_matches_reference_state_dict(cfg):
    rc, h = _create(cfg)
    assert rc == 0
    L = _lib.lib()
    names = []
    for i in range(L.d3d_engine_num_weights(h)):
        name, n = C.c_char_p(), C.c_int64()
        assert L.d3d_engine_weight_info(h, i, C.byref(name), C.byref(n)) == 0
        names.append((name.value.decode(), n.value))
    spec = [(n, int(np.prod(s))) for n, s, _, _ in denoiser_param_spec(cfg)]
    assert names == spec
    # argument validation on the host side
    a = np.zeros(7, np.float32)
    assert L.d3d_engine_set_weight(h, b"no.such.weight", a.ctypes.data_as(C.c_void_p), 7) == -1
    assert L.d3d_engine_set_weight(h, b"fusion_layer.bias", a.ctypes.data_as(C.c_void_p), 7) == -1
    assert b"size mismatch" in L.d3d_last_error()
    assert L.d3d_engine_commit_weights(h) == -2          # weights missing -> D3D_ESTATE
    assert b"missing weight" in L.d3d_last_error()
    L.d3d_engine_destroy(h)


def test_engine_rejects_unsupported_configs():
    assert _create(DenoiserConfig(num_frame=9, embed_dim=48, depth=1, num_heads=8))[0] == -5      # D % 32
    assert _create(DenoiserConfig(num_frame=9, embed_dim=32, depth=1, num_heads=5))[0] == -1      # D % H
    assert _create(DenoiserConfig(num_frame=9, embed_dim=32, depth=1), precision=7)[0] == -5


@pytest.mark.skipif(not NO_GPU, reason="checks the no-device failure mode")
def test_product_path_fails_loudly_without_a_device():
    cfg = DenoiserConfig(num_frame=9, embed_dim=32, depth=1)
    rc, h = _create(cfg)
    L = _lib.lib()
    for name, arr in synth_state_dict(cfg, 0).items():
        assert L.d3d_engine_set_weight(h, name.encode(), arr.ctypes.data_as(C.c_void_p), arr.size) == 0
    assert L.d3d_engine_commit_weights(h) == -3          # D3D_EHIP: no CPU path
    L.d3d_engine_destroy(h)
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=9, embed_dim=32, depth=1)
    diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=3).eval()
    with pytest.raises(d3d.D3DError):
        diff(clean_3d_pose=torch.zeros(1, 9, 17, 3), noisy_2d_pose=torch.zeros(1, 9, 17, 2), output_loss=False)
    with pytest.raises(d3d.D3DError):
        net.forward_denoise(torch.zeros(1, 9, 17, 5), torch.zeros(1, dtype=torch.long))


@pytest.mark.skipif(not NO_GPU, reason="checks the no-device failure mode")
def test_machine_probe_needs_a_device_and_checks_its_arguments():
    L = _lib.lib()
    r = C.c_float(0.0)
    assert L.d3d_probe_machine(0, 10.0, C.byref(r), None) == -3          # D3D_EHIP: nothing is estimated on the host
    assert L.d3d_probe_machine(5, 10.0, C.byref(r), None) == -1
    assert L.d3d_probe_machine(1, -1.0, C.byref(r), None) == -1


def test_product_package_never_imports_the_oracle():
    import subprocess, sys
    code = "import sys; import diff3dhpe_amd, diff3dhpe_amd.parallel, diff3dhpe_amd.evaluate; " \
           "bad=[m for m in sys.modules if m.startswith('oracle')]; assert not bad, bad"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
    for fn in os.listdir(os.path.join(ROOT, "diff3dhpe_amd")):
        if fn.endswith(".py"):
            src = open(os.path.join(ROOT, "diff3dhpe_amd", fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_registry_and_ctor_surface():
    assert d3d.HPE_model("ConditionalDiffusionMixSTES2SGRANDLinLift") is d3d.ConditionalDiffusionMixSTES2SGRANDLinLift
    assert d3d.HPE_model("ConditionalDiffusionMixSTES2FGRANDLinLift") is d3d.ConditionalDiffusionMixSTES2FGRANDLinLift
    with pytest.raises(KeyError):
        d3d.HPE_model("nope")
    # runner call at RUN:178-180
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=81, num_joints=17, in_chans=2, embed_dim=512, depth=8, num_heads=8,
                                      mlp_ratio=2., qkv_bias=True, qk_scale=None, drop_path_rate=0.1, with_time_emb=True)
    assert sum(p.numel() for p in net.parameters()) == 43674115 == param_count(net.cfg)
    assert sum(p.numel() for p in d3d.HPE_model(d3d.S2F_NAME)(num_frame=27, embed_dim=512, depth=8).parameters()) == 43646495
    assert sum(p.numel() for p in d3d.HPE_model(d3d.S2S_NAME)(num_frame=243, embed_dim=512, depth=8).parameters()) == 43757059
    assert sum(p.numel() for p in d3d.HPE_model(d3d.S2S_NAME)().parameters()) == 94883 - (81 - 9) * 32   # class defaults, T=9
    assert (net.Spatial_pos_embed == 0).all() and (net.Temporal_pos_embed == 0).all()      # zeros at init (S2S:193,205)
    with pytest.raises(ValueError):
        d3d.GaussianDiffusion(model=net, beta_schedule="quadratic")
    with pytest.raises(AssertionError):
        d3d.GaussianDiffusion(model=net, timesteps=10, sampling_timesteps=20)
    with pytest.raises(TypeError):
        d3d.GaussianDiffusion(model=torch.nn.Linear(2, 2))


def test_ctor_accepts_the_arguments_the_reference_accepts():
    """S2S:140-142, 184: qkv_bias=False (no qkv bias parameters: S2S:67), a qk_scale override (S2S:65) and a LayerNorm factory with
    another eps are constructor arguments of the reference; the module mirrors their effect on the parameter set and hands the engine
    derived tensors (zero bias vectors, rescaled q rows).  Anything that is not an affine LayerNorm over the embedding is refused."""
    from functools import partial
    kw = dict(num_frame=9, embed_dim=64, depth=1, num_heads=8)
    plain = d3d.HPE_model(d3d.S2S_NAME)(**kw)
    nb = d3d.HPE_model(d3d.S2S_NAME)(qkv_bias=False, **kw)
    assert {k for k in plain.state_dict()} - {k for k in nb.state_dict()} == {"STEblocks.0.attn.qkv.bias", "TTEblocks.0.attn.qkv.bias"}
    t = nb._named_tensors()
    assert t["STEblocks.0.attn.qkv.bias"].shape == (192,) and not t["STEblocks.0.attn.qkv.bias"].any()
    qs = d3d.HPE_model(d3d.S2S_NAME)(qk_scale=0.25, **kw)                 # head_dim 8: default scale 8 ** -0.5
    qs.load_state_dict(plain.state_dict())
    r = 0.25 / 8 ** -0.5
    a, b = qs._named_tensors(), plain._named_tensors()
    assert torch.equal(a["TTEblocks.0.attn.qkv.weight"][:64], b["TTEblocks.0.attn.qkv.weight"][:64] * r)
    assert torch.equal(a["TTEblocks.0.attn.qkv.weight"][64:], b["TTEblocks.0.attn.qkv.weight"][64:])
    assert torch.equal(a["TTEblocks.0.attn.qkv.bias"][:64], b["TTEblocks.0.attn.qkv.bias"][:64] * r)
    assert torch.equal(qs.STEblocks[0].attn.qkv.weight, plain.STEblocks[0].attn.qkv.weight)        # the parameters themselves are untouched
    ne = d3d.HPE_model(d3d.S2S_NAME)(norm_layer=partial(torch.nn.LayerNorm, eps=1e-3), **kw)
    assert ne.Spatial_norm.eps == ne.Temporal_norm.eps == ne.STEblocks[0].norm1.eps == ne.TTEblocks[0].norm2.eps == 1e-3
    assert ne.head[0].eps == 1e-5 and plain.Spatial_norm.eps == 1e-6                              # the head's LayerNorm is not norm_layer (S2S:218)
    with pytest.raises(NotImplementedError):
        d3d.HPE_model(d3d.S2S_NAME)(norm_layer=torch.nn.BatchNorm1d, **kw)
    with pytest.raises(NotImplementedError):
        d3d.HPE_model(d3d.S2S_NAME)(norm_layer=partial(torch.nn.LayerNorm, elementwise_affine=False), **kw)


def test_state_dict_layout_and_checkpoint_loading():
    cfg = DenoiserConfig(num_frame=27, embed_dim=32, depth=2)
    net = d3d.HPE_model(d3d.S2S_NAME)(num_frame=27, embed_dim=32, depth=2)
    diff = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=5, loss_type="l2", clip_denoised=True)
    sd = diff.state_dict()
    bufs = [k for k in sd if not k.startswith("model.")]
    assert bufs == ['betas', 'alphas_cumprod', 'alphas_cumprod_prev', 'sqrt_recip_alphas', 'sqrt_alphas_cumprod',
                    'sqrt_one_minus_alphas_cumprod', 'log_one_minus_alphas_cumprod', 'sqrt_recip_alphas_cumprod',
                    'sqrt_recipm1_alphas_cumprod', 'posterior_variance', 'posterior_log_variance_clipped',
                    'posterior_mean_coef1', 'posterior_mean_coef2', 'p2_loss_weight']
    assert {k[len("model."):] for k in sd if k.startswith("model.")} == {n for n, _, _, _ in denoiser_param_spec(cfg)}
    # reference checkpoints: DataParallel 'module.' prefix, loader drops keys containing 'alphas' (RUN:226-235)
    ck = {"module." + k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 9, prefix="model.").items()}
    ck["module.alphas_cumprod"] = torch.zeros(1000)
    filtered = {k: v for k, v in ck.items() if "alphas" not in k}
    res = diff.load_state_dict({k[len("module."):]: v for k, v in filtered.items()}, strict=False)
    assert not res.unexpected_keys and all("model." not in k for k in res.missing_keys)
    assert torch.equal(net.fusion_layer.weight.detach(), ck["module.model.fusion_layer.weight"])
    g = gold("schedules")
    for k in bufs:
        assert np.array_equal(sd[k].numpy(), g["cosine/" + k]), k
    for sched in ("linear", "logcosine"):
        d = d3d.GaussianDiffusion(model=net, timesteps=1000, sampling_timesteps=9, beta_schedule=sched)
        for k in ("betas", "alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "sqrt_alphas_cumprod"):
            assert np.array_equal(d.state_dict()[k].numpy(), g[f"{sched}/{k}"]), (sched, k)
    assert diff.ddim_times() == [999, 799, 599, 399, 199, -1]
    assert diff.sqrt_alphas_cumprod_prev.dtype == torch.float64 and diff.sqrt_alphas_cumprod_prev.shape == (1001,)


def test_reference_checkpoint_file_roundtrip(tmp_path):
    """A file in the reference's on-disk layout (RUN:451-460: metadata + 'model_diffusion' with DataParallel 'module.'
    keys) loads the way the reference's loader does it (RUN:226-235): 'alphas' tables are ignored, the rest lands in the
    engine's parameters; save_checkpoint writes that layout back."""
    from diff3dhpe_amd.checkpoint import load_checkpoint, save_checkpoint, reference_state_dict
    cfg = DenoiserConfig(num_frame=27, embed_dim=32, depth=2)
    mk = lambda: d3d.GaussianDiffusion(model=d3d.HPE_model(d3d.S2S_NAME)(num_frame=27, embed_dim=32, depth=2), timesteps=1000,
                                       sampling_timesteps=5, loss_type="l2", clip_denoised=True)
    src = mk()
    src.model.load_state_dict({k: torch.from_numpy(v) for k, v in synth_state_dict(cfg, 4).items()})
    path = str(tmp_path / "epoch_7.bin")
    save_checkpoint(src, path, epoch=7, lr=1e-4)
    blob = torch.load(path, map_location="cpu", weights_only=False)
    assert set(blob) == {"epoch", "best_epoch", "min_loss", "min_train_loss", "lr", "random_state", "optimizer", "model_diffusion"}
    assert all(k.startswith("module.") for k in blob["model_diffusion"])
    blob["model_diffusion"]["module.alphas_cumprod"] = torch.full((1000,), 123.0)     # a stale table must not be loaded
    blob["model_diffusion"]["module.sqrt_alphas_cumprod"] = torch.full((1000,), 123.0)
    torch.save(blob, path)
    dst = mk()
    meta = load_checkpoint(dst, path)
    assert meta["epoch"] == 7 and not meta["unexpected_keys"]
    assert all("alphas" in k for k in meta["missing_keys"])
    assert float(dst.alphas_cumprod.max()) < 1.0 and float(dst.sqrt_alphas_cumprod.max()) <= 1.0
    for (ka, a), (kb, b) in zip(src.model.state_dict().items(), dst.model.state_dict().items()):
        assert ka == kb and torch.equal(a, b), ka
    assert "alphas_cumprod" not in reference_state_dict(blob) and "model.fusion_layer.weight" in reference_state_dict(blob)


def test_synth_is_deterministic_and_well_scaled():
    a = hash_uniform("x", 1000, 3)
    assert np.array_equal(a, hash_uniform("x", 1000, 3)) and not np.array_equal(a, hash_uniform("x", 1000, 4))
    assert -1 <= a.min() < -0.9 and 0.9 < a.max() < 1 and abs(a.mean()) < 0.1
    assert abs(float(hash_uniform("probe", 4, 0)[0]) - float(hash_uniform("probe", 1, 0)[0])) == 0.0
    i1, i2 = synth_inputs(2, 9, seed=5), synth_inputs(2, 9, seed=5)
    assert all(np.array_equal(i1[k], i2[k]) for k in i1)
    assert np.abs(i1["x2d"]).max() <= 1 and np.abs(i1["gt3d"][:, :, 0]).max() == 0


def _device_isa(obj_name, allow_none=False):
    """Disassembly of the gfx950 code object inside one of the library's object files (diff3dhpe_amd/build/<obj_name>); with allow_none
    an object file without device code (host-only source) gives ''."""
    import shutil, subprocess, tempfile
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    src = os.path.join(ROOT, "diff3dhpe_amd", "build", obj_name)
    # no skip: this image has hipcc and llvm-objdump on the build container AND on the GPU box; a missing object file is built
    assert os.path.exists(objdump), "llvm-objdump of the ROCm toolchain not found: the ISA pins cannot be checked"
    if not os.path.exists(src):
        from diff3dhpe_amd.build import build
        build(verbose=False)
    assert os.path.exists(src), src
    with tempfile.TemporaryDirectory() as tmp:
        o = os.path.join(tmp, obj_name)
        shutil.copy(src, o)
        subprocess.run([objdump, "--offloading", o], check=True, capture_output=True, cwd=tmp)
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        if allow_none and not co:
            return ""
        assert co, os.listdir(tmp)
        return subprocess.run([objdump, "-d", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout


def test_no_kernel_uses_the_packed_fp32_form_that_deviates_beside_mfma():
    """gfx950 deviation found in round 6 (experiments/probes/pk_beside_mfma*.hip; experiments/NOTES.md section 000): a
    v_pk_{add,mul,fma}_f32 whose SRC1 low half selects the HIGH dword of its register pair -- `op_sel:[x,1]` / `op_sel:[x,1,y]`, what the
    compiler emits when it folds the broadcast of a value that sits in an odd register -- now and then computes the low result of lanes
    48-63 with src1 = 0 while the other wave of its SIMD STARTS issuing MFMAs after the matrix pipe has been idle (1.5e-5 per
    wave-instruction beside MFMA bursts in the probe and in a de-phased GEMM epilogue; a continuous MFMA stream only hits at its start).  It is what made k_head deviate on a GPU shared by two processes (rounds 1-2).  No kernel of the library may contain it:
    run-time scalars meet packed arithmetic through splat2_rt (x3q_epilogue_acc.h), the row kernels are built without the SLP
    vectoriser (build.py EXTRA_FLAGS)."""
    import re
    from diff3dhpe_amd.build import SOURCES
    bad = []
    for src in SOURCES:
        cur = None
        for line in _device_isa(src.replace(".hip", ".o"), allow_none=True).splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = m.group(1)
                continue
            t = line.strip()
            if not (t.startswith("v_pk_") and "_f32" in t.split()[0]):
                continue
            sel = re.search(r"op_sel:\[([01,]+)\]", t)
            if sel and len(sel.group(1).split(",")) >= 2 and sel.group(1).split(",")[1] == "1":
                bad.append((src, cur, t.split("//")[0].strip()))
    assert not bad, bad[:10]


def test_head_kernel_instruction_stream_is_the_one_that_is_stable_on_a_shared_gpu():
    """k_head's 3-row dot product must be compiled as the stream that never deviated when two processes shared a GPU (0 of ~900
    traced samplings; the compiler's free schedule deviated in 1 launch of 60, mechanism unidentified: experiments/NOTES.md):
      (a) ONE weight fragment loaded, waited for (vmcnt(0)) and consumed at a time -- never several in flight behind counted waits;
      (b) the three running sums o[0], o[1], o[2] live in separate registers and are updated by scalar v_add_f32: between a
          fragment's load and the next there is no v_pk_add_f32 and no v_pk_* with an op_sel / op_sel_hi modifier (the deviating
          stream kept o[0] / o[1] in one register pair, updated by v_pk_fma_f32 with op_sel).  Packed multiplies INSIDE one sum
          (v_pk_mul_f32 / v_pk_fma_f32 on the x,z / y,w halves of one fragment, no op_sel) are part of the kept stream.
    A toolchain update that re-schedules the loop fails HERE, on the build box, instead of silently changing results on a shared
    GPU; the run-time cross-check is tests/test_gpu_round4.py::test_two_ranks_on_one_device_repeat."""
    import re
    isa = _device_isa("kernels_elem.o")
    for nv, fence in ((1, 1), (2, 1), (4, 1), (1, 0), (2, 0), (4, 0)):   # "head_fence" on (three evaluations) / off (one: the default)
        m = re.search(r"<_ZN3d3d6k_headILi%dELb%dEEEvNS_8HeadArgsE>:\n(.*?)(\n\n|\Z)" % (nv, fence), isa, re.S)
        assert m, f"k_head<{nv}, {fence}> not found"
        ops, text = [], []
        for line in m.group(1).splitlines():
            t = line.strip().split("//")[0].strip()
            text.append(t)
            if t.startswith("global_load_dwordx4"):
                ops.append("L")
            elif t.startswith("s_waitcnt") and "vmcnt(" in t:
                ops.append("W" + re.search(r"vmcnt\((\d+)\)", t).group(1))
            elif t.startswith("global_load") or t.startswith("global_store") or t.startswith("buffer_"):
                ops.append("M")
            else:
                ops.append(None)
        # The kernel evaluates the dot products up to three times (round 5: run-time fence -- twice, and a third time on a mismatch); the
        # last 16-byte loads of the kernel are their weight fragments: all 3 x 6 of them at D = 512 (NV = 2, the production width, where
        # the two-process experiments ran); at the other widths the compiler sinks some LayerNorm-vector loads between the FIRST
        # evaluation's fragments, so of that one only the last fragment's three are identified by position.
        n_tail = (18 if nv == 2 else 6 * nv + 3) if fence else (6 if nv == 2 else 3)
        idx = [i for i, o in enumerate(ops) if o == "L"][-n_tail:]
        assert len(idx) == n_tail, (nv, len(idx))
        for n, i in enumerate(idx):
            dpp = next((j for j in range(i, len(text)) if "_dpp" in text[j]), len(text))   # the wave reduction behind an evaluation
            end = min(idx[n + 1], dpp) if n + 1 < len(idx) else dpp
            seg_ops = [o for o in ops[i + 1:end] if o]
            assert "W0" in seg_ops, (nv, seg_ops)                                        # (a) waited for before anything else loads
            assert not any(o == "L" for o in seg_ops)
            if nv != 2:      # (b) is checked at the production width: at the others LayerNorm arithmetic (packed) sits between the loads
                continue
            seg = text[i + 1:end]
            packed = [t for t in seg if t.startswith("v_pk_")]
            assert not any(t.startswith("v_pk_add_f32") or "op_sel" in t for t in packed), (nv, packed)   # (b)
            assert any(t.startswith("v_add_f32") for t in seg), (nv, seg)                # the sum's own scalar update
