"""Host-side mirror of the reference's denoiser classes and registry.

Same constructor signature, same ``state_dict`` keys/shapes, same ``forward_denoise`` contract as
``common/nets/model_conditional_diffusion_mixste_s2s_grand_linLift.py:139-257`` (seq2seq) and
``..._s2f_grand_linLift.py:139-266`` (seq2frame), selected by name like ``common/nets/load_net.py:5-10`` -- but the
modules below only *hold parameters*.  No layer here has a PyTorch forward: ``forward_denoise`` hands the tensors to
the HIP engine (libd3d_hip.so) and fails loudly when there is no HIP device.
"""
from __future__ import annotations

import os
import warnings
from contextlib import contextmanager
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib
from .engine import Engine
from .spec import DenoiserConfig, S2F_NAME, S2S_NAME, denoiser_param_spec


class _Holder(nn.Module):
    """Parameter container; calling it is a bug (the engine does the math)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("diff3dhpe_amd modules are parameter holders; use forward_denoise / GaussianDiffusion")


class _AttnParams(_Holder):
    def __init__(self, dim: int, qkv_bias: bool = True):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)     # S2S:67
        self.proj = nn.Linear(dim, dim)


class _MlpParams(_Holder):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class _BlockParams(_Holder):
    """Keys: norm1, attn.{qkv,proj}, norm2, time_mlp.1, mlp.{fc1,fc2} (reference Block, S2S:90-109)."""

    def __init__(self, dim: int, hidden: int, time_dim: Optional[int], qkv_bias: bool = True, norm_eps: float = 1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=norm_eps)
        self.attn = _AttnParams(dim, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=norm_eps)
        self.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_dim, dim)) if time_dim else None
        self.mlp = _MlpParams(dim, hidden)


class _MixSTEDenoiser(nn.Module):
    _seq2frame = False
    # engine arithmetic mode; override per instance (or on the class) at any time:
    #   "auto"  (default) the F16X3 engine, with its range guard READ after every call (see _guarded below): when a checkpoint / input
    #           leaves the range in which F16X3 is fp32-accurate, the call is repeated on a lazily created exact-fp32 engine, a one-time
    #           warning names the flag, and the model stays on the fp32 engine until its weights change (or reset_precision_fallback())
    #   "f16x3" fp32-accurate GEMMs / attention from 3 fp16 MFMAs on hi/lo operand splits; a set range flag RAISES D3DError
    #   "fp32"  exact fp32 MFMA everywhere
    #   "bf16"  second-class (bf16 operands; cannot meet the 1e-4 gate; BASELINE configs[1])
    # "auto", "f16x3" and "fp32" pass the 1e-4 parity gate.  The reference loads arbitrary checkpoints (RUN:226-235): results that
    # are silently not fp32-accurate are not an option, hence the guard is on by default (D3D_CHECK_RANGE=0 or range_check = False
    # turns the read off: one ~2 us kernel + one event wait per call).
    precision = "auto"
    range_check = os.environ.get("D3D_CHECK_RANGE", "1").strip().lower() not in ("0", "false", "no", "off")
    # One process per GPU is the supported multi-GPU form (torchrun; parallel.py).  nn.DataParallel over SEVERAL devices in one
    # process (RUN:216-218 with --gpu_id 0,1,...) would need one engine per device driven from DataParallel's worker threads:
    # that path has never run on hardware, so it fails loudly instead of being trusted.  Flip this (or export D3D_ALLOW_MULTI_DEVICE=1
    # for the unchanged runner with --gpu_id 0,1,...) to try it anyway.
    allow_multi_device = os.environ.get("D3D_ALLOW_MULTI_DEVICE", "0").strip().lower() in ("1", "true", "yes", "on")

    def __init__(self, num_frame=9, num_joints=17, in_chans=2, embed_dim=32, depth=4, num_heads=8, mlp_ratio=2.,
                 qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2, norm_layer=None,
                 with_time_emb=True, **kwargs):
        super().__init__()
        # The three arguments the runner never changes (RUN:179) but the reference accepts (S2S:140-142, 184):
        #   qkv_bias=False   no ".attn.qkv.bias" parameters (S2S:67); the engine is handed zero vectors
        #   qk_scale         overrides head_dim ** -0.5 (S2S:65); the engine's attention scale is fixed (it is folded into the q planes as
        #                    a power of two), so the q rows of every qkv weight / bias are handed over multiplied by qk_scale / head_dim ** -0.5
        #   norm_layer       a LayerNorm factory, e.g. partial(nn.LayerNorm, eps=...): its eps goes to norm1 / norm2 of every block and to
        #                    the two post-norms (S2S:95, 101, 202, 214); anything that is not an affine nn.LayerNorm is refused
        self._qkv_bias = bool(qkv_bias)
        self._q_rescale = 1.0 if qk_scale is None or not qk_scale else float(qk_scale) / float((embed_dim // num_heads) ** -0.5)
        norm_eps = 1e-6
        if norm_layer is not None:
            probe = norm_layer(embed_dim)
            if not isinstance(probe, nn.LayerNorm) or not probe.elementwise_affine or tuple(probe.normalized_shape) != (embed_dim,):
                raise NotImplementedError("norm_layer must build an affine nn.LayerNorm over the embedding (e.g. partial(nn.LayerNorm, eps=1e-5)); "
                                          f"got {type(probe).__name__}")
            norm_eps = float(probe.eps)
            if not 0.0 < norm_eps < 1.0:
                raise ValueError(f"norm_layer eps must lie in (0, 1), got {norm_eps}")
        self._norm_eps = norm_eps
        # drop_rate / attn_drop_rate / drop_path_rate only act in training mode (S2S:123-128), which is out of scope.
        self.cfg = DenoiserConfig(num_frame=num_frame, num_joints=num_joints, in_chans=in_chans, embed_dim=embed_dim,
                                  depth=depth, num_heads=num_heads, mlp_ratio=mlp_ratio, with_time_emb=bool(with_time_emb),
                                  seq2frame=self._seq2frame)
        D, Dm = embed_dim, self.cfg.mlp_hidden
        time_dim = 2 * D if with_time_emb else None
        # index 0 of the reference Sequential is SinusoidalPosEmb (no parameters); keep the indices 1 and 3
        self.time_mlp = nn.Sequential(_Holder(), nn.Linear(D, time_dim), nn.GELU(), nn.Linear(time_dim, time_dim)) \
            if with_time_emb else None
        self.fusion_layer = nn.Linear(3 + in_chans, D)
        self.block_depth = depth
        self.Spatial_pos_embed = nn.Parameter(torch.zeros(1, num_joints, D))
        self.STEblocks = nn.ModuleList([_BlockParams(D, Dm, time_dim, qkv_bias, norm_eps) for _ in range(depth)])
        self.Spatial_norm = nn.LayerNorm(D, eps=norm_eps)
        self.Temporal_pos_embed = nn.Parameter(torch.zeros(1, num_frame, D))
        self.TTEblocks = nn.ModuleList([_BlockParams(D, Dm, time_dim, qkv_bias, norm_eps) for _ in range(depth)])
        self.Temporal_norm = nn.LayerNorm(D, eps=norm_eps)
        if self._seq2frame:
            self.weighted_mean = nn.Conv1d(in_channels=num_frame, out_channels=1, kernel_size=1)
        self.head = nn.Sequential(nn.LayerNorm(D), nn.Linear(D, 3))
        # engines are shared (by reference) with DataParallel replicas, keyed by device index
        self._engines: Dict[int, Engine] = {}
        self._engine_sig: Dict[int, object] = {}
        self._engines_fb: Dict[int, Engine] = {}       # "auto": the exact-fp32 engines the guard falls back to, made on first need
        self._engine_sig_fb: Dict[int, object] = {}
        # guard state, one dict shared (by reference) with DataParallel replicas: "fallback" = weights signature for which the guard
        # fired ("auto" then runs the fp32 engine directly), "warned", "deferred" = open list of Pending checks (evaluate.py), "stats"
        self._guard = {"fallback": None, "warned": False, "deferred": None, "posted": 0, "flagged": 0, "reruns": 0}
        self._src_sig = None

    # ------------------------------------------------------------------ engine management
    def _tensor(self, name: str) -> torch.Tensor:
        # attribute walk instead of state_dict(): DataParallel replicas carry plain tensors, not nn.Parameters
        obj = self
        for part in name.split("."):
            obj = obj[int(part)] if part.isdigit() else getattr(obj, part)
        return obj

    def _named_tensors(self):
        """The tensors the engine expects, by reference state-dict name: the parameters themselves, except what the constructor arguments
        qkv_bias=False / qk_scale turn into derived tensors (zero bias vectors; q rows multiplied by qk_scale / head_dim ** -0.5)."""
        D = self.cfg.embed_dim
        out = {}
        for name, _, _, _ in denoiser_param_spec(self.cfg):
            if name.endswith(".attn.qkv.bias") and not self._qkv_bias:
                t = torch.zeros(3 * D, dtype=torch.float32)
            else:
                t = self._tensor(name)
            if self._q_rescale != 1.0 and (name.endswith(".attn.qkv.weight") or name.endswith(".attn.qkv.bias")):
                t = t.detach().to(torch.float32).clone()
                t[:D] *= self._q_rescale
            out[name] = t
        return out

    def _param_signature(self):
        # (over self.parameters(): 0.3 ms for the 240 tensors -- the attribute walk of _named_tensors() is 1.4 ms, per call, and shows in
        # the latency of small batches; a DataParallel replica has no Parameters and is never asked: it carries the source's _src_sig)
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _replicate_for_data_parallel(self):
        # nn.DataParallel re-broadcasts the parameters on every call (RUN:217); remember the source tensors' identity
        # so a replica's engine re-uploads weights only when the source module's weights actually changed.
        replica = super()._replicate_for_data_parallel()
        replica._src_sig = self._param_signature()
        return replica

    def _primary_precision(self) -> str:
        return "f16x3" if self.precision == "auto" else self.precision

    def engine_for(self, device: torch.device, fallback: bool = False) -> Engine:
        """The engine of this model on `device`; fallback=True: the exact-fp32 engine precision="auto" repeats a flagged call on."""
        if device.type != "cuda":
            raise _lib.D3DError("engine_for() needs a HIP device")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if not self.allow_multi_device and self._engines and idx not in self._engines:
            p = self._tensor("fusion_layer.weight")
            moved = self._src_sig is None and p.is_cuda and p.device.index == idx
            if moved:   # the module itself was moved (.to('cuda:1') after running on cuda:0): one device at a time -- release the
                self._engines.clear()         # stale engine (its weights, tables and workspace live on the old device)
                self._engine_sig.clear()
                self._engines_fb.clear()
                self._engine_sig_fb.clear()
            else:
                raise _lib.D3DError(
                    f"this model already runs on cuda:{next(iter(self._engines))} and is now asked to run on cuda:{idx}: diff3dhpe_amd "
                    "supports ONE device per process (launch one process per GPU: `python bench.py --gpus N`, torchrun --nproc-per-node "
                    "N, or pass one id to --gpu_id); nn.DataParallel over several devices in one process is untested -- set "
                    "allow_multi_device = True on the model class (or D3D_ALLOW_MULTI_DEVICE=1) to try it (moving the whole module "
                    "with .to() is fine)")
        engines, sigs = (self._engines_fb, self._engine_sig_fb) if fallback else (self._engines, self._engine_sig)
        want = "fp32" if fallback else self._primary_precision()
        eng = engines.get(idx)
        if eng is None or eng.precision != want:
            eng = Engine(self.cfg, precision=want, device=torch.device("cuda", idx))
            if self._norm_eps != 1e-6:
                import struct
                eng.set_option("norm_eps_bits", struct.unpack("<I", struct.pack("<f", self._norm_eps))[0])
            engines[idx] = eng
            sigs[idx] = None
        sig = self._src_sig if self._src_sig is not None else self._param_signature()
        if sigs.get(idx) != sig:
            eng.load_weights(self._named_tensors())
            sigs[idx] = sig
        return eng

    # ------------------------------------------------------------------ the range guard, read by default
    def reset_precision_fallback(self) -> None:
        """precision="auto": go back to the F16X3 engine after the guard moved this model to the fp32 engine."""
        self._guard["fallback"] = None

    def _weights_sig(self):
        return self._src_sig if self._src_sig is not None else self._param_signature()

    def _on_fallback(self) -> bool:
        return self.precision == "auto" and self._guard["fallback"] is not None and self._guard["fallback"] == self._weights_sig()

    def _flagged(self, flags: int, what: str) -> bool:
        """A range flag was read for a call of this model: "auto" moves the model to the fp32 engine (True: the caller repeats the
        call); every other precision raises.  False: nothing to repeat (only the head kernel's self-repaired recompute bit)."""
        if flags & _lib.RANGE_RECOMPUTE:        # repaired inside the kernel: a fault indicator, not a precision matter -- say so, once
            self._guard["recomputes"] = self._guard.get("recomputes", 0) + 1
            if not self._guard.get("warned_recompute"):
                self._guard["warned_recompute"] = True
                warnings.warn(f"diff3dhpe_amd: {what}: " + Engine.describe_range_flags(_lib.RANGE_RECOMPUTE), RuntimeWarning, stacklevel=4)
            flags &= ~_lib.RANGE_RECOMPUTE
            if not flags:
                return False
        if flags & _lib.RANGE_INDEX:            # not a precision matter either: no engine computes a row whose timestep is outside the tables
            raise IndexError(f"{what}: " + Engine.describe_range_flags(_lib.RANGE_INDEX) + " (the row's output is NaN; DIFF:21-24)")
        self._guard["flagged"] += 1
        msg = Engine.describe_range_flags(flags)
        if self.precision != "auto":
            raise _lib.D3DError(f"{what}: F16X3 operand range exceeded by {msg}: the result is not fp32-accurate for this checkpoint / "
                                "input -- use precision='auto' (repeats such calls on the exact-fp32 engine) or precision='fp32'")
        self._guard["fallback"] = self._weights_sig()
        if not self._guard["warned"]:
            self._guard["warned"] = True
            warnings.warn(f"diff3dhpe_amd: {what}: the F16X3 range guard fired ({msg}); precision='auto' repeats the call on the "
                          "exact-fp32 engine and keeps this model there until its weights change (reset_precision_fallback() to retry)",
                          RuntimeWarning, stacklevel=4)
        return True

    # tickets of one engine that a deferred block may leave open: the engine keeps 256 slots (d3d_engine_range_post recycles the oldest)
    _MAX_OPEN_TICKETS = 192

    def _lazy_read(self, eng, what: str) -> None:
        """Guard read for an engine on which no precision flag can rise (fp32, bf16, the "auto" fallback): only the fault indicators
        travel in its word -- the head fence's D3D_RANGE_RECOMPUTE and the bound check of the timestep gathers, D3D_RANGE_INDEX.  The
        ticket is posted and the OPEN ones are polled without waiting, so the host does not synchronise with every call; a
        raised bit is therefore reported at the next guarded call at the latest (the public q_sample / p_losses entry points check
        their timesteps on the host beforehand: engine.py _timesteps)."""
        lazy = self._guard.setdefault("lazy", [])
        lazy.append((eng, eng.post_range(), what))
        self._guard["posted"] += 1
        flags, first = 0, None
        keep = []
        for i, (e, t, w) in enumerate(lazy):
            # earlier calls' tickets: their work is behind this call's on the stream and long done -- waiting on them is free and makes
            # "at the next guarded call at the latest" true; this call's own ticket is only polled
            f = e.take_range(t, block=i + 1 < len(lazy))
            if f is None:
                keep.append((e, t, w))
            elif f:
                flags |= f
                first = first or w
        self._guard["lazy"] = keep
        if flags & (_lib.RANGE_RECOMPUTE | _lib.RANGE_INDEX):
            self._flagged(flags & (_lib.RANGE_RECOMPUTE | _lib.RANGE_INDEX), first)      # warns once / raises IndexError

    def flush_range_checks(self) -> None:
        """Read (waiting on each one's own event) every guard ticket still open from calls on an engine that is read without waiting
        (fp32, bf16, the "auto" fallback): a fault bit raised by an earlier call is reported now instead of at the next call."""
        lazy, self._guard["lazy"] = self._guard.get("lazy", []), []
        flags, first = 0, None
        for e, t, w in lazy:
            f = e.take_range(t, block=True)
            if f:
                flags |= f
                first = first or w
        if flags & (_lib.RANGE_RECOMPUTE | _lib.RANGE_INDEX):
            self._flagged(flags & (_lib.RANGE_RECOMPUTE | _lib.RANGE_INDEX), first)

    def _guarded(self, get_engine, fn, what: str):
        """Run fn(engine) on the model's engine and READ the F16X3 range guard for it (no device-wide synchronisation: one one-lane
        kernel behind the call, one wait on ITS event).  get_engine(fallback) -> a ready engine.  With an open deferred list
        (deferred_range_checks(): evaluate() resolves it at its own per-batch synchronisation) the read is postponed.  Engines on which
        no precision flag can rise (fp32, bf16, the fallback) are read without waiting (_lazy_read)."""
        fb_sig = self._guard["fallback"]
        if fb_sig is not None and fb_sig != self._weights_sig():      # new weights: back to the F16X3 engine, and a new fallback warns again
            self._guard["fallback"] = None
            self._guard["warned"] = False
        if self._on_fallback():
            eng = get_engine(True)
            res = fn(eng)
            if self.range_check:
                self._lazy_read(eng, what)
            return res
        eng = get_engine(False)
        res = fn(eng)
        if not self.range_check:
            return res
        if eng.precision != "f16x3":
            self._lazy_read(eng, what)
            return res
        ticket = eng.post_range()
        self._guard["posted"] += 1
        pend = self._guard["deferred"]
        if pend is not None:
            pend.append((eng, ticket, what))
            if len(pend) >= self._MAX_OPEN_TICKETS:      # the engine's ticket ring is finite: fold the oldest half into the box now
                self._guard["deferred_box"].fold(len(pend) // 2)
            return res
        flags = eng.take_range(ticket, block=True)
        if not flags:
            return res
        if not self._flagged(flags, what):      # raises unless "auto"
            return res
        self._guard["reruns"] += 1
        if self._guard["fallback"] is not None:
            # the F16X3 engine's workspace (34 GB at B = 512, T = 243) is dead weight while the model runs on the fp32 engine: release it so
            # that the repeat -- which allocates the fp32 engine's own -- does not double the peak; it comes back lazily
            eng._ws, eng._ws_B = None, 0
        return fn(get_engine(True))

    @contextmanager
    def deferred_range_checks(self):
        """Inside: guarded calls only POST their range tickets.  The yielded object's resolve() -- call it behind a synchronisation
        the caller makes anyway -- reads them all: False = every call was in range; True = precision="auto" has moved the model to
        its fp32 engine and the caller must repeat the work of the block; other precisions raise D3DError."""
        outer, outer_box = self._guard["deferred"], self._guard.get("deferred_box")
        box = _Deferred(self)
        self._guard["deferred"], self._guard["deferred_box"] = box.items, box
        try:
            yield box
        finally:
            self._guard["deferred"], self._guard["deferred_box"] = outer, outer_box

    def _compute_device(self, *tensors) -> torch.device:
        for t in tensors:
            if isinstance(t, torch.Tensor) and t.is_cuda:
                return t.device
        p = self._tensor("fusion_layer.weight")
        if p.is_cuda:
            return p.device
        if torch.cuda.is_available():   # CPU tensors in, HIP compute: still the engine, never a CPU fallback
            return torch.device("cuda", torch.cuda.current_device())
        raise _lib.D3DError("no HIP device available: diff3dhpe_amd has no CPU execution path")

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def forward_denoise(self, x: torch.Tensor, time: torch.Tensor) -> torch.Tensor:
        """x: (B,T,J,in_chans+3) = cat([2D pose, noisy 3D pose], -1); time: (B,) long|float -> (B,T,J,3) [(B,1,J,3) S2F]."""
        dev = self._compute_device(x)
        c = self.cfg.in_chans
        out = self._guarded(lambda fb: self.engine_for(dev, fb),
                            lambda eng: eng.denoise(x[..., :c], x[..., c:], time if self.cfg.with_time_emb else None), "forward_denoise")
        return out.to(x.device)

    def forward(self, x, time):
        return self.forward_denoise(x, time)


class _Deferred:
    """Range tickets posted inside _MixSTEDenoiser.deferred_range_checks()."""

    def __init__(self, net: _MixSTEDenoiser):
        self.net = net
        self.items = []
        self._flags, self._what = 0, None      # what fold() has already read

    def fold(self, n: int) -> None:
        """Read the n oldest tickets now (blocking on each one's own event) and remember their flags: an engine keeps a finite ring of
        ticket slots, a block with more guarded calls than that must not let the oldest expire."""
        for eng, ticket, w in self.items[:n]:
            f = eng.take_range(ticket, block=True)
            if f:
                self._flags |= f
                self._what = self._what or w
        del self.items[:n]

    def resolve(self) -> bool:
        flags, what = self._flags, self._what
        self._flags, self._what = 0, None
        for eng, ticket, w in self.items:
            f = eng.take_range(ticket, block=True)
            if f:
                flags |= f
                what = what or w
        self.items.clear()
        if not flags:
            return False
        if not self.net._flagged(flags, what):  # raises unless "auto"
            return False
        self.net._guard["reruns"] += 1
        return True


class ConditionalDiffusionMixSTES2SGRANDLinLift(_MixSTEDenoiser):
    _seq2frame = False


class ConditionalDiffusionMixSTES2FGRANDLinLift(_MixSTEDenoiser):
    _seq2frame = True


def HPE_model(MODEL_NAME: str):
    """Registry by class name (reference common/nets/load_net.py:5-10); unknown names raise KeyError like the reference."""
    models = {
        S2S_NAME: ConditionalDiffusionMixSTES2SGRANDLinLift,
        S2F_NAME: ConditionalDiffusionMixSTES2FGRANDLinLift,
    }
    return models[MODEL_NAME]
