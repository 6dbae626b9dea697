"""Python handle on one libd3d_hip engine (one per device).  PyTorch is used only as plumbing: device buffers,
the current HIP stream and host<->device copies.  All arithmetic happens inside the library's HIP kernels."""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Mapping, Optional

import numpy as np
import torch

from . import _lib
from .spec import DenoiserConfig, denoiser_param_spec


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _f32c(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class Engine:
    def __init__(self, cfg: DenoiserConfig, precision: str = "fp32", device=None):
        if not torch.cuda.is_available():
            raise _lib.D3DError("diff3dhpe_amd needs a HIP device: the engine has no CPU path")
        self.cfg = cfg
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.precision = precision
        c = _lib.Config(cfg.num_frame, cfg.num_joints, cfg.in_chans, cfg.embed_dim, cfg.depth, cfg.num_heads,
                        cfg.mlp_hidden, int(cfg.with_time_emb), int(cfg.seq2frame), _lib.PRECISIONS[precision])
        h = C.c_void_p()
        _lib.check(_lib.lib().d3d_engine_create(C.byref(c), C.byref(h)))
        self._h = h
        self._ws: Optional[torch.Tensor] = None
        self._ws_B = 0
        self.sampling_timesteps = None
        self.num_timesteps = None
        self.weights_version = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _lib.lib().d3d_engine_destroy(h)
            except Exception:
                pass
            self._h = None

    # ---------------------------------------------------------------- weights / schedule
    def expected_weights(self):
        L = _lib.lib()
        out = []
        for i in range(L.d3d_engine_num_weights(self._h)):
            name, n = C.c_char_p(), C.c_int64()
            _lib.check(L.d3d_engine_weight_info(self._h, i, C.byref(name), C.byref(n)))
            out.append((name.value.decode(), n.value))
        return out

    def load_weights(self, sd: Mapping[str, object]) -> None:
        """sd: denoiser tensors keyed by reference state-dict names WITHOUT the 'model.' prefix."""
        L = _lib.lib()
        for name, numel in self.expected_weights():
            if name not in sd:
                raise KeyError(f"missing weight {name}")
            v = sd[name]
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            _lib.check(L.d3d_engine_set_weight(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.size))
        if self.cfg.with_time_emb:
            half = self.cfg.embed_dim // 2
            # SinusoidalPosEmb's frequency table exactly as the host framework forms it (S2S:31-33)
            freqs = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).to(torch.float32).numpy()
            _lib.check(L.d3d_engine_set_time_freqs(self._h, freqs.ctypes.data_as(C.c_void_p), half))
        with torch.cuda.device(self.device):
            _lib.check(L.d3d_engine_commit_weights(self._h))
        self.sampling_timesteps = None

    def set_schedule(self, alphas_cumprod: torch.Tensor, sqrt_one_minus_alphas_cumprod: torch.Tensor,
                     sampling_timesteps: int, eta: float, clip_denoised: bool,
                     sqrt_alphas_cumprod: Optional[torch.Tensor] = None) -> None:
        ac = np.ascontiguousarray(alphas_cumprod.detach().cpu().numpy(), dtype=np.float32)
        so = np.ascontiguousarray(sqrt_one_minus_alphas_cumprod.detach().cpu().numpy(), dtype=np.float32)
        with torch.cuda.device(self.device):
            st = torch.cuda.current_stream(self.device)
            _lib.check(_lib.lib().d3d_engine_set_schedule(self._h, ac.size, ac.ctypes.data_as(C.c_void_p),
                                                         so.ctypes.data_as(C.c_void_p), int(sampling_timesteps),
                                                         float(eta), int(bool(clip_denoised)), C.c_void_p(st.cuda_stream)))
            if sqrt_alphas_cumprod is not None:
                sa = np.ascontiguousarray(sqrt_alphas_cumprod.detach().cpu().numpy(), dtype=np.float32)
                _lib.check(_lib.lib().d3d_engine_set_sqrt_alphas_cumprod(self._h, sa.ctypes.data_as(C.c_void_p), sa.size))
        self.sampling_timesteps = int(sampling_timesteps)
        self.num_timesteps = int(ac.size)
        self.eta = float(eta)

    # ---------------------------------------------------------------- compute
    def _workspace(self, B: int) -> torch.Tensor:
        if self._ws is None or B > self._ws_B:
            nbytes = _lib.lib().d3d_workspace_bytes(self._h, B)
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            self._ws_B = B
        return self._ws

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _out_frames(self) -> int:
        return 1 if self.cfg.seq2frame else self.cfg.num_frame

    def denoise(self, x2d: torch.Tensor, y: torch.Tensor, time: Optional[torch.Tensor]) -> torch.Tensor:
        """forward_denoise on cat([x2d, y], -1): x2d (B,T,J,in), y (B,T,J,3) or (B,1,J,3) (broadcast over T), time (B,) or (1,)."""
        cfg = self.cfg
        B = x2d.shape[0]
        assert tuple(x2d.shape) == (B, cfg.num_frame, cfg.num_joints, cfg.in_chans), x2d.shape
        assert y.shape[0] == B and y.shape[1] in (1, cfg.num_frame) and tuple(y.shape[2:]) == (cfg.num_joints, 3), y.shape
        x2d, y = _f32c(x2d, self.device), _f32c(y, self.device)
        n_t, tdev = 0, None
        if cfg.with_time_emb:
            tdev = _f32c(time.reshape(-1), self.device)
            n_t = tdev.numel()
        out = torch.empty((B, self._out_frames(), cfg.num_joints, 3), dtype=torch.float32, device=self.device)
        if B == 0:          # an empty shard (fewer windows than ranks): the reference returns an empty tensor, the C ABI takes B >= 1
            return out
        ws = self._workspace(B)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().d3d_denoise(self._h, _ptr(x2d), _ptr(y), int(y.shape[1]), _ptr(tdev), n_t, _ptr(out), B, _ptr(ws),
                                              ws.numel(), self._stream()))
        return out

    def ddim_sample(self, x2d: torch.Tensor, init_noise: torch.Tensor, step_noise: Optional[torch.Tensor] = None,
                    trajectory: bool = False):
        """The whole S-step DDIM loop. Returns y0, or (y0, x_reverse_diffusion, x_start_est) with trajectory=True."""
        cfg = self.cfg
        if self.sampling_timesteps is None:
            raise _lib.D3DError("set_schedule() must be called before ddim_sample()")
        B, S = x2d.shape[0], self.sampling_timesteps
        Fo = self._out_frames()
        assert tuple(x2d.shape) == (B, cfg.num_frame, cfg.num_joints, cfg.in_chans), x2d.shape
        assert tuple(init_noise.shape) == (B, Fo, cfg.num_joints, 3), init_noise.shape
        x2d, init_noise = _f32c(x2d, self.device), _f32c(init_noise, self.device)
        if step_noise is not None:
            assert tuple(step_noise.shape) == (S, B, Fo, cfg.num_joints, 3), step_noise.shape
            step_noise = _f32c(step_noise, self.device)
        out = torch.empty((B, Fo, cfg.num_joints, 3), dtype=torch.float32, device=self.device)
        rev = x0s = None
        if trajectory:
            rev = torch.empty((B, Fo, cfg.num_joints, 3, S), dtype=torch.float32, device=self.device)
            x0s = torch.empty_like(rev)
        if B == 0:          # (as in denoise)
            return (out, rev, x0s) if trajectory else out
        ws = self._workspace(B)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().d3d_ddim_sample(self._h, _ptr(x2d), _ptr(init_noise), _ptr(step_noise), _ptr(out),
                                                  _ptr(rev), _ptr(x0s), B, _ptr(ws), ws.numel(), self._stream()))
        return (out, rev, x0s) if trajectory else out

    # ---------------------------------------------------------------- profiling (HIP events inside the library)
    def set_graph_mode(self, on: bool) -> None:
        """Replay the whole DDIM loop as one hipGraph per (B, workspace) (eta == 0, no trajectory capture)."""
        _lib.check(_lib.lib().d3d_engine_set_graph_mode(self._h, int(on)))

    def set_option(self, key: str, value: int) -> None:
        """Explicit engine switch (include/d3d.h: "fused_postnorm", "fold_layernorm", "streams"); the library reads no environment."""
        _lib.check(_lib.lib().d3d_engine_set_option(self._h, key.encode(), int(value)))

    def info(self, key: str) -> int:
        """Read-only engine facts (include/d3d.h d3d_engine_get_info): "graphs_cached", "graphs_captured", "streams", "device"."""
        v = C.c_int64(0)
        _lib.check(_lib.lib().d3d_engine_get_info(self._h, key.encode(), C.byref(v)))
        return int(v.value)

    def range_flags(self, clear: bool = True) -> int:
        """F16X3 range guard (include/d3d.h): _lib.RANGE_ACT | RANGE_WEIGHT | RANGE_STATS bits; synchronises the current stream.
        The flags belong to THIS engine (its own word of device memory), cleared on read."""
        f = C.c_uint32(0)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().d3d_engine_range_flags(self._h, C.byref(f), int(clear), self._stream()))
        return int(f.value)

    @staticmethod
    def describe_range_flags(f: int) -> str:
        what = [n for b, n in ((_lib.RANGE_ACT, "an activation (|x| > 8188)"),
                               (_lib.RANGE_WEIGHT, "a non-finite (or beyond ~1e9: unrepresentable) GEMM weight"),
                               (_lib.RANGE_STATS, "a LayerNorm input row with |mean| > 16 standard deviations (one-pass statistics)"),
                               (_lib.RANGE_INDEX, "a timestep index outside [0, num_timesteps)"),
                               (_lib.RANGE_RECOMPUTE, "the head kernel's two evaluations of a row disagreeing (repaired by a third; "
                                                      "a machine / toolchain fault indicator, experiments/NOTES.md section 1.4)")) if f & b]
        return " and ".join(what) if what else "nothing"

    def check_range(self) -> None:
        """Raise D3DError if the F16X3 range guard fired since the last check (use precision='fp32' then).  Synchronises the stream;
        the Python classes use post_range() / take_range() instead."""
        f = self.range_flags(clear=True)
        if f:
            raise _lib.D3DError("F16X3 operand range exceeded by " + self.describe_range_flags(f) +
                                ": results are not fp32-accurate for this checkpoint/input -- use precision='fp32'")

    def post_range(self) -> int:
        """Enqueue a snapshot (and reset) of this engine's range word behind everything on the current stream; returns its ticket
        (include/d3d.h d3d_engine_range_post).  No synchronisation."""
        t = C.c_int64(-1)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().d3d_engine_range_post(self._h, self._stream(), C.byref(t)))
        return int(t.value)

    def take_range(self, ticket: int, block: bool = True) -> Optional[int]:
        """Flags of a ticket; block=True waits for THAT snapshot's event only, block=False returns None while it has not run."""
        f, ready = C.c_uint32(0), C.c_int32(0)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().d3d_engine_range_take(self._h, int(ticket), int(bool(block)), C.byref(f), C.byref(ready)))
        return int(f.value) if ready.value else None

    def set_trace(self, capacity: int, views: int = 1) -> None:
        """Debug trace: checksum every buffer the block-flow kernels write (0 turns it off); views: see include/d3d.h."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().d3d_engine_set_trace(self._h, int(capacity), int(views)))
        self._trace_cap = int(capacity)

    def trace_read(self):
        """[(tag, checksum)] in launch order since the last read; tag fields: see include/d3d.h."""
        cap = getattr(self, "_trace_cap", 0)
        sums, tags, n = (C.c_uint64 * cap)(), (C.c_uint32 * cap)(), C.c_int32(0)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().d3d_engine_trace_read(self._h, sums, tags, cap, C.byref(n), self._stream()))
        return [(int(tags[i]), int(sums[i])) for i in range(n.value)]

    def set_profiling(self, on: bool) -> None:
        _lib.check(_lib.lib().d3d_engine_set_profiling(self._h, int(on)))

    def profile_reset(self) -> None:
        _lib.check(_lib.lib().d3d_engine_profile_reset(self._h))

    def profile_read(self) -> Dict[str, Dict[str, float]]:
        """Per kernel class: total event-timed ms, launches, algorithmic flops and bytes of the launches timed."""
        L = _lib.lib()
        out = {}
        for c in range(_lib.KC_COUNT):
            ms, n, fl, by = C.c_double(), C.c_int64(), C.c_double(), C.c_double()
            _lib.check(L.d3d_engine_profile_read(self._h, c, C.byref(ms), C.byref(n), C.byref(fl), C.byref(by)))
            out[L.d3d_kernel_class_name(c).decode()] = {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}
        return out

    def head(self, X: torch.Tensor) -> torch.Tensor:
        """Regression head (LayerNorm eps 1e-5 + Linear D -> 3, S2S:217-220) on (rows, D) rows: (rows, 3), raw."""
        X = _f32c(X, self.device).reshape(-1, self.cfg.embed_dim)
        out = torch.empty((X.shape[0], 3), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().d3d_op_head(self._h, _ptr(X), _ptr(out), X.shape[0], self._stream()))
        return out

    def time_embedding(self, times: torch.Tensor) -> torch.Tensor:
        """Per-block time vectors for the given timesteps: (n, 2*depth, D), execution order STE0, TTE0, STE1, ..."""
        cfg = self.cfg
        t = _f32c(times.reshape(-1), self.device)
        n = t.numel()
        out = torch.empty((n, 2 * cfg.depth, cfg.embed_dim), dtype=torch.float32, device=self.device)
        scratch = torch.empty(n * 5 * cfg.embed_dim, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().d3d_op_time_embedding(self._h, _ptr(t), n, _ptr(out), _ptr(scratch), self._stream()))
        return out

    def _timesteps(self, t: torch.Tensor, B: int, check: bool = True) -> torch.Tensor:
        """(B,) integer timesteps for the table gathers of q_sample / the p_losses tail, checked like the reference's `table[t]`
        (extract, DIFF:21-24: IndexError outside the table; a ONE-entry t broadcasts over the batch as there, any other short t is refused).
        Cold path: one tiny reduction."""
        t = t.detach().reshape(-1)
        if t.numel() == 1 and B > 1:            # extract() reshapes to (t.shape[0], 1, 1, 1): a one-entry t broadcasts over the batch there
            t = t.expand(B)
        if t.numel() != B:
            raise IndexError(f"timestep tensor has {t.numel()} entries for a batch of {B}")
        n = self.num_timesteps
        if B and n is not None and check:       # (check=False: t was drawn by p_losses itself -- in range by construction; the kernels
            lo, hi = (int(v) for v in torch.aminmax(t))      # (ONE reduction, one read-back)
            if lo < 0 or hi >= n:
                raise IndexError(f"timestep index out of range: [{lo}, {hi}] outside [0, {n})")   # still bound-check every gather)
        return t.to(device=self.device, dtype=torch.int32).contiguous()

    def q_sample(self, x_start: torch.Tensor, t: torch.Tensor, noise: torch.Tensor, check_t: bool = True) -> torch.Tensor:
        B = x_start.shape[0]
        xs, nz = _f32c(x_start, self.device), _f32c(noise, self.device)
        ti = self._timesteps(t, B, check_t)
        out = torch.empty_like(xs)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().d3d_q_sample(self._h, _ptr(xs), _ptr(nz), _ptr(ti), _ptr(out), B, xs.numel() // B,
                                               self._stream()))
        return out


    def weighted_loss(self, model_out: torch.Tensor, target: torch.Tensor, t: torch.Tensor, loss_type: str, clip_loss: bool,
                      check_t: bool = True) -> torch.Tensor:
        """p_losses tail (DIFF:411-418): loss_fn(model_out, target, 'none') * min(1 + ac[t] / sqrt(1 - ac)[t], 3 if clip_loss)."""
        B = target.shape[0]
        mo, tg = _f32c(model_out, self.device), _f32c(target, self.device)
        assert mo.shape == tg.shape, (mo.shape, tg.shape)
        ti = self._timesteps(t, B, check_t)
        out = torch.empty_like(tg)
        if B == 0:
            return out
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().d3d_weighted_loss(self._h, _ptr(mo), _ptr(tg), _ptr(ti), _ptr(out), B, tg.numel() // B,
                                                    {"l1": 1, "l2": 2}[loss_type], int(bool(clip_loss)), self._stream()))
        return out


def repeat_batch(x: torch.Tensor, repeat_n: int) -> torch.Tensor:
    """x.repeat(repeat_n, 1, ...) on the device (DIFF:433): (B, ...) -> (repeat_n * B, ...)."""
    if repeat_n == 1 or x.shape[0] == 0:
        return x
    dev = x.device
    xs = _f32c(x, dev)
    out = torch.empty((repeat_n * xs.shape[0],) + tuple(xs.shape[1:]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().d3d_repeat_batch(_ptr(xs), _ptr(out), xs.shape[0], xs.numel() // xs.shape[0], int(repeat_n), st))
    return out


def hypothesis_mean(pred: torch.Tensor, repeat_n: int) -> torch.Tensor:
    """torch.mean(pred.view(repeat_n, b, ...), dim=0) on the device (DIFF:441/448): (repeat_n * B, ...) -> (B, ...)."""
    if repeat_n == 1 or pred.shape[0] == 0:
        return pred
    dev = pred.device
    ps = _f32c(pred, dev)
    B = ps.shape[0] // repeat_n
    out = torch.empty((B,) + tuple(ps.shape[1:]), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().d3d_hypothesis_mean(_ptr(ps), _ptr(out), B, ps.numel() // ps.shape[0], int(repeat_n), st))
    return out


# ---------------------------------------------------------------------------------------- stand-alone op wrappers
def op_linear(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
              epi: str = "none", precision: str = "fp32") -> torch.Tensor:
    epi_id = {"none": 0, "gelu": 1, "residual": 2}[epi]
    M, K = A.shape
    N = W.shape[0]
    dev = A.device
    A, W = _f32c(A, dev), _f32c(W, dev)
    bias = _f32c(bias, dev) if bias is not None else None
    residual = _f32c(residual, dev) if residual is not None else None
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().d3d_op_linear(_ptr(A), _ptr(W), _ptr(bias), _ptr(residual), _ptr(out), M, N, K, epi_id,
                                            _lib.PRECISIONS[precision], st))
    return out


def op_linear_bench(A: torch.Tensor, W: torch.Tensor, bias=None, residual=None, epi: str = "none", precision: str = "fp32",
                    variant: int = 0, reps: int = 10):
    """Returns (C, mean ms per launch) -- GEMM micro-benchmark through the C ABI (events on the launch stream)."""
    epi_id = {"none": 0, "gelu": 1, "residual": 2}[epi]
    M, K = A.shape
    N = W.shape[0]
    dev = A.device
    A, W = _f32c(A, dev), _f32c(W, dev)
    bias = _f32c(bias, dev) if bias is not None else None
    residual = _f32c(residual, dev) if residual is not None else None
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    ms = C.c_float(0.0)
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().d3d_op_linear_bench(_ptr(A), _ptr(W), _ptr(bias), _ptr(residual), _ptr(out), M, N, K, epi_id,
                                                  _lib.PRECISIONS[precision], variant, reps, C.byref(ms), st))
    return out, ms.value


def probe_machine(device=None, ms_target: float = 100.0) -> dict:
    """What this device sustains for the two resources that co-limit the F16X3 GEMM k-loop (d3d_probe_machine, include/d3d.h):
    {"mfma_f16_tflops": fp16 MFMA work in register loops at the power-limited clock,
     "l2_to_lds_gbps": the k-loop's LDS-DMA staging stream alone, summed over the CUs}."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    out = {}
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        for what, key in ((0, "mfma_f16_tflops"), (1, "l2_to_lds_gbps")):
            r = C.c_float(0.0)
            _lib.check(_lib.lib().d3d_probe_machine(what, float(ms_target), C.byref(r), st))
            out[key] = r.value
    return out


def op_linear_postnorm(A: torch.Tensor, W: torch.Tensor, bias: torch.Tensor, residual: torch.Tensor, gamma: torch.Tensor,
                       beta: torch.Tensor, eps: float = 1e-6, pos: Optional[torch.Tensor] = None, pos_div: int = 1,
                       tvec: Optional[torch.Tensor] = None, rows_per_batch: int = 1, with_stats: bool = False, reps: int = 1):
    """LayerNorm(residual + A W^T + bias) [+ pos[(m // pos_div) % len(pos)]] [+ tvec[m // rows_per_batch] or the one tvec]
    through the F16X3 GEMM whose epilogue applies the norm (fc2 + post-norm of a block, S2S:131-135 + 236/245).
    Returns (Y, stats or None, mean ms per launch); stats = per-row (sum, sum of squares) of Y from the plane form."""
    M, K = A.shape
    N = W.shape[0]
    dev = A.device
    A, W, bias, residual = _f32c(A, dev), _f32c(W, dev), _f32c(bias, dev), _f32c(residual, dev)
    gamma, beta = _f32c(gamma, dev), _f32c(beta, dev)
    pos = _f32c(pos, dev) if pos is not None else None
    tvec = _f32c(tvec, dev) if tvec is not None else None
    stride = 0 if tvec is None or tvec.dim() == 1 or tvec.shape[0] == 1 else N
    out = torch.empty((M, N), dtype=torch.float32, device=dev)
    stats = torch.empty((M, 2), dtype=torch.float32, device=dev) if with_stats else None
    ms = C.c_float(0.0)
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().d3d_op_linear_postnorm(
            _ptr(A), _ptr(W), _ptr(bias), _ptr(residual), _ptr(gamma), _ptr(beta), float(eps), _ptr(pos), int(pos_div),
            int(pos.shape[0]) if pos is not None else 1, _ptr(tvec), stride, int(rows_per_batch), _ptr(out), _ptr(stats), M, N, K,
            int(reps), C.byref(ms), st))
    return out, stats, ms.value


def op_layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float) -> torch.Tensor:
    dev = x.device
    D = x.shape[-1]
    x2 = _f32c(x, dev).reshape(-1, D)
    out = torch.empty_like(x2)
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().d3d_op_layernorm(_ptr(x2), _ptr(_f32c(gamma, dev)), _ptr(_f32c(beta, dev)), _ptr(out),
                                               x2.shape[0], D, float(eps), st))
    return out.reshape(x.shape)


def op_attention(qkv: torch.Tensor, B: int, T: int, J: int, H: int, temporal: bool, precision: str = "fp32",
                 force_generic: bool = False) -> torch.Tensor:
    """qkv: (B*T*J, 3*D) packed GEMM output -> (B*T*J, D)."""
    dev = qkv.device
    D = qkv.shape[-1] // 3
    q = _f32c(qkv, dev).reshape(B * T * J, 3 * D)
    out = torch.empty((B * T * J, D), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().d3d_op_attention(_ptr(q), _ptr(out), B, T, J, D, H, int(temporal),
                                               _lib.PRECISIONS[precision], int(force_generic), st))
    return out


def tta_mpjpe(pred: torch.Tensor, pred_flip: Optional[torch.Tensor], gt: torch.Tensor, target_mask: Optional[torch.Tensor],
              scale: float, joints_left, joints_right, want_merged: bool = False):
    """evaluate() tail on device (RUN:583-590, LOSS:15-22). Returns (sum_err, n_joints[, merged])."""
    dev = pred.device
    B, T, J, _ = pred.shape
    p = _f32c(pred, dev)
    pf = _f32c(pred_flip, dev) if pred_flip is not None else None
    g = _f32c(gt, dev)
    m = target_mask.detach().to(device=dev, dtype=torch.uint8).contiguous() if target_mask is not None else None
    merged = torch.empty_like(p) if want_merged else None
    sums = torch.zeros(2, dtype=torch.float64, device=dev)
    jl = (C.c_int32 * len(joints_left))(*joints_left)
    jr = (C.c_int32 * len(joints_right))(*joints_right)
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().d3d_tta_mpjpe(_ptr(p), _ptr(pf), _ptr(g), _ptr(m), float(scale), jl, jr, len(joints_left),
                                            _ptr(merged), _ptr(sums), B, T, J, st))
    s = sums.cpu()
    return (float(s[0]), int(s[1]), merged) if want_merged else (float(s[0]), int(s[1]))


def pose_metrics(merged: torch.Tensor, gt: torch.Tensor, target_mask: Optional[torch.Tensor] = None):
    """evaluate()'s other three protocols on the merged, de-normalised prediction (RUN:602-614; LOSS:43-81 P-MPJPE, 83-93 N-MPJPE,
    132-142 MPJVE) -- d3d_pose_metrics.  merged / gt: (..., J, 3) with the same leading shape, flattened to frames in memory order;
    target_mask: one flag per frame (None: every frame kept).  Returns (kept_frames, n_mpjpe, p_mpjpe, mpjve) of THIS batch in the data's
    unit -- mpjve is nan for a batch of one kept frame (numpy's mean of an empty difference in the reference)."""
    dev = merged.device
    J = merged.shape[-2]
    p, g = _f32c(merged, dev), _f32c(gt, dev)
    assert p.shape == g.shape and p.shape[-1] == 3, (p.shape, g.shape)
    N = p.numel() // (J * 3)
    m = target_mask.detach().to(device=dev, dtype=torch.uint8).contiguous() if target_mask is not None else None
    assert m is None or m.numel() == N, (m.numel(), N)
    sums = torch.zeros(5, dtype=torch.float64, device=dev)
    if N > 0:
        with torch.cuda.device(dev):
            st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(_lib.lib().d3d_pose_metrics(_ptr(p), _ptr(g), _ptr(m), _ptr(sums), N, J, st))
    s = sums.cpu().tolist()
    kept, pairs = int(s[3]), int(s[4])
    nan = float("nan")
    return (kept, s[0] / (J * kept) if kept else nan, s[1] / (J * kept) if kept else nan, s[2] / (J * pairs) if pairs else nan)


def window_gather(seq: torch.Tensor, T: int, flip: bool = False, joints_left=(), joints_right=(), want_mask: bool = True):
    """(n, J, C) device tensor -> (windows, T, J, C) evaluation windows [+ (windows, T) bool target mask] (GEN:27-48, 247-276)."""
    dev = seq.device
    n, J, Cc = seq.shape
    sq = _f32c(seq, dev)
    nc = _lib.lib().d3d_num_windows(n, T)
    out = torch.empty((nc, T, J, Cc), dtype=torch.float32, device=dev)
    mask = torch.empty((nc, T), dtype=torch.uint8, device=dev) if want_mask else None
    jl = (C.c_int32 * len(joints_left))(*joints_left)
    jr = (C.c_int32 * len(joints_right))(*joints_right)
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().d3d_window_gather(_ptr(sq), n, T, J, Cc, int(flip), jl, jr, len(joints_left), _ptr(out), _ptr(mask), st))
    return (out, mask.bool()) if want_mask else out


def window_gather_s2f(seq: torch.Tensor, T: int, flip: bool = False, joints_left=(), joints_right=(), first: int = 0,
                      count: Optional[int] = None) -> torch.Tensor:
    """(n, J, C) device tensor -> (count, T, J, C) seq2frame windows: one per target frame first .. first + count - 1, frames
    f - (T-1)/2 .. f + (T-1)/2 edge-replicated (GEN:402-420, 492-512 with out_all=False, stride 1)."""
    dev = seq.device
    n, J, Cc = seq.shape
    count = n - first if count is None else count
    sq = _f32c(seq, dev)
    out = torch.empty((count, T, J, Cc), dtype=torch.float32, device=dev)
    if count == 0:
        return out
    jl = (C.c_int32 * len(joints_left))(*joints_left)
    jr = (C.c_int32 * len(joints_right))(*joints_right)
    with torch.cuda.device(dev):
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(_lib.lib().d3d_window_gather_s2f(_ptr(sq), n, T, J, Cc, int(flip), jl, jr, len(joints_left), int(first), int(count),
                                                    _ptr(out), st))
    return out
