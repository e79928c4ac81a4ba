"""Python handle around the native engine (liboracle_hip.so). PyTorch is used for device memory and streams
only: every method passes raw `data_ptr()`s and the current HIP stream across the C ABI."""
import ctypes as C

import torch

from . import _lib
from .dims import Dims

EPI = {'none': 0, 'quickgelu': 1, 'gelu': 2, 'residual': 3, 'swiglu': 4, 'f32': 5}


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk_bf16(t, name):
    if t.dtype != torch.bfloat16 or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f'{name} must be a contiguous CUDA bfloat16 tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}')


class Engine:
    def __init__(self, dims: Dims, max_batch=8, max_seq=2400, max_images=32, max_new_tokens=512,
                 pad_token_id=0, eos_token_id=2, device=None):
        if not torch.cuda.is_available():
            raise _lib.OracleHipError('no HIP device: the ORacle engine runs on MI355X only (no CPU fallback)')
        self.lib = _lib.lib()
        self.dims = dims
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        cfg = _lib.OrcConfig(
            vocab=dims.vocab, d_model=dims.d_model, n_layers=dims.n_layers, n_heads=dims.n_heads, d_ff=dims.d_ff,
            rms_eps=dims.rms_eps, rope_theta=dims.rope_theta,
            clip_layers=dims.clip_layers, clip_d=dims.clip_d, clip_heads=dims.clip_heads, clip_ff=dims.clip_ff,
            clip_image=dims.clip_image, clip_patch=dims.clip_patch, clip_eps=dims.clip_eps,
            pool_layers=dims.pool_layers, pool_heads=dims.pool_heads, pool_ff=dims.pool_ff, pool_maxpos=dims.pool_maxpos,
            pool_eps=dims.pool_eps,
            max_batch=max_batch, max_seq=max_seq, max_images=max_images, max_new_tokens=max_new_tokens,
            pad_token_id=pad_token_id, eos_token_id=eos_token_id)
        self.cfg = cfg
        self.max_batch, self.max_seq, self.max_images, self.max_new_tokens = max_batch, max_seq, max_images, max_new_tokens
        self.pad_token_id, self.eos_token_id = pad_token_id, eos_token_id
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.orc_engine_create(C.byref(cfg), C.byref(self._h)), 'orc_engine_create')
        self.finalized = False
        self.batch = 0

    def close(self):
        if getattr(self, '_h', None) and self._h.value:
            self.lib.orc_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def load_weight(self, name, tensor):
        t = tensor.detach()
        if t.dtype != torch.bfloat16:
            t = t.to(torch.bfloat16)
        t = t.contiguous()
        shape = (C.c_int64 * t.dim())(*t.shape)
        _lib.check(self.lib.orc_engine_load_weight(self._h, name.encode(), _ptr(t), 1 if t.is_cuda else 0, shape, t.dim()),
                   f'load_weight({name})')

    def load_state(self, state, finalize=True):
        with torch.cuda.device(self.device):
            for k, v in state.items():
                self.load_weight(k, v)
            if finalize:
                self.finalize()

    def finalize(self):
        with torch.cuda.device(self.device):
            _lib.check(self.lib.orc_engine_finalize(self._h), 'orc_engine_finalize')
        self.finalized = True

    # ------------------------------------------------------------------ vision
    def vision_encode(self, images):
        """(n,3,I,I) bf16 -> (n, patches, clip_d): CLIPVisionTower.forward (clip_encoder.py:40-51)."""
        _chk_bf16(images, 'images')
        n = images.shape[0]
        out = torch.empty(n, self.dims.clip_patches, self.dims.clip_d, dtype=torch.bfloat16, device=images.device)
        _lib.check(self.lib.orc_vision_encode(self._h, _ptr(images), n, _ptr(out), _stream()), 'orc_vision_encode')
        return out

    def vision_global(self, images):
        _chk_bf16(images, 'images')
        n = images.shape[0]
        out = torch.empty(n, self.dims.clip_d, dtype=torch.bfloat16, device=images.device)
        _lib.check(self.lib.orc_vision_global(self._h, _ptr(images), n, _ptr(out), _stream()), 'orc_vision_global')
        return out

    def pool_views(self, features, views_per_row):
        """features (sum(V_b), patches, clip_d) -> (B, patches, clip_d): pad_embeddings + ImageEmbeddingPooler."""
        _chk_bf16(features, 'features')
        b = len(views_per_row)
        if sum(views_per_row) != features.shape[0]:
            raise ValueError('views_per_row does not add up to the number of feature maps')
        arr = (C.c_int32 * b)(*[int(v) for v in views_per_row])
        out = torch.empty(b, self.dims.clip_patches, self.dims.clip_d, dtype=torch.bfloat16, device=features.device)
        _lib.check(self.lib.orc_pool_views(self._h, _ptr(features), arr, b, _ptr(out), _stream()), 'orc_pool_views')
        return out

    def project(self, x):
        _chk_bf16(x, 'x')
        rows = x.numel() // self.dims.clip_d
        out = torch.empty(*x.shape[:-1], self.dims.d_model, dtype=torch.bfloat16, device=x.device)
        _lib.check(self.lib.orc_project(self._h, _ptr(x), rows, _ptr(out), _stream()), 'orc_project')
        return out

    def view_reduce(self, x, mode):
        """(B,V,...) -> (B,...), mode 'max' | 'avg'."""
        _chk_bf16(x, 'x')
        b, v = x.shape[:2]
        n = x[0, 0].numel()
        out = torch.empty(b, *x.shape[2:], dtype=torch.bfloat16, device=x.device)
        _lib.check(self.lib.orc_view_reduce(self._h, _ptr(x), b, v, n, 0 if mode == 'max' else 1, _ptr(out), _stream()), 'orc_view_reduce')
        return out

    def pool2x2(self, x):
        """(n, G*G, d) -> (n, (G/2)^2, d)."""
        _chk_bf16(x, 'x')
        n, gg, d = x.shape
        g = int(round(gg ** 0.5))
        out = torch.empty(n, (g // 2) ** 2, d, dtype=torch.bfloat16, device=x.device)
        _lib.check(self.lib.orc_pool2x2(self._h, _ptr(x), n, g, d, _ptr(out), _stream()), 'orc_pool2x2')
        return out

    # ------------------------------------------------------------------ splice + LLaMA
    def splice(self, codes, image_features, descriptors=None):
        """codes int32 (rows,) device; returns (rows, d_model)."""
        rows = codes.numel()
        out = torch.empty(rows, self.dims.d_model, dtype=torch.bfloat16, device=codes.device)
        _lib.check(self.lib.orc_splice(self._h, _ptr(codes), rows, _ptr(image_features), _ptr(descriptors), _ptr(out), _stream()), 'orc_splice')
        return out

    def prefill(self, embeds, position_ids, left_pad, want_logits=False):
        """embeds (B,T,D) bf16, position_ids (B,T) int32 device, left_pad list[int]. Resets the greedy state."""
        _chk_bf16(embeds, 'embeds')
        b, t, _ = embeds.shape
        pos = position_ids.to(torch.int32).contiguous()
        lp = (C.c_int32 * b)(*[int(x) for x in left_pad])
        logits = torch.empty(b, self.dims.vocab, dtype=torch.float32, device=embeds.device) if want_logits else None
        _lib.check(self.lib.orc_prefill(self._h, _ptr(embeds), _ptr(pos), lp, b, t, _ptr(logits), _stream()), 'orc_prefill')
        self.batch = b
        return logits

    def set_row_limits(self, limits):
        arr = (C.c_int32 * len(limits))(*[int(x) for x in limits])
        _lib.check(self.lib.orc_set_row_limits(self._h, arr, len(limits), _stream()), 'orc_set_row_limits')

    def decode_step(self, forced_ids=None, want_logits=False):
        f = forced_ids.to(torch.int32).contiguous() if forced_ids is not None else None
        logits = torch.empty(self.batch, self.dims.vocab, dtype=torch.float32, device=self.device) if want_logits else None
        _lib.check(self.lib.orc_decode_step(self._h, _ptr(f), _ptr(logits), _stream()), 'orc_decode_step')
        return logits

    def decode_steps(self, n):
        _lib.check(self.lib.orc_decode_steps(self._h, int(n), _stream()), 'orc_decode_steps')

    def read_state(self):
        """-> (tokens (B, steps) int64 CPU tensor, steps, n_unfinished, last raw argmax (B,))"""
        import numpy as np
        b = self.batch
        toks = np.empty((b, self.max_new_tokens), dtype=np.int32)
        steps, nunf = C.c_int32(), C.c_int32()
        am = (C.c_int32 * b)()
        _lib.check(self.lib.orc_read_state(self._h, toks.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(steps), C.byref(nunf), am,
                                           _stream()), 'orc_read_state')
        t = torch.from_numpy(toks).long()[:, :min(steps.value, self.max_new_tokens)]
        return t, steps.value, nunf.value, torch.tensor(list(am), dtype=torch.long)

    # ------------------------------------------------------------------ continuous batching
    def stream_begin(self, n_rows):
        """n_rows decode rows, none attached to a slot yet (the engine has max_batch KV-cache slots)."""
        _lib.check(self.lib.orc_stream_begin(self._h, int(n_rows), _stream()), 'orc_stream_begin')
        self.batch = int(n_rows)

    def prefill_slots(self, slots, embeds, position_ids, left_pad, limits, want_logits=False):
        """embeds (n, T, D) bf16 left-padded, position_ids (n, T), into the unattached KV slots `slots`; limits = length stops."""
        _chk_bf16(embeds, 'embeds')
        n, t, _ = embeds.shape
        pos = position_ids.to(device=embeds.device, dtype=torch.int32).contiguous()
        arr = lambda xs: (C.c_int32 * n)(*[int(x) for x in xs])
        logits = torch.empty(n, self.dims.vocab, dtype=torch.float32, device=embeds.device) if want_logits else None
        _lib.check(self.lib.orc_prefill_slots(self._h, arr(slots), n, _ptr(embeds), _ptr(pos), arr(left_pad), arr(limits), t,
                                              _ptr(logits), _stream()), 'orc_prefill_slots')
        return logits

    def stream_attach(self, row, slot):
        _lib.check(self.lib.orc_stream_attach(self._h, int(row), int(slot)), 'orc_stream_attach')

    def read_slots(self):
        """-> (tokens (max_batch, max_new_tokens) int64 CPU tensor, tokens emitted per slot (list), unfinished flags (list))"""
        import numpy as np
        b = self.max_batch
        toks = np.empty((b, self.max_new_tokens), dtype=np.int32)
        steps = (C.c_int32 * b)()
        unf = (C.c_int32 * b)()
        _lib.check(self.lib.orc_read_slots(self._h, toks.ctypes.data_as(C.POINTER(C.c_int32)), steps, unf, _stream()), 'orc_read_slots')
        return torch.from_numpy(toks).long(), list(steps), list(unf)

    def profile_enable(self, on=True):
        _lib.check(self.lib.orc_profile_enable(self._h, 1 if on else 0), 'orc_profile_enable')

    def profile_read(self):
        """-> (total_ms, launches, algorithmic bytes per launch) of the sampled gate/up GEMV launches."""
        ms, n, by = C.c_double(), C.c_int32(), C.c_int64()
        _lib.check(self.lib.orc_profile_read(self._h, C.byref(ms), C.byref(n), C.byref(by)), 'orc_profile_read')
        return ms.value, n.value, by.value

    def decode_weight_bytes(self):
        return int(self.lib.orc_decode_weight_bytes(self._h))

    def kv_bytes_per_key(self):
        return int(self.lib.orc_kv_bytes_per_key(self._h))

    def decode_weight_stats(self):
        """Bookkeeping of the lossless 12-bit decode copies: dict(matrices, compressed, elements, escapes, blocks, flagged_blocks)."""
        out = (C.c_int64 * 6)()
        _lib.check(self.lib.orc_decode_weight_stats(self._h, out), 'orc_decode_weight_stats')
        return dict(zip(('matrices', 'compressed', 'elements', 'escapes', 'blocks', 'flagged_blocks'), [int(v) for v in out]))

    def decode_stream_bytes(self, which=0):
        """Physical weight bytes streamed per decode step (which=0) or per gate/up launch (which=1): 12-bit copies where active."""
        return int(self.lib.orc_decode_stream_bytes(self._h, which))


# ---------------------------------------------------------------------- single-kernel hooks (tests, profiling)
def test_gemm(x, w, bias=None, residual=None, epilogue='none', use_gemv=False, kernel=None):
    """kernel: None (automatic), 'tile128' (register-staged 128^2), 'dma256' / 'dma128' (LDS-DMA, BM x 256 tiles of one
    height), 'mix<n>' (LDS-DMA, n 256-row m-tiles + 128-row tiles for the remaining rows in one launch).
    use_gemv='cw': the decode GEMV on the lossless 12-bit copy of w (must equal use_gemv=True bit for bit);
    'legacy' / 'legacy_cw': the same two on the r2 kernels (branchy k-loop) instead of the pipelined ones."""
    l = _lib.lib()
    m, k = x.shape
    n = w.shape[0]
    nout = n // 2 if epilogue == 'swiglu' else n
    out = torch.empty(m, nout, dtype=torch.float32 if epilogue == 'f32' else torch.bfloat16, device=x.device)
    _lib.check(l.orc_test_gemm(_ptr(x), m, k, _ptr(w), n, _ptr(bias), _ptr(residual), EPI[epilogue],
                               {'cw': 5, 'legacy': 6, 'legacy_cw': 7, 'cw_rowmajor': 8}[use_gemv] if isinstance(use_gemv, str) else 1 if use_gemv else (100 + int(kernel[3:]) if isinstance(kernel, str) and kernel.startswith('mix') else {None: 0, 'tile128': 2, 'dma256': 3, 'dma128': 4}[kernel]),
                               _ptr(out), _stream()), 'orc_test_gemm')
    return out


def test_folded_norm(x, w_norm, w, eps):
    """-> (Linear(rmsnorm(x)) with the reference's rounding points, the decode path's folded form), both (M, N) bf16."""
    m, k = x.shape
    n = w.shape[0]
    a = torch.empty(m, n, dtype=torch.bfloat16, device=x.device)
    b = torch.empty_like(a)
    _lib.check(_lib.lib().orc_test_folded_norm(_ptr(x), m, k, _ptr(w_norm), _ptr(w), n, eps, _ptr(a), _ptr(b), _stream()), 'orc_test_folded_norm')
    return a, b


def test_rmsnorm(x, w, eps):
    out = torch.empty_like(x)
    _lib.check(_lib.lib().orc_test_rmsnorm(_ptr(x), _ptr(w), x.shape[0], x.shape[1], eps, _ptr(out), _stream()), 'rmsnorm')
    return out


def test_layernorm(x, w, b, eps):
    out = torch.empty_like(x)
    _lib.check(_lib.lib().orc_test_layernorm(_ptr(x), _ptr(w), _ptr(b), x.shape[0], x.shape[1], eps, _ptr(out), _stream()), 'layernorm')
    return out


def test_attention(q, k, v, kbeg=None, kend=None, causal=False, q_off=0, scale=None):
    """q (B,H,Sq,hd), k/v (B,H,Sk,hd) contiguous bf16; kbeg/kend int32 device tensors (B,) or None."""
    b, h, sq, hd = q.shape
    sk = k.shape[2]
    out = torch.empty_like(q)
    scale = hd ** -0.5 if scale is None else scale
    _lib.check(_lib.lib().orc_test_attention(_ptr(q), _ptr(k), _ptr(v), b, h, sq, sk, hd, _ptr(kbeg), _ptr(kend),
                                             1 if causal else 0, q_off, scale, _ptr(out), _stream()), 'attention')
    return out


def test_decode_attention(q, kcache, vcache, n_keys, kbeg=None):
    """q (B, H*128), kcache / vcache (B, H, Tmax, 128) contiguous bf16; kbeg int32 device (B,) or None -> (B, H*128)."""
    b, h, tmax, hd = kcache.shape
    assert hd == 128 and q.shape == (b, h * 128)
    out = torch.empty_like(q)
    _lib.check(_lib.lib().orc_test_decode_attention(_ptr(q), _ptr(kcache), _ptr(vcache), b, h, tmax, _ptr(kbeg), int(n_keys),
                                                    _ptr(out), _stream()), 'decode_attention')
    return out
