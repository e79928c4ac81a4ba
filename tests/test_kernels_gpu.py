"""GPU parity of the individual HIP kernels (called through the C ABI test hooks) against the torch-CPU
ops the oracle is made of. Tolerance: max rel err <= 1e-2 (bf16), stated by BASELINE.json's north_star;
most outputs are bit-identical, the rest differ by one bf16 ulp from fp32 summation order."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-2


def _dev():
    return torch.device('cuda')


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)


@pytest.mark.parametrize('use_gemv,M', [(False, 300), (False, 128), (False, 1), (True, 1), (True, 8), (True, 16), (True, 17), (True, 32)])
@pytest.mark.parametrize('epi', ['none', 'quickgelu', 'gelu', 'residual', 'swiglu', 'f32'])
def test_gemm_epilogues(use_gemv, M, epi):
    from oracle_amd import engine as E
    if use_gemv and epi == 'quickgelu':
        pytest.skip('decode path has no QuickGELU GEMM')
    K, N = 512, 384
    x = _rand((M, K), 1)
    w = _rand((N, K), 2, 0.05)
    bias = _rand((N,), 3, 0.1) if epi not in ('swiglu', 'f32') else None
    res = _rand((M, N), 4) if epi == 'residual' else None
    y = F.linear(x, w, bias)
    if epi == 'quickgelu':
        ref = y * torch.sigmoid(1.702 * y)
    elif epi == 'gelu':
        ref = F.gelu(y)
    elif epi == 'residual':
        ref = res + y
    elif epi == 'swiglu':
        ref = F.silu(y[:, :N // 2]) * y[:, N // 2:]
    elif epi == 'f32':
        ref = y.float()
    else:
        ref = y
    d = _dev()
    out = E.test_gemm(x.to(d), w.to(d), None if bias is None else bias.to(d), None if res is None else res.to(d), epi, use_gemv)
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    assert rel_err(out, ref) <= TOL
    exact = (out.float().cpu() == ref.float()).float().mean().item()
    assert exact > 0.6, f'only {exact:.3f} of the outputs are bit-identical'


@pytest.mark.parametrize('K,N', [(4096, 4096), (11008, 4096), (4096, 256), (640, 1024)])
def test_gemm_shapes(K, N):
    from oracle_amd import engine as E
    M = 257
    x = _rand((M, K), 5)
    w = _rand((N, K), 6, 0.02)
    ref = F.linear(x, w)
    d = _dev()
    out = E.test_gemm(x.to(d), w.to(d))
    assert rel_err(out, ref) <= TOL
    out2 = E.test_gemm(x[:8].contiguous().to(d), w.to(d), use_gemv=True)
    assert rel_err(out2, ref[:8]) <= TOL


@pytest.mark.parametrize('kernel', ['tile128', 'dma256', 'dma128', 'mix2', 'mix1'])
@pytest.mark.parametrize('epi', ['none', 'quickgelu', 'gelu', 'residual', 'swiglu', 'f32'])
def test_gemm_kernels_agree(kernel, epi):
    """Every large-M GEMM kernel (register-staged 128^2; LDS-DMA BM x 256) against torch, ragged M and padded N."""
    from oracle_amd import engine as E
    M, K, N = 777, 1024, 640
    x = _rand((M, K), 11)
    w = _rand((N, K), 12, 0.03)
    bias = _rand((N,), 13, 0.1) if epi not in ('swiglu', 'f32') else None
    res = _rand((M, N), 14) if epi == 'residual' else None
    y = F.linear(x, w, bias)
    ref = {'none': y, 'quickgelu': y * torch.sigmoid(1.702 * y), 'gelu': F.gelu(y), 'residual': (res + y) if res is not None else y,
           'swiglu': F.silu(y[:, :N // 2]) * y[:, N // 2:], 'f32': y.float()}[epi]
    d = _dev()
    out = E.test_gemm(x.to(d), w.to(d), None if bias is None else bias.to(d), None if res is None else res.to(d), epi, kernel=kernel)
    torch.cuda.synchronize()
    assert out.shape == ref.shape
    assert rel_err(out, ref) <= TOL
    assert (out.float().cpu() == ref.float()).float().mean().item() > 0.6


def test_norms():
    from oracle_amd import engine as E
    d = _dev()
    for dim in (512, 1024, 4096):
        x = _rand((37, dim), 7, 3.0)
        w = (1 + 0.1 * _rand((dim,), 8).float()).to(torch.bfloat16)
        b = _rand((dim,), 9, 0.1)
        xf = x.float()
        ref = w * (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)).to(torch.bfloat16)
        out = E.test_rmsnorm(x.to(d), w.to(d), 1e-5)
        assert rel_err(out, ref) <= TOL
        assert (out.cpu() == ref).float().mean() > 0.98
        ref = F.layer_norm(x, (dim,), w, b, 1e-5)
        out = E.test_layernorm(x.to(d), w.to(d), b.to(d), 1e-5)
        assert rel_err(out, ref) <= TOL
        assert (out.cpu() == ref).float().mean() > 0.98


def _eager(q, k, v, add, scale):
    w = torch.matmul(q, k.transpose(-1, -2)) * scale
    if add is not None:
        w = w + add
    w = F.softmax(w, dim=-1, dtype=torch.float32).to(q.dtype)
    return torch.matmul(w, v)


@pytest.mark.parametrize('hd,H,Sq', [(64, 3, 577), (128, 2, 200), (128, 2, 64), (64, 2, 33)])
def test_attention_plain(hd, H, Sq):
    from oracle_amd import engine as E
    B = 2
    q, k, v = _rand((B, H, Sq, hd), 10), _rand((B, H, Sq, hd), 11), _rand((B, H, Sq, hd), 12)
    ref = _eager(q, k, v, None, hd ** -0.5)
    d = _dev()
    out = E.test_attention(q.to(d), k.to(d), v.to(d))
    assert rel_err(out, ref) <= TOL


def test_attention_causal_leftpad():
    from oracle_amd import engine as E
    B, H, S, hd = 3, 2, 150, 128
    q, k, v = _rand((B, H, S, hd), 13), _rand((B, H, S, hd), 14), _rand((B, H, S, hd), 15)
    pad = [0, 37, 100]
    dt = torch.bfloat16
    i = torch.arange(S)[:, None]
    j = torch.arange(S)[None, :]
    add = torch.zeros(B, 1, S, S, dtype=dt)
    for b in range(B):
        ok = (j <= i) & (j >= pad[b])
        add[b, 0].masked_fill_(~ok, torch.finfo(dt).min)
    ref = _eager(q, k, v, add, hd ** -0.5)
    d = _dev()
    kbeg = torch.tensor(pad, dtype=torch.int32, device=d)
    out = E.test_attention(q.to(d), k.to(d), v.to(d), kbeg=kbeg, causal=True).cpu()
    for b in range(B):   # rows inside the padding are never read by the path
        assert rel_err(out[b, :, pad[b]:], ref[b, :, pad[b]:]) <= TOL
        assert torch.isfinite(out[b].float()).all()


def test_attention_key_padding():
    from oracle_amd import engine as E
    B, H, S, hd = 2, 2, 1152, 128
    q, k, v = _rand((B, H, S, hd), 16), _rand((B, H, S, hd), 17), _rand((B, H, S, hd), 18)
    valid = [1152, 576]
    dt = torch.bfloat16
    add = torch.zeros(B, 1, 1, S, dtype=dt)
    for b in range(B):
        add[b, 0, 0, valid[b]:] = torch.finfo(dt).min
    ref = _eager(q, k, v, add, hd ** -0.5)
    d = _dev()
    kend = torch.tensor(valid, dtype=torch.int32, device=d)
    out = E.test_attention(q.to(d), k.to(d), v.to(d), kend=kend)
    assert rel_err(out, ref) <= TOL


@pytest.mark.parametrize('epi', ['none', 'residual', 'swiglu'])
def test_gemm_mixed_tile_heights_bitwise(epi):
    """Prefill-sized M (8 x 739 rows), where the planner mixes 256-row and 128-row tiles in one launch: the automatic
    choice, an explicit mix and both pure forms must agree bit for bit (same k order and MFMA per output element)."""
    from oracle_amd import engine as E
    M, K, N = 5912, 512, 4096
    x = _rand((M, K), 21)
    w = _rand((N, K), 22, 0.03)
    res = _rand((M, N), 23) if epi == 'residual' else None
    d = _dev()
    outs = {k: E.test_gemm(x.to(d), w.to(d), None, None if res is None else res.to(d), epi, kernel=k).cpu()
            for k in (None, 'dma256', 'dma128', 'mix16', 'mix7')}
    y = F.linear(x, w)
    ref = {'none': y, 'residual': (res + y) if res is not None else y, 'swiglu': F.silu(y[:, :N // 2]) * y[:, N // 2:]}[epi]
    assert rel_err(outs[None], ref) <= TOL
    for k, v in outs.items():
        assert torch.equal(v, outs['dma256']), k


@pytest.mark.parametrize('B,H', [(2, 32), (6, 32), (8, 32), (16, 32), (16, 12), (1, 32), (3, 4), (32, 32), (21, 4)])
def test_decode_attention_split_and_unsplit_branches(B, H):
    """Decode attention over the KV cache through the engine's own launch: B*H in {64, 192, 256, 512, ...} covers the
    split + combine path (B*H < 192) and the un-split single-workgroup path the batch-8 headline runs (8 x 32 = 256),
    with ragged left padding. Reference = the eager formula of LlamaAttention for one query token (scores rounded to
    bf16 after the product and after the scale, fp32 softmax, bf16 probabilities)."""
    from oracle_amd import engine as E
    tmax, n_keys = 800, 777
    q = _rand((B, H * 128), 30 + B)
    kc, vc = _rand((B, H, tmax, 128), 31 + B), _rand((B, H, tmax, 128), 32 + B)
    pad = [(37 * b * b) % 700 for b in range(B)]
    if B > 1:
        pad[-1] = n_keys - 1                                            # a row with a single visible key
    dt = torch.bfloat16
    add = torch.zeros(B, 1, 1, n_keys, dtype=dt)
    for b in range(B):
        add[b, 0, 0, :pad[b]] = torch.finfo(dt).min
    ref = _eager(q.view(B, H, 1, 128), kc[:, :, :n_keys], vc[:, :, :n_keys], add, 128 ** -0.5).reshape(B, H * 128)
    d = _dev()
    kbeg = torch.tensor(pad, dtype=torch.int32, device=d)
    out = E.test_decode_attention(q.to(d), kc.to(d), vc.to(d), n_keys, kbeg).cpu()
    assert torch.isfinite(out.float()).all()
    assert rel_err(out, ref) <= TOL, (B, H, rel_err(out, ref))
    for b in range(B):                                                  # per row, so a wrong row cannot hide behind a large one
        assert rel_err(out[b], ref[b]) <= TOL, (b, pad[b])
    if B > 1:                                                           # one visible key: the output IS that value row
        assert torch.equal(out[-1].view(H, 128), vc[-1, :, n_keys - 1])


@pytest.mark.parametrize('M', [1, 8, 16, 24, 32])
@pytest.mark.parametrize('epi,N,K', [('swiglu', 22016, 4096), ('f32', 32000, 4096), ('residual', 4096, 11008), ('none', 12288, 4096)])
def test_decode_gemv_at_llama7b_shapes(epi, N, K, M):
    """The decode GEMVs at the real LLaMA-7B shapes and 1 / 8 / 16 live rows: gate/up (688 units) and lm_head (1000 units)
    take the PERSISTENT kernels the headline bench runs (gemv_pers_kernel<2,SWIGLU,12,1>, <2,F32,8,1>), down the 16-wave
    one-unit kernel, 12288 x 4096 the 384-unit shape of the fused q/k/v projection. Also pins batch invariance: row 0 of
    the M-row launch is bit-identical to the same row launched alone (the K split depends on the shape only)."""
    from oracle_amd import engine as E
    x = _rand((M, K), 41)
    w = _rand((N, K), 42, 0.02)
    res = _rand((M, N), 43) if epi == 'residual' else None
    y = F.linear(x, w)
    if epi == 'swiglu':
        ref = F.silu(y[:, :N // 2]) * y[:, N // 2:]
    elif epi == 'residual':
        ref = res + y
    elif epi == 'f32':
        ref = F.linear(x.float(), w.float())
    else:
        ref = y
    d = _dev()
    out = E.test_gemm(x.to(d), w.to(d), None, None if res is None else res.to(d), epi, use_gemv=True).cpu()
    assert rel_err(out, ref) <= TOL, rel_err(out, ref)
    if M > 1:
        one = E.test_gemm(x[:1].to(d), w.to(d), None, None if res is None else res[:1].to(d), epi, use_gemv=True).cpu()
        assert torch.equal(one[0], out[0])
    if M > 16:       # rows of the second MFMA B tile: the same bits as when they run alone as rows 0..M-17
        tail = E.test_gemm(x[16:].to(d), w.to(d), None, None if res is None else res[16:].to(d), epi, use_gemv=True).cpu()
        assert torch.equal(tail, out[16:])


def _nasty_weights(N, K, seed):
    """Gaussian weights with everything the 12-bit block format has to escape or cover, at rates that stay under the 256 patch
    entries of a GEMV unit (up to 176k elements): exact zeros (+0 and -0), values 2^-20 and 2^-30 times the scale (below a
    block's 16 binades), bf16 denormals, three huge outliers (each lifts its block's base and pushes ~50 ordinary values of that
    block out of range), an all-zero 16-row tile (padding), a dense run of zeros inside one block."""
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(N, K, generator=g) * 0.02
    r = torch.rand(N, K, generator=g)
    w[r < 1e-4] = 0.0
    w[(r >= 1e-3) & (r < 1.05e-3)] = -0.0
    w[(r >= 2e-3) & (r < 2.1e-3)] *= 2.0 ** -20
    w[(r >= 3e-3) & (r < 3.05e-3)] *= 2.0 ** -30
    w[(r >= 4e-3) & (r < 4.02e-3)] = 1e-40                      # bf16 denormal
    for n, k in ((3, 7), (N // 3, K // 2 + 1), (N - 2, K - 5)):
        w[n, k] = 90.0
    w[32:48] = 0.0
    w[5, 64:96] = 0.0
    w[64:80][torch.rand(16, K, generator=g) < 0.4] = 0.0        # a pruned tile: thousands of zeros in one unit (they need no table space)
    w[81, ::2] = -0.0
    return w.to(torch.bfloat16)


@pytest.mark.parametrize('M', [1, 8, 32])
@pytest.mark.parametrize('epi,N,K', [('swiglu', 22016, 4096), ('f32', 32000, 4096), ('residual', 4096, 11008), ('residual', 4096, 4096),
                                     ('none', 12288, 4096), ('swiglu', 2816, 512), ('residual', 512, 1408), ('f32', 32000, 512),
                                     ('none', 1536, 512)])
def test_compressed_weight_stream_is_bit_exact(epi, N, K, M):
    """The decode GEMVs on the lossless 12-bit copy of the weights (768 B per 1-KiB fragment block, gemv.hip) against the same
    kernels on the bf16 weights: every output bit equal, for plain Gaussian weights and for weights full of escapes (zeros,
    tiny values, denormals, outliers), at the 7B shapes and the TINY ones, 1 / 8 / 32 rows."""
    from oracle_amd import engine as E
    d = _dev()
    x = _rand((M, K), 51).to(d)
    res = _rand((M, N), 53).to(d) if epi == 'residual' else None
    for w in (_rand((N, K), 52, 0.02), _nasty_weights(N, K, 54)):
        w = w.to(d)
        plain = E.test_gemm(x, w, None, res, epi, use_gemv=True)
        packed = E.test_gemm(x, w, None, res, epi, use_gemv='cw')
        assert torch.equal(plain.view(torch.int32 if epi == 'f32' else torch.int16), packed.view(torch.int32 if epi == 'f32' else torch.int16))


def _shaped_weights(kind, N, K, seed):
    """Weight distributions a trained / LoRA-merged LLaMA shows and a random init does not."""
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(N, K, generator=g) * 0.02
    if kind == 'student_t3':                                   # heavy tails: block maxima 10-20 sigma
        chi = (torch.randn(3, N, K, generator=g) ** 2).sum(0) / 3
        w = w / chi.sqrt()
    elif kind == 'outlier_channels':                           # a few input channels 30x, a few output rows 8x (massive activations' partners)
        cols = torch.randperm(K, generator=g)[:8]                # (a whole row 20x lifts its 16-row tile 4-5 binades: > 254 escapes in a
        rows = torch.randperm(N, generator=g)[:6]                #  K = 11008 unit, the matrix then keeps bf16 -- DESIGN 3 table)
        w[:, cols] *= 30.0
        w[rows] *= 8.0
    elif kind == 'lognormal_rows':                             # per-row scales e^N(0, 0.5): the 16 rows of a block spread over ~4 binades
        w = w * torch.exp(torch.randn(N, 1, generator=g) * 0.5)
    elif kind == 'lora_merged':                                # W + (alpha / r) B A, r = 16, alpha / r = 2 (the released checkpoints' merge)
        a_ = torch.randn(16, K, generator=g) * 0.02
        b_ = torch.randn(N, 16, generator=g) * 0.05
        w = w + 2.0 * (b_ @ a_)
    else:
        raise ValueError(kind)
    return w.to(torch.bfloat16)


@pytest.mark.parametrize('kind', ['student_t3', 'outlier_channels', 'lognormal_rows', 'lora_merged'])
@pytest.mark.parametrize('epi,N,K', [('swiglu', 22016, 4096), ('residual', 4096, 11008), ('none', 12288, 4096), ('f32', 32000, 4096)])
def test_compressed_weight_stream_on_non_gaussian_weights(kind, epi, N, K):
    """The 12-bit stream at the 7B shapes on heavy-tailed, outlier-channel, per-row-scaled and LoRA-merged weights: every matrix
    takes it (no unit above 254 patch entries) and every output bit equals the bf16 stream's. tools/decode_bench DIST_EXP times
    the same distributions (table in DESIGN 3)."""
    from oracle_amd import engine as E
    d = _dev()
    w = _shaped_weights(kind, N, K, 71).to(d)
    x = _rand((8, K), 72).to(d)
    res = _rand((8, N), 73).to(d) if epi == 'residual' else None
    plain = E.test_gemm(x, w, None, res, epi, use_gemv=True)
    packed = E.test_gemm(x, w, None, res, epi, use_gemv='cw')
    v = torch.int32 if epi == 'f32' else torch.int16
    assert torch.equal(plain.view(v), packed.view(v))


@pytest.mark.parametrize('M', [1, 8, 17])
@pytest.mark.parametrize('epi,N,K', [('residual', 4096, 11008), ('residual', 4096, 4096), ('swiglu', 22016, 4096), ('f32', 32000, 4096),
                                     ('residual', 512, 1408), ('none', 1024, 3072), ('none', 768, 832)])
def test_interleaved_compressed_blocks_equal_row_major(epi, N, K, M):
    """r3: the 768-B blocks of a compressed decode matrix are stored interleaved in groups of 16 k-tiles (all n-tiles' blocks of a group
    together, the last group shorter when K / 32 is not a multiple of 16: K = 11008 -> 344 = 21 x 16 + 8, K = 1408 -> 44, K = 832 -> 26),
    so that a launch reads one contiguous window per step (gemv.hip cw_block_index). Same data, different placement: every output bit
    equals the row-major order's, on the pipelined kernels, the persistent / r2 fallbacks (short K) and the r2 kernels forced."""
    from oracle_amd import engine as E
    d = _dev()
    x = _rand((M, K), 81).to(d)
    res = _rand((M, N), 83).to(d) if epi == 'residual' else None
    w = _nasty_weights(N, K, 84).to(d) if N * K < 5e7 else _rand((N, K), 82, 0.02).to(d)
    v = torch.int32 if epi == 'f32' else torch.int16
    a = E.test_gemm(x, w, None, res, epi, use_gemv='cw')
    b = E.test_gemm(x, w, None, res, epi, use_gemv='cw_rowmajor')
    c = E.test_gemm(x, w, None, res, epi, use_gemv='legacy_cw')
    assert torch.equal(a.view(v), b.view(v))
    assert torch.equal(a.view(v), c.view(v))


@pytest.mark.parametrize('M', [1, 8, 16, 17, 32])
@pytest.mark.parametrize('epi,N,K', [('residual', 4096, 11008), ('residual', 4096, 4096), ('none', 12288, 4096), ('residual', 512, 1408),
                                     ('none', 4096, 2048), ('residual', 1024, 768 * 4)])
def test_pipelined_gemv_equals_r2_kernels(epi, N, K, M):
    """r3 one-unit decode GEMVs (gemv_pipe_kernel: peeled k-loop with counted waits, activation fragments through a buffer
    descriptor, side bytes by one vector load + v_readlane) against the r2 kernels they replace (same K split): every output
    bit equal, on the bf16 stream and on the 12-bit stream, Gaussian and escape-heavy weights, ragged k-runs (K = 11008 -> 21 / 22
    k-tiles per wave; K = 1408 -> 2 / 3: falls back to the r2 kernel; K = 3072 -> 6 per wave)."""
    from oracle_amd import engine as E
    d = _dev()
    x = _rand((M, K), 61).to(d)
    res = _rand((M, N), 63).to(d) if epi == 'residual' else None
    for w in (_rand((N, K), 62, 0.02), _nasty_weights(N, K, 64)):
        w = w.to(d)
        for new, old in ((True, 'legacy'), ('cw', 'legacy_cw')):
            a = E.test_gemm(x, w, None, res, epi, use_gemv=new)
            b = E.test_gemm(x, w, None, res, epi, use_gemv=old)
            assert torch.equal(a.view(torch.int16), b.view(torch.int16)), (new, old)


@pytest.mark.parametrize('M', [1, 8, 32])
@pytest.mark.parametrize('N,K', [(12288, 4096), (22016, 4096), (32000, 4096), (1536, 768)])
def test_folded_rmsnorm_against_exact(N, K, M):
    """The decode path folds RMSNorm into the projection that consumes it (W' = bf16(W diag(w_norm)), row scale in the GEMV
    epilogue): one weight rounding more and one activation rounding fewer than the reference's eager RMSNorm -> Linear
    (DESIGN 3 'Changed rounding points'; ORC_DECODE_NORM=exact runs the reference's form). At the 7B projection shapes, with
    norm weights log-uniform over 0.01..4 (a trained model's range) and rows of very different scale, against fp64 math:
    (a) the folded form's RMS error is no larger than the exact form's (measured 0.8 x: it rounds the activation once, not twice),
    (b) its WORST element is no further from exact math than the exact form's worst element (x 1.25, both in units of the row's rms),
    (c) the two forms differ from each other by what two sets of independent 2^-9 input roundings over K terms must differ:
        max |folded - exact| <= 8 sigma of their combined error (large outputs carry larger roundings: not Gaussian). (VERDICT r2 asked for "within 2 bf16 ulps": measured 8.5 ulps of
        max(|y|, rms / 4) at the worst of 12288 x M outputs -- an output is a sum of K = 4096 rounded products, so two correct
        forms cannot agree to an output ulp; the bound that can hold is (a) - (c).)
    K = 768: a width whose 48 sum-of-squares partials are not a multiple of 16 per scale wave."""
    from oracle_amd import engine as E
    g = torch.Generator().manual_seed(100 + M)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(torch.bfloat16)
    wn = torch.exp(torch.empty(K).uniform_(math.log(0.01), math.log(4.0), generator=g)).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.02).to(torch.bfloat16)
    d = _dev()
    exact, folded = [t.cpu().double() for t in E.test_folded_norm(x.to(d), wn.to(d), w.to(d), 1e-5)]
    x64 = x.double()
    ref = (x64 * torch.rsqrt((x64 ** 2).mean(-1, keepdim=True) + 1e-5) * wn.double()) @ w.double().T
    rms = ref.pow(2).mean(-1, keepdim=True).sqrt()
    ee, ef = (exact - ref) / rms, (folded - ref) / rms
    e_exact, e_fold = ee.pow(2).mean().sqrt().item(), ef.pow(2).mean().sqrt().item()
    assert e_fold <= 1.1 * e_exact + 1e-6, (e_fold, e_exact)
    assert ef.abs().max().item() <= 1.25 * ee.abs().max().item(), (ef.abs().max().item(), ee.abs().max().item())
    sigma = math.sqrt(e_exact ** 2 + e_fold ** 2)
    assert ((folded - exact) / rms).abs().max().item() <= 8.0 * sigma, (((folded - exact) / rms).abs().max().item(), sigma)


def test_compressed_weight_stream_refuses_more_than_256_escapes_per_unit():
    """A unit (16 rows x K of a one-tile kernel) whose blocks need more than 254 non-zero patch entries cannot be compressed: the hook
    reports it (the engine then keeps that matrix in bf16, engine.hip compress_for_decode)."""
    from oracle_amd import engine as E
    d = _dev()
    w = (torch.randn(256, 4096) * 0.02).to(torch.bfloat16)
    w[:16, ::4] = (torch.randn(16, 1024) * 1e-12).to(torch.bfloat16)      # 1024 far-out-of-range values in unit 0
    x = _rand((4, 4096), 3).to(d)
    with pytest.raises(RuntimeError, match='not compressible'):
        E.test_gemm(x, w.to(d), None, None, 'none', use_gemv='cw')
