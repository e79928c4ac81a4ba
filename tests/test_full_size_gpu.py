"""GPU parity at the FULL size of the headline bench -- 32 LLaMA layers x 4096, 24-layer CLIP ViT-L/14-336, the learned
multi-view pooler, vocabulary 32000 -- against the oracle's own code (oracle/cpu_ref.py) executed by PyTorch-ROCm eager
on the same card. The CPU oracle needs ~30 s per frame at this size (profiles/r2/cpu_point.json), so the 7B depth was so far
covered only by bench.py's self-check; here the restatement's torch ops run on the GPU through rocBLAS / eager kernels:
that is, op for op, what the reference executes on a GPU box (torch eager bf16), and its fp32 run on the same bf16-valued
weights is the exact-math yardstick.

Bar (the rule of tests/test_engine_wide_gpu.py): at every step the engine's logits are within 1.5 x the reference's OWN
bf16-vs-fp32 distance `floor` (never below the north_star's 1e-2) of the reference's fp32 logits, in max|diff| / max|logit|
-- the engine is no further from exact math than the reference's stack is -- and therefore within (1 + 1.5) x floor of the
reference's bf16 logits (triangle inequality of the two distances; the eager run is not bit-reproducible across library
versions, so this side is not held tighter than what the first implies); the fp32 reference's greedy token is the engine's
argmax unless its margin is inside 2e-2 of max|logit|. All runs are teacher-forced along the fp32 reference's greedy tokens, so one flip cannot change later steps."""
import pytest
import torch

from conftest import rms_rel_rows

pytestmark = pytest.mark.gpu
TOL = 1e-2
STEPS = 6
RMS_FACTOR = 1.1    # per row and step: ||engine - fp32||_2 / ||fp32||_2 over ALL 32000 logits <= 1.1 x eager bf16's own (tests/test_engine_wide_gpu.py)


def _prompts(n, vocab):
    g = torch.Generator().manual_seed(91)
    rows = []
    for i in range(n):
        pre = torch.randint(3, vocab, (4 + 7 * i,), generator=g).tolist()
        post = torch.randint(3, vocab, (3 + 5 * (n - i),), generator=g).tolist()
        rows.append([1] + pre + [-200] + post)
    L = max(len(r) for r in rows)
    return torch.tensor([[0] * (L - len(r)) + r for r in rows])


def test_full_size_7b_prefill_and_decode_against_eager_reference_on_gpu():
    from oracle import cpu_ref as cr
    from oracle_amd.dims import FULL as d
    from oracle_amd.engine import Engine
    from oracle_amd.llava.model.language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM
    from oracle_amd.synth import synth_images, synth_state
    dev = torch.device('cuda', 0)
    torch.manual_seed(5)                                                    # device_fast weights come from the global generator
    B, V = 4, 2
    st = synth_state(d, seed=5, device=dev, device_fast=True)               # bf16 values
    ids = _prompts(B, d.vocab)
    imgs = synth_images(B, V, d, seed=808)
    eng = Engine(d, max_batch=B, max_seq=1280, max_images=B * V, max_new_tokens=16, device=dev)
    eng.load_state(st)
    model = LlavaLlamaForCausalLM(LlavaConfig(d, tokenizer_padding_side='left'), eng)

    # the reference's arithmetic, torch eager on the GPU: fp32 first (its greedy tokens define the common path), then bf16
    with torch.inference_mode(), torch.device(dev):
        st32 = {k: v.float() for k, v in st.items()}
        _, tr32 = cr.greedy_generate(st32, d, ids.to(dev), imgs.to(dev).float(), max_new_tokens=STEPS, eos_token_id=-1,
                                     return_trace=True, trace_logits_stride=1)
        del st32
        torch.cuda.empty_cache()
        path = torch.stack([t['argmax'] for t in tr32], dim=1)              # (B, STEPS)
        _, tr16 = cr.greedy_generate(st, d, ids.to(dev), imgs.to(dev), max_new_tokens=STEPS, eos_token_id=-1,
                                     forced_tokens=path, return_trace=True, trace_logits_stride=1)
    del st
    torch.cuda.empty_cache()

    mask = ids != 0
    pos = mask.long().cumsum(-1) - 1
    pos.masked_fill_(~mask, 1)
    out = model(input_ids=ids.to(dev), attention_mask=mask.to(dev), position_ids=pos.to(dev), images=imgs.to(dev), use_cache=True)
    cur = out.logits[:, -1].float()
    worst = (0.0, 0.0, 0.0)
    flips = 0
    rms_rows = []
    for step in range(STEPS):
        exact, gold = tr32[step]['logits_sub'].float(), tr16[step]['logits_sub'].float()
        r_eng, r_ref = rms_rel_rows(cur, exact), rms_rel_rows(gold, exact)      # all logits (trace stride 1), per row
        rms_rows.append((r_eng, r_ref))
        am = exact.abs().max(-1).values
        floor = ((gold - exact).abs().max(-1).values / am).max().item()
        band = 1.5 * max(floor, TOL / 1.5)
        err = ((cur - gold).abs().max(-1).values / am).max().item()
        err_exact = ((cur - exact).abs().max(-1).values / am).max().item()
        worst = max(worst, (err, err_exact, floor))
        assert err_exact <= band, (step, err_exact, floor)
        assert err <= band * 2.5 / 1.5, (step, err, floor)
        tok = path[:, step]
        slack = 2 * TOL * cur.abs().max(-1).values
        assert bool((cur.gather(1, tok[:, None])[:, 0] >= cur.max(-1).values - slack).all()), step
        flips += int((cur.argmax(-1) != tok).sum())
        if step + 1 < STEPS:
            cur = eng.decode_step(tok, want_logits=True).float()
    r_eng, r_ref = torch.stack([a for a, _ in rms_rows]), torch.stack([b for _, b in rms_rows])    # (STEPS, B)
    print(f'full size RMS-relative error vs eager fp32 over all {d.vocab} logits, per (step, row): engine median {float(r_eng.median()):.4f} / max '
          f'{float(r_eng.max()):.4f}; eager bf16 median {float(r_ref.median()):.4f} / max {float(r_ref.max()):.4f}; worst ratio engine / eager '
          f'{float((r_eng / r_ref).max()):.3f}, median ratio {float((r_eng / r_ref).median()):.3f}')
    print('  engine per row (max over steps):', [round(float(x), 4) for x in r_eng.max(0).values], ' eager bf16:', [round(float(x), 4) for x in r_ref.max(0).values])
    assert bool((r_eng <= RMS_FACTOR * r_ref).all()), (r_eng, r_ref)
    print(f'full size (32 layers, 24-layer CLIP, {B} rows x {V} views, T={int(mask.sum(1).max()) - 1 + 576}): worst engine vs eager bf16 '
          f'{worst[0]:.4f}, vs eager fp32 {worst[1]:.4f} (eager bf16-vs-fp32 floor {worst[2]:.4f}); engine argmax differs from the '
          f'fp32 token in {flips} of {B * STEPS} (row, step) pairs, all inside the margin')
    eng.close()


def _select_prompts(n, seed):
    g = torch.Generator().manual_seed(seed)
    rows = []
    for _ in range(n):
        k = int(torch.randint(3, 15, (1,), generator=g))
        rows.append([1] + torch.randint(3, 19000, (k,), generator=g).tolist() + [-200])
    L = max(len(r) for r in rows)
    return torch.tensor([[0] * (L - len(r)) + r for r in rows])


def test_full_size_7b_image_selected_answers_token_identical_to_eager_reference():
    """Free-running greedy generate() at the full 7B size on weights whose answer is selected by the image
    (oracle_amd/synth.py install_select_chains; the TINY version of this test runs against the real reference's golden
    tokens, tests/test_token_identity_gpu.py). Expected tokens: the oracle's greedy_generate executed in torch eager bf16 on
    the GPU; 48 frames in 3 ragged batches of 16, 3 views each. At this depth and with random weights the eager bf16 run
    itself sits ~5 % of max|logit| from its own fp32 run (`floor`, measured per frame), and picks the fp32 run's first token
    for 46 of 48 frames, so "identical to eager bf16" cannot be the bar for near-tie frames. Bar:
    (i) per frame, the engine's logits at the selecting step are within 1.5 x max(floor of this frame, median floor, 1e-2 / 1.5)
        of the fp32 logits -- the engine is no further from exact math than the reference is (this bounds every possible flip:
        a token the engine prefers over the eager one can trail it by at most twice that distance in exact math);
    (ii) a frame whose first token equals the eager run's is token-identical to it to the end (EOS and pad fill included);
    (iii) the engine agrees with the fp32 run's first token about as often as eager bf16 does (>= its count - 2).
    The identical-frame rate is printed."""
    import math

    from oracle import cpu_ref as cr
    from oracle_amd.dims import FULL as d
    from oracle_amd.engine import Engine
    from oracle_amd.llava.model.language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM
    from oracle_amd.synth import install_select_chains, synth_images, synth_state
    dev = torch.device('cuda', 0)
    torch.manual_seed(7)
    st = synth_state(d, seed=0, device=dev, device_fast=True)
    cal_ids = _select_prompts(8, 300).to(dev)
    cal_imgs = synth_images(8, 3, d, seed=6000).to(dev)
    with torch.inference_mode(), torch.device(dev):                          # calibration: mean last-position lm_head input
        feats = cr.encode_image_features(st, d, cal_imgs, 'learned')
        emb, mask, pos, _ = cr.splice(st, cal_ids, cal_ids != 0, feats)
        h, _ = cr.llama_forward(st, d, emb, mask, pos, return_hidden=True)
        center = h.float().mean(0).cpu()
        del feats, emb, h
    head = {k: st[k].cpu() for k in ('model.embed_tokens.weight', 'lm_head.weight')}
    boost = 4.0 + 3.0 * 0.0004 * d.d_model * math.sqrt(2 * d.n_layers)
    install_select_chains(head, d, center=center, boost=boost)
    for k, v in head.items():
        st[k] = v.to(dev)
    st32 = {k: v.float() for k, v in st.items()}
    eng = Engine(d, max_batch=16, max_seq=1792, max_images=48, max_new_tokens=32, device=dev)
    eng.load_state(st)
    model = LlavaLlamaForCausalLM(LlavaConfig(d, tokenizer_padding_side='left'), eng)
    same, n, unexplained, firsts, margins, floors, errs = 0, 0, [], set(), [], [], []
    first_like_fp32 = {'eager_bf16': 0, 'engine': 0}
    for b0 in range(0, 48, 16):
        ids = _select_prompts(16, 31000 + b0)
        imgs = synth_images(16, 3, d, seed=7000 + b0)
        with torch.inference_mode(), torch.device(dev):
            ref, tr = cr.greedy_generate(st, d, ids.to(dev), imgs.to(dev), max_new_tokens=24, return_trace=True, trace_logits_stride=1)
            _, tr32 = cr.greedy_generate(st32, d, ids.to(dev), imgs.to(dev).float(), max_new_tokens=1, return_trace=True,
                                         trace_logits_stride=1)
        ref = ref.cpu()
        l16, l32 = tr[0]['logits_sub'].float(), tr32[0]['logits_sub'].float()
        absmax = l32.abs().max(-1).values
        floor = ((l16 - l32).abs().max(-1).values / absmax).cpu()
        absmax = absmax.cpu()
        top2 = tr[0]['top2_val'].float().cpu()
        first32 = tr32[0]['argmax'].cpu()
        m0 = ids != 0
        p0 = m0.long().cumsum(-1) - 1
        p0.masked_fill_(~m0, 1)
        leng = model(input_ids=ids.to(dev), attention_mask=m0.to(dev), position_ids=p0.to(dev), images=imgs.to(dev),
                     use_cache=True).logits[:, -1].float()
        err = ((leng - l32).abs().max(-1).values / l32.abs().max(-1).values).cpu()
        out = model.generate(ids.to(dev), images=imgs.to(dev), do_sample=False, use_cache=True, max_new_tokens=24).cpu()
        w = max(ref.shape[1], out.shape[1])
        ref = torch.cat([ref, ref.new_zeros(16, w - ref.shape[1])], 1)
        out = torch.cat([out, out.new_zeros(16, w - out.shape[1])], 1)
        L = ids.shape[1]
        for i in range(16):
            n += 1
            firsts.add(int(ref[i, L]))
            margin = float(top2[i, 0] - top2[i, 1])
            margins.append(margin)
            floors.append(float(floor[i]))
            first_like_fp32['eager_bf16'] += int(ref[i, L] == first32[i])
            first_like_fp32['engine'] += int(out[i, L] == first32[i])
            errs.append((float(err[i]), float(floor[i])))
            if torch.equal(out[i], ref[i]):
                same += 1
            elif out[i, L] == ref[i, L]:
                unexplained.append((b0 + i, out[i, L:].tolist(), ref[i, L:].tolist(), margin))
    margins.sort()
    floors.sort()
    print(f'full size, image-selected answers: token-identical frames {same} / {n} = {same / n:.4f}; distinct first tokens '
          f'{len(firsts)}; eager bf16 first-step margin min {margins[0]:.3f} / median {margins[len(margins) // 2]:.2f}; eager '
          f'bf16-vs-fp32 floor median {floors[len(floors) // 2]:.4f} / max {floors[-1]:.4f}; first token equals the fp32 run\'s: '
          f'eager bf16 {first_like_fp32["eager_bf16"]} / {n}, engine {first_like_fp32["engine"]} / {n}')
    mine = sorted(e for e, _ in errs)
    q = lambda v, f: v[min(len(v) - 1, int(f * len(v)))]
    print(f'engine vs fp32 at the selecting step: median {q(mine, 0.5):.4f} / p90 {q(mine, 0.9):.4f} / max {mine[-1]:.4f}; eager bf16 '
          f'vs fp32: median {q(floors, 0.5):.4f} / p90 {q(floors, 0.9):.4f} / max {floors[-1]:.4f}')
    assert q(mine, 0.5) <= 1.5 * max(q(floors, 0.5), TOL / 1.5)
    assert q(mine, 0.9) <= 1.5 * max(q(floors, 0.9), TOL / 1.5)
    assert not unexplained, unexplained[:3]
    assert first_like_fp32['engine'] >= first_like_fp32['eager_bf16'] - 2
    assert len(firsts) >= 3
    eng.close()


def _class_frames(n, seed, base, eps):
    """n frames, each = one of the base images (its class) + eps x fresh pixel noise, clipped as synth_images clips."""
    g = torch.Generator().manual_seed(seed)
    cls = torch.randint(0, base.shape[0], (n,), generator=g)
    noise = torch.randn((n,) + tuple(base.shape[1:]), generator=g)
    return (base[cls].float() + eps * noise).clamp(-1.8, 2.2).to(torch.bfloat16), cls


RATE_FRAMES = 256
RATE_CLASSES = 4


def test_full_size_7b_token_identity_rate_256_frames():
    """north_star's ">= 99 % token identity", asserted at the FULL 7B size (VERDICT r2 item 2 / missing 4) on a population whose
    answer is decided by the image with a margin far above the noise of a 32-layer bf16 stack -- as a trained model's answers
    are. 256 frames (16 ragged batches of 16, 3 views, random prompts ending in <image>): every frame shows one of four image
    CLASSES (a base image + 25 % fresh pixel noise), the first answer token is argmax_k <h, class-k displacement of the last
    hidden state> (install_select_chains(dirs=...), calibrated on the card with the oracle's code on 32 other frames), the
    answer then runs along that class's chain (6 + 3k tokens, EOS, pad fill). With random directions (the 48-frame test above)
    signal and rounding noise are both spread over d_model dims and ~4 % of the frames are near ties in which eager bf16
    contradicts its OWN fp32 run; here the whole signal lies along the direction.
    Population property (asserted, independent of the engine): on every frame eager bf16 picks the first token of its own fp32
    run, that token is the frame's class, and the fp32 top-2 margin is > 4 x the eager bf16-vs-fp32 logit distance.
    Bar: engine tokens == eager bf16 tokens (the oracle's code on the card) on >= 99 % of the frames; expected 256 / 256."""
    import math

    from oracle import cpu_ref as cr
    from oracle_amd.dims import FULL as d
    from oracle_amd.engine import Engine
    from oracle_amd.llava.model.language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM
    from oracle_amd.synth import install_select_chains, select_chain_layout, synth_images, synth_state
    dev = torch.device('cuda', 0)
    torch.manual_seed(11)
    st = synth_state(d, seed=0, device=dev, device_fast=True)
    base = synth_images(RATE_CLASSES, 3, d, seed=9100).float()
    # calibration on the card: class means of the last-position lm_head input over 32 frames of the same construction
    hs, cs = [], []
    for j in range(2):
        cal_ids = _select_prompts(16, 400 + j).to(dev)
        cal_imgs, cal_cls = _class_frames(16, 500 + j, base, 0.25)
        with torch.inference_mode(), torch.device(dev):
            feats = cr.encode_image_features(st, d, cal_imgs.to(dev), 'learned')
            emb, mask, pos, _ = cr.splice(st, cal_ids, cal_ids != 0, feats)
            h, _ = cr.llama_forward(st, d, emb, mask, pos, return_hidden=True)
        hs.append(h.float().cpu())
        cs.append(cal_cls)
        del feats, emb, h
    hs, cs = torch.cat(hs), torch.cat(cs)
    assert all(int((cs == k).sum()) >= 3 for k in range(RATE_CLASSES)), cs
    means = torch.stack([hs[cs == k].mean(0) for k in range(RATE_CLASSES)])
    center = means.mean(0)
    disp = means - center
    spread = torch.stack([(hs[cs == k] - means[k]).norm(dim=-1).mean() for k in range(RATE_CLASSES)])
    print(f'calibration: |h| {float(hs.norm(dim=-1).mean()):.1f}, class displacement norms {[round(float(x), 1) for x in disp.norm(dim=-1)]}, '
          f'within-class spread {[round(float(x), 1) for x in spread]}')
    head = {k: st[k].cpu() for k in ('model.embed_tokens.weight', 'lm_head.weight')}
    boost = 4.0 + 3.0 * 0.0004 * d.d_model * math.sqrt(2 * d.n_layers)
    install_select_chains(head, d, center=center, boost=boost, dirs=disp, g_sel=60.0 / float(disp.norm(dim=-1).mean()))
    for k, v in head.items():
        st[k] = v.to(dev)
    st32 = {k: v.float() for k, v in st.items()}
    starts = [lo for lo, _ in select_chain_layout()]
    eng = Engine(d, max_batch=16, max_seq=1792, max_images=48, max_new_tokens=32, device=dev)
    eng.load_state(st)
    model = LlavaLlamaForCausalLM(LlavaConfig(d, tokenizer_padding_side='left'), eng)
    same, n, per_class, ratios, bad_pop, diffs = 0, 0, [0] * RATE_CLASSES, [], [], []
    for b0 in range(0, RATE_FRAMES, 16):
        ids = _select_prompts(16, 41000 + b0)
        imgs, cls = _class_frames(16, 8000 + b0, base, 0.25)
        with torch.inference_mode(), torch.device(dev):
            ref, tr = cr.greedy_generate(st, d, ids.to(dev), imgs.to(dev), max_new_tokens=24, return_trace=True, trace_logits_stride=1)
            _, tr32 = cr.greedy_generate(st32, d, ids.to(dev), imgs.to(dev).float(), max_new_tokens=1, return_trace=True,
                                         trace_logits_stride=1)
        ref = ref.cpu()
        l16, l32 = tr[0]['logits_sub'].float(), tr32[0]['logits_sub'].float()
        dist = (l16 - l32).abs().max(-1).values.cpu()
        top32 = tr32[0]['top2_val'].float().cpu()
        first32, first16 = tr32[0]['argmax'].cpu(), tr[0]['argmax'].cpu()
        out = model.generate(ids.to(dev), images=imgs.to(dev), do_sample=False, use_cache=True, max_new_tokens=24).cpu()
        w = max(ref.shape[1], out.shape[1])
        ref = torch.cat([ref, ref.new_zeros(16, w - ref.shape[1])], 1)
        out = torch.cat([out, out.new_zeros(16, w - out.shape[1])], 1)
        for i in range(16):
            n += 1
            k = int(cls[i])
            per_class[k] += 1
            ratios.append(float((top32[i, 0] - top32[i, 1]) / dist[i]))
            if not (int(first16[i]) == int(first32[i]) == starts[k]) or ratios[-1] <= 4.0:
                bad_pop.append((b0 + i, k, int(first16[i]), int(first32[i]), ratios[-1]))
            if torch.equal(out[i], ref[i]):
                same += 1
            else:
                diffs.append((b0 + i, k, out[i, ids.shape[1]:].tolist(), ref[i, ids.shape[1]:].tolist()))
        if (b0 // 16) % 4 == 3:
            print(f'  {n} frames: identical {same}', flush=True)
    ratios.sort()
    print(f'full size, class-selected answers: token-identical frames {same} / {n} = {same / n:.4f}; frames per class {per_class}; '
          f'fp32 top-2 margin / eager bf16-vs-fp32 logit distance: min {ratios[0]:.1f}, median {ratios[len(ratios) // 2]:.1f}')
    assert n == RATE_FRAMES and min(per_class) >= RATE_FRAMES // (4 * RATE_CLASSES), per_class
    assert not bad_pop, ('population is not decided: eager bf16 contradicts its own fp32 run / the class', bad_pop[:4])
    assert same / n >= 0.99, diffs[:3]
    eng.close()
