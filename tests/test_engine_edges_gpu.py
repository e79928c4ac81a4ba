"""GPU edge cases of the hot path through the mirrored API: the largest batch the engine takes (16 rows, ragged left
padding, rows finishing at different steps) against the CPU oracle and against the same rows run alone (batch
invariance); a text-only prompt; capacity / argument errors raised the way the mirror documents them."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def chain16():
    from oracle_amd.dims import TINY
    from oracle_amd.engine import Engine
    from oracle_amd.llava.model.language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM
    from oracle_amd.synth import install_chain, synth_state
    st = install_chain(synth_state(TINY, seed=0), TINY)
    eng = Engine(TINY, max_batch=16, max_seq=700, max_images=48, max_new_tokens=48)
    eng.load_state(st)
    model = LlavaLlamaForCausalLM(LlavaConfig(TINY, tokenizer_padding_side='left'), eng)
    yield st, eng, model
    eng.close()


def _prompts(n):
    from oracle_amd.synth import CHAIN_BASE, CHAIN_LEN
    e = CHAIN_BASE + CHAIN_LEN
    g = torch.Generator().manual_seed(3)
    rows = []
    for i in range(n):
        pre = torch.randint(3, 1000, (1 + i % 5,), generator=g).tolist()
        post = torch.randint(3, 1000, (i % 4,), generator=g).tolist()
        rows.append([1] + pre + [-200] + post + [e - 3 - 2 * i])          # row i emits 3 + 2i tokens, then EOS
    L = max(len(r) for r in rows)
    return torch.tensor([[0] * (L - len(r)) + r for r in rows])


def test_batch_of_16_ragged_rows_matches_oracle_and_single_rows(chain16):
    from oracle import cpu_ref as cr
    from oracle_amd.dims import TINY
    from oracle_amd.synth import synth_images
    st, eng, model = chain16
    ids = _prompts(16)
    imgs = synth_images(16, 3, TINY, seed=77)
    out = model.generate(ids.cuda(), images=imgs.cuda(), do_sample=False, use_cache=True, max_new_tokens=40).cpu()
    ref = cr.greedy_generate(st, TINY, ids, imgs, max_new_tokens=40)
    assert torch.equal(out, ref)
    new = out[:, ids.shape[1]:]
    for i in (0, 7, 15):                                                    # EOS after 3 + 2i tokens, pad afterwards
        n = 3 + 2 * i
        assert new[i, n].item() == 2 and (n + 1 >= new.shape[1] or bool((new[i, n + 1:] == 0).all()))
    # the same rows alone (no padding, batch 1): identical tokens up to their EOS
    for i in (0, 9, 15):
        row = ids[i][ids[i] != 0][None]
        one = model.generate(row.cuda(), images=imgs[i:i + 1].cuda(), max_new_tokens=40).cpu()
        n = 3 + 2 * i + 1
        assert torch.equal(one[0, row.shape[1]:row.shape[1] + n], new[i, :n])


def test_text_only_prompt_and_batch_one(chain16):
    from oracle import cpu_ref as cr
    from oracle_amd.dims import TINY
    from oracle_amd.synth import CHAIN_BASE, CHAIN_LEN
    st, eng, model = chain16
    ids = torch.tensor([[1, 17, 18, 19, CHAIN_BASE + CHAIN_LEN - 6]])
    out = model.generate(ids.cuda(), images=None, max_new_tokens=12).cpu()
    emb = st['model.embed_tokens.weight'][ids]
    logits, _ = cr.llama_forward(st, TINY, emb, torch.ones_like(ids, dtype=torch.bool), torch.arange(ids.shape[1])[None])
    assert out[0, ids.shape[1]].item() == int(logits.argmax(-1))
    assert out[0, -1].item() == 2 and out.shape[1] == ids.shape[1] + 7


def test_capacity_and_argument_errors(chain16):
    from oracle_amd import _lib
    from oracle_amd.dims import TINY
    from oracle_amd.synth import synth_images
    st, eng, model = chain16
    ids = _prompts(2)
    imgs = synth_images(2, 3, TINY, seed=5).cuda()
    with pytest.raises(ValueError, match='max_new_tokens'):
        model.generate(ids.cuda(), images=imgs, max_new_tokens=49)
    from oracle_amd.synth import CHAIN_BASE, CHAIN_LEN
    ends = torch.tensor([[CHAIN_BASE + CHAIN_LEN - 30], [CHAIN_BASE + CHAIN_LEN - 35]])        # >= 30 tokens before EOS
    long_ids = torch.cat([ids[:, :-1], ids.new_full((2, 100), 5), ends], 1)                    # T = 681 of 700 slots
    with pytest.warns(UserWarning, match='KV capacity'):
        out = model.generate(long_ids.cuda(), images=imgs, max_new_tokens=40)
    assert out.shape[1] - long_ids.shape[1] == 700 - (long_ids.shape[1] - 1 + 576)             # clamped to the free slots
    with pytest.raises(ValueError, match='KV capacity'):
        model.generate(torch.cat([ids, ids.new_full((2, 130), 5)], 1).cuda(), images=imgs, max_new_tokens=4)
    # the C ABI itself refuses a decode step that would write past the cache (K/V slot = cur_len - 1 was unchecked)
    t = long_ids.shape[1] - 1 + 576
    o = model(input_ids=long_ids.cuda(), attention_mask=(long_ids != 0).cuda(), images=imgs)
    eng.decode_steps(700 - t - 2)
    eng.decode_step(torch.tensor([5, 5]).cuda())
    eng.decode_step(torch.tensor([5, 5]).cuda())                                               # slot 699: the last one
    with pytest.raises(_lib.OracleHipError, match='KV cache full'):
        eng.decode_step(torch.tensor([5, 5]).cuda())
    with pytest.raises(_lib.OracleHipError, match='KV cache full'):
        eng.decode_steps(1)
    with pytest.raises(_lib.OracleHipError, match='KV cache full'):
        model(input_ids=torch.tensor([[5], [5]]).cuda(), past_key_values=o.past_key_values)
    with pytest.raises(NotImplementedError):
        model.generate(ids.cuda(), images=imgs, do_sample=True)
    right_padded = torch.tensor([[1, 5, -200, 6, 0, 0], [1, 5, -200, 6, 7, 8]])
    # the splice re-pads to the configured side (llava_arch.py:338-359); the native LLaMA path refuses right padding
    model.config.tokenizer_padding_side = 'right'
    try:
        with pytest.raises(NotImplementedError, match='left-padded'):
            model(input_ids=right_padded.cuda(), attention_mask=(right_padded != 0).cuda(), images=imgs)
    finally:
        model.config.tokenizer_padding_side = 'left'
    model(input_ids=right_padded.cuda(), attention_mask=(right_padded != 0).cuda(), images=imgs)         # re-padded left: fine
    with pytest.raises(_lib.OracleHipError):
        eng.pool_views(torch.zeros(5 * 3, 576, 1024, dtype=torch.bfloat16, device='cuda'), [5, 5, 5])   # 5 views > pooler's 2304 positions
    with pytest.raises(ValueError):
        eng.vision_encode(torch.zeros(1, 3, 336, 336, device='cuda'))                                    # fp32 images are refused


@pytest.mark.parametrize('seed', [0, 1, 2, 3])
def test_random_ragged_batches_match_oracle(chain16, seed):
    """Seeded random batch shapes (1..16 rows, 1 / 3 / 4 views, ragged text around the image placeholder, rows ending at
    different steps): generate() must return exactly the CPU oracle's tokens, pad fill included."""
    import random
    from oracle import cpu_ref as cr
    from oracle_amd.dims import TINY
    from oracle_amd.synth import CHAIN_BASE, CHAIN_LEN, synth_images
    st, eng, model = chain16
    rng = random.Random(100 + seed)
    b = rng.choice([1, 2, 3, 5, 8, 11, 16])
    views = rng.choice([1, 3, 4])
    e = CHAIN_BASE + CHAIN_LEN
    rows = []
    for i in range(b):
        pre = [rng.randrange(3, 1000) for _ in range(rng.randint(0, 9))]
        post = [rng.randrange(3, 1000) for _ in range(rng.randint(0, 6))]
        rows.append([1] + pre + [-200] + post + [e - rng.randint(2, 30)])     # 2..30 chain tokens, then EOS
    L = max(len(r) for r in rows)
    ids = torch.tensor([[0] * (L - len(r)) + r for r in rows])
    imgs = synth_images(b, views, TINY, seed=900 + seed)
    mv = 'learned' if views > 1 else 'concat'
    model.config.mv_type = mv
    try:
        out = model.generate(ids.cuda(), images=imgs.cuda(), do_sample=False, use_cache=True, max_new_tokens=40).cpu()
    finally:
        model.config.mv_type = 'learned'
    ref = cr.greedy_generate(st, TINY, ids, imgs, mv_type=mv, max_new_tokens=40)
    assert torch.equal(out, ref), (b, views)


def test_batch_of_32_ragged_rows_matches_oracle():
    """32 rows in one generate(): two MFMA B-operand tiles per weight fragment in every decode GEMV, 32-row prefill, sampler
    and greedy state over 32 slots. Tokens must equal the CPU oracle's, pad fill included, and rows 16..31 must equal the
    same rows run as a batch of 16."""
    from oracle import cpu_ref as cr
    from oracle_amd.dims import TINY
    from oracle_amd.engine import Engine
    from oracle_amd.llava.model.language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM
    from oracle_amd.synth import install_chain, synth_images, synth_state
    st = install_chain(synth_state(TINY, seed=0), TINY)
    eng = Engine(TINY, max_batch=32, max_seq=700, max_images=96, max_new_tokens=72)
    eng.load_state(st)
    model = LlavaLlamaForCausalLM(LlavaConfig(TINY, tokenizer_padding_side='left'), eng)
    ids = _prompts(32)
    imgs = synth_images(32, 3, TINY, seed=78)
    out = model.generate(ids.cuda(), images=imgs.cuda(), do_sample=False, use_cache=True, max_new_tokens=70).cpu()
    ref = cr.greedy_generate(st, TINY, ids, imgs, max_new_tokens=70)
    assert torch.equal(out, ref)
    half = model.generate(ids[16:].cuda(), images=imgs[16:].cuda(), max_new_tokens=70).cpu()
    L = ids.shape[1]
    n = min(half.shape[1], out.shape[1])
    assert torch.equal(half[:, L:n], out[16:, L:n])
    eng.close()


def test_compressed_decode_weights_give_the_same_bits(monkeypatch):
    """The engine streams lossless 12-bit copies of the decode weights (gemv.hip; default) -- against the same engine with
    ORC_DECODE_WEIGHTS=plain (bf16 stream): prefill logits, the logits of 6 decode steps and the free-running tokens of a
    ragged batch are equal bit for bit."""
    from oracle_amd.dims import TINY
    from oracle_amd.engine import Engine
    from oracle_amd.llava.model.language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM
    from oracle_amd.synth import install_chain, synth_images, synth_state
    st = install_chain(synth_state(TINY, seed=0), TINY)
    ids = _prompts(9)
    imgs = synth_images(9, 3, TINY, seed=91)
    got = {}
    for mode in ('plain', 'cw12'):
        monkeypatch.setenv('ORC_DECODE_WEIGHTS', mode)
        eng = Engine(TINY, max_batch=9, max_seq=700, max_images=27, max_new_tokens=40)
        eng.load_state(st)
        model = LlavaLlamaForCausalLM(LlavaConfig(TINY, tokenizer_padding_side='left'), eng)
        toks = model.generate(ids.cuda(), images=imgs.cuda(), do_sample=False, use_cache=True, max_new_tokens=30).cpu()
        mask = ids != 0
        pos = mask.long().cumsum(-1) - 1
        pos.masked_fill_(~mask, 1)
        logits = [model(input_ids=ids.cuda(), attention_mask=mask.cuda(), position_ids=pos.cuda(), images=imgs.cuda(),
                        use_cache=True).logits[:, -1].float().cpu()]
        for step in range(6):
            logits.append(eng.decode_step(logits[-1].argmax(-1).cuda(), want_logits=True).float().cpu())
        got[mode] = (toks, torch.stack(logits))
        eng.close()
    assert torch.equal(got['plain'][0], got['cw12'][0])
    assert torch.equal(got['plain'][1].view(torch.int32), got['cw12'][1].view(torch.int32))
