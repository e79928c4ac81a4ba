"""GPU: continuous batching (generate_stream / orc_stream_begin + orc_prefill_slots + orc_stream_attach + orc_read_slots): a
stream of ragged batches through n decode rows over the engine's 8 KV-cache slots (next batch prefilled ahead into free
slots, finished rows re-attached) must return, batch by batch, exactly what generate() returns for that batch alone (and
what the CPU oracle returns): per-slot cached lengths, frozen finished rows, per-row length stops."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def chain():
    from oracle_amd.dims import TINY
    from oracle_amd.engine import Engine
    from oracle_amd.llava.model.language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM
    from oracle_amd.synth import install_chain, synth_state
    st = install_chain(synth_state(TINY, seed=0), TINY)
    eng = Engine(TINY, max_batch=8, max_seq=700, max_images=32, max_new_tokens=48)
    eng.load_state(st)
    model = LlavaLlamaForCausalLM(LlavaConfig(TINY, tokenizer_padding_side='left'), eng)
    yield st, eng, model
    eng.close()


def _batches(sizes, seed):
    import random
    from oracle_amd.dims import TINY
    from oracle_amd.synth import CHAIN_BASE, CHAIN_LEN, synth_images
    rng = random.Random(seed)
    e = CHAIN_BASE + CHAIN_LEN
    out = []
    for k, b in enumerate(sizes):
        rows = []
        for i in range(b):
            pre = [rng.randrange(3, 1000) for _ in range(rng.randint(0, 9))]
            post = [rng.randrange(3, 1000) for _ in range(rng.randint(0, 6))]
            rows.append([1] + pre + [-200] + post + [e - rng.randint(2, 30)])      # 2..30 chain tokens, then EOS
        L = max(len(r) for r in rows)
        ids = torch.tensor([[0] * (L - len(r)) + r for r in rows])
        out.append({'input_ids': ids.cuda(), 'images': synth_images(b, 3, TINY, seed=300 + 17 * k + seed).cuda()})
    return out


@pytest.mark.parametrize('n_rows,chunk', [(4, 4), (8, 8), (1, 3), (3, 5)])
def test_stream_equals_generate_batch_by_batch(chain, n_rows, chunk):
    from oracle import cpu_ref as cr
    from oracle_amd.dims import TINY
    st, eng, model = chain
    items = _batches([3, 8, 1, 5, 2, 8], seed=n_rows)
    outs = [o.cpu() for o in model.generate_stream(iter(items), max_new_tokens=40, n_rows=n_rows, chunk=chunk)]
    assert len(outs) == len(items)
    for k, (item, out) in enumerate(zip(items, outs)):
        ref = model.generate(item['input_ids'], images=item['images'], do_sample=False, use_cache=True, max_new_tokens=40).cpu()
        assert torch.equal(out, ref), (k, out[:, item['input_ids'].shape[1]:], ref[:, item['input_ids'].shape[1]:])
    for k in (0, 3):                                                              # and the CPU oracle
        ref = cr.greedy_generate(st, TINY, items[k]['input_ids'].cpu(), items[k]['images'].cpu(), max_new_tokens=40)
        assert torch.equal(outs[k], ref)


def test_stream_length_stops_and_capacity(chain):
    from oracle_amd import _lib
    st, eng, model = chain
    items = _batches([4, 4], seed=9)
    items[0]['limits'] = [1, 3, 40, 2]
    outs = [o.cpu() for o in model.generate_stream(iter(items), max_new_tokens=40, n_rows=3, chunk=2)]
    full = model.generate(items[0]['input_ids'], images=items[0]['images'], max_new_tokens=40).cpu()
    L = items[0]['input_ids'].shape[1]
    for b, lim in enumerate(items[0]['limits']):
        n = min(lim, int((full[b, L:] != 0).sum()))                               # the stop, or the row's own EOS if earlier
        assert torch.equal(outs[0][b, L:L + n], full[b, L:L + n])
        assert bool((outs[0][b, L + n:] == 0).all())
    assert torch.equal(outs[1], model.generate(items[1]['input_ids'], images=items[1]['images'], max_new_tokens=40).cpu())
    # the C ABI refuses a slot outside the engine, a slot that is decoding, a stop beyond the token buffer, a prompt that cannot fit
    eng.stream_begin(2)
    emb = torch.zeros(1, 10, 512, dtype=torch.bfloat16, device='cuda')
    pos = torch.arange(10, dtype=torch.int32)[None]
    with pytest.raises(_lib.OracleHipError, match='slot outside'):
        eng.prefill_slots([8], emb, pos, [0], [4])
    with pytest.raises(_lib.OracleHipError, match='length stop'):
        eng.prefill_slots([0], emb, pos, [0], [49])
    with pytest.raises(_lib.OracleHipError, match='exceeds max_seq'):
        eng.prefill_slots([0], torch.zeros(1, 690, 512, dtype=torch.bfloat16, device='cuda'), torch.arange(690, dtype=torch.int32)[None], [0], [20])
    eng.prefill_slots([5], emb, pos, [0], [4])
    eng.stream_attach(1, 5)
    with pytest.raises(_lib.OracleHipError, match='attached'):
        eng.prefill_slots([5], emb, pos, [0], [4])
    with pytest.raises(_lib.OracleHipError, match='attached to another row'):
        eng.stream_attach(0, 5)
    with pytest.raises(_lib.OracleHipError, match='teacher forcing'):
        eng.decode_step(torch.tensor([5, 5]).cuda())
    eng.decode_steps(6)                                                             # row 1 <- slot 5 runs to its stop of 4 tokens
    toks, steps, unf = eng.read_slots()
    assert steps[5] == 4 and unf[5] == 0 and len(steps) == 8


def test_stream_mixes_batches_of_very_different_prompt_lengths(chain):
    """Incoming batches whose prompts differ by ~100 tokens share stream prefills (a short frame is then left-padded far beyond its
    own batch's length: different pad offset, different tile alignment, different co-batched rows than in generate()): the
    tokens must still be generate()'s for every batch. Tokens, not logits bits, are what this guarantees."""
    import random
    from oracle_amd.dims import TINY
    from oracle_amd.synth import CHAIN_BASE, CHAIN_LEN, synth_images
    st, eng, model = chain
    rng = random.Random(77)
    e = CHAIN_BASE + CHAIN_LEN
    items = []
    for k, (b, extra) in enumerate([(3, 0), (2, 110), (4, 5), (1, 64), (3, 0)]):
        rows = [[1] + [rng.randrange(3, 1000) for _ in range(extra + rng.randint(0, 4))] + [-200] + [rng.randrange(3, 1000) for _ in range(rng.randint(0, 3))]
                + [e - rng.randint(2, 24)] for _ in range(b)]
        L = max(len(r) for r in rows)
        ids = torch.tensor([[0] * (L - len(r)) + r for r in rows])
        items.append({'input_ids': ids.cuda(), 'images': synth_images(b, 3, TINY, seed=900 + k).cuda()})
    outs = [o.cpu() for o in model.generate_stream(iter(items), max_new_tokens=32, n_rows=4, chunk=3)]
    for k, (item, out) in enumerate(zip(items, outs)):
        ref = model.generate(item['input_ids'], images=item['images'], max_new_tokens=32).cpu()
        assert torch.equal(out, ref), k


def test_generate_row_limit_of_one_emits_one_token(chain):
    """orc_set_row_limits with a limit of 1: the prefill's own token is the row's only one (r2 emitted a second token: ADVICE)."""
    st, eng, model = chain
    item = _batches([4], seed=5)[0]
    L = item['input_ids'].shape[1]
    free = model.generate(item['input_ids'], images=item['images'], max_new_tokens=20).cpu()
    out = model.generate(item['input_ids'], images=item['images'], max_new_tokens=20, limits=[1, 20, 1, 3]).cpu()
    for b, lim in enumerate([1, 20, 1, 3]):
        n = min(lim, int((free[b, L:] != 0).sum()))
        assert torch.equal(out[b, L:L + n], free[b, L:L + n]), b
        assert bool((out[b, L + n:] == 0).all()), (b, out[b, L:])


def test_stream_16_rows_over_32_slots_equals_generate():
    """Decode width 16 over 32 KV-cache slots (two of the reference's batches of 8 decode side by side, the next two wait
    prefilled): still exactly generate()'s tokens for every incoming batch."""
    from oracle_amd.dims import TINY
    from oracle_amd.engine import Engine
    from oracle_amd.llava.model.language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM
    from oracle_amd.synth import install_chain, synth_state
    st = install_chain(synth_state(TINY, seed=0), TINY)
    eng = Engine(TINY, max_batch=32, max_seq=700, max_images=128, max_new_tokens=48)
    eng.load_state(st)
    model = LlavaLlamaForCausalLM(LlavaConfig(TINY, tokenizer_padding_side='left'), eng)
    items = _batches([8, 8, 8, 5, 8, 3, 8], seed=21)
    outs = [o.cpu() for o in model.generate_stream(iter(items), max_new_tokens=40, chunk=4)]          # default width: 16 rows
    refs = [model.generate(it['input_ids'], images=it['images'], max_new_tokens=40).cpu() for it in items]
    for k, (o, r) in enumerate(zip(outs, refs)):
        assert torch.equal(o, r), k
    outs32 = [o.cpu() for o in model.generate_stream(iter(items), max_new_tokens=40, n_rows=32, chunk=4)]   # 32 rows: all slots decode
    for k, (o, r) in enumerate(zip(outs32, refs)):
        assert torch.equal(o, r), k
    with pytest.raises(Exception, match='32'):
        model.generate(torch.cat([items[0]['input_ids']] * 5), images=torch.cat([items[0]['images']] * 5), max_new_tokens=4)
    eng.close()
