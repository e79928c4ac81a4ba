"""GPU: the released-checkpoint path end to end (reference LLaVA/llava/model/builder.py:50-91,150-178):
load_pretrained_model(<lora dir>, <base dir>) -> prefix handling + LoRA merge + CLIP remap + strict pooler load -> native
engine -> generate(). Tokens must equal an engine loaded directly from the merged state and the CPU oracle on that state.
The merge itself is pinned by its formula only (peft 0.4.0 is not importable offline: merge-vs-peft parity unpinned)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_released_layout_loads_into_the_engine_and_generates(tmp_path):
    from oracle import cpu_ref as cr
    from oracle_amd.dims import TINY
    from oracle_amd.engine import Engine
    from oracle_amd.llava.model import checkpoint as ck
    from oracle_amd.llava.model.builder import load_pretrained_model
    from oracle_amd.llava.model.language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM
    from oracle_amd.synth import CHAIN_BASE, CHAIN_LEN, install_chain, synth_images, synth_state
    from release_layout import write_release
    full = install_chain(synth_state(TINY, seed=3), TINY)
    _, base_dir, lora_dir, adapter, *_ = write_release(str(tmp_path), TINY, full=full)
    tok, model, proc, ctx = load_pretrained_model(lora_dir, base_dir, 'oracle_lora_ckpt', max_batch=2, max_seq=768, max_images=6,
                                                  max_new_tokens=32)
    model.config.tokenizer_padding_side = 'left'
    assert model.config.mv_type == 'learned' and model.engine.dims == TINY
    merged, _ = ck.assemble_state(lora_dir, base_dir)
    assert not torch.equal(merged['model.layers.0.self_attn.q_proj.weight'], full['model.layers.0.self_attn.q_proj.weight'])
    e = CHAIN_BASE + CHAIN_LEN
    ids = torch.tensor([[0, 0, 1, 5, -200, 7, e - 6], [1, 9, 10, 11, -200, 12, e - 9]])
    imgs = synth_images(2, 3, TINY, seed=31)
    out = model.generate(ids.cuda(), images=imgs.cuda(), do_sample=False, use_cache=True, max_new_tokens=16).cpu()
    ref = cr.greedy_generate(merged, TINY, ids, imgs, max_new_tokens=16)
    assert torch.equal(out, ref), (out[:, ids.shape[1]:], ref[:, ids.shape[1]:])
    # prefill logits of the directory-loaded engine == an engine fed the merged state directly, bit for bit
    mask = ids != 0
    a = model(input_ids=ids.cuda(), attention_mask=mask.cuda(), images=imgs.cuda()).logits.cpu()
    eng2 = Engine(TINY, max_batch=2, max_seq=768, max_images=6, max_new_tokens=32)
    eng2.load_state(merged)
    model2 = LlavaLlamaForCausalLM(LlavaConfig(TINY, tokenizer_padding_side='left'), eng2)
    b = model2(input_ids=ids.cuda(), attention_mask=mask.cuda(), images=imgs.cuda()).logits.cpu()
    assert torch.equal(a, b)
    # and it is the merged weights that run: the un-merged base gives different logits
    eng3 = Engine(TINY, max_batch=2, max_seq=768, max_images=6, max_new_tokens=32)
    eng3.load_state(full)
    c = LlavaLlamaForCausalLM(LlavaConfig(TINY, tokenizer_padding_side='left'), eng3)(
        input_ids=ids.cuda(), attention_mask=mask.cuda(), images=imgs.cuda()).logits.cpu()
    assert not torch.equal(a, c)
    for x in (model.engine, eng2, eng3):
        x.close()
