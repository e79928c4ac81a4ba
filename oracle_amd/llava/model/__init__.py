from .language_model.llava_llama import LlavaConfig, LlavaLlamaForCausalLM  # noqa: F401
