"""Host-side mirror of the reference's LLaVA package for the inference hot path (same module names,
function names and argument meaning as /root/reference/LLaVA/llava/...), backed by the native engine."""
