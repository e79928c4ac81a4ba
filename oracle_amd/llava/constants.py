"""Model constants of the hot path (values: reference LLaVA/llava/constants.py:7-15)."""
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
VIS_DESCRIPTOR_TOKEN = "***"
VIS_DESCRIPTOR_TOKEN_INDEX = 18610
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"
IMAGE_PLACEHOLDER = "<image-placeholder>"
