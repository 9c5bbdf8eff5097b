from videoswap_amd.attention import EDLoRA_AttnProcessor  # noqa: F401
from videoswap_amd.edlora import bind_concept_prompt, encode_edlora_prompt, revise_edlora_unet_attention_forward  # noqa: F401
