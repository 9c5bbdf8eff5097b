"""Generate tests/golden/processors.pt and edlora.pt from the REFERENCE's own code (build container only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_processors.py

* processors.pt: the reference's attention processors (videoswap/utils/edlora_util.py:13-82 EDLoRA_AttnProcessor,
  videoswap/utils/p2p_utils/attention_register.py:15-173 EDLoRA_AttnControlProcessor / AttnControlProcessor), imported
  verbatim through oracle/ref_import.py, run on the restated diffusers `Attention` (fp32, CPU) with seeded weights /
  inputs and the toy controller of tests/golden/toy.py.  (xformers is absent here, so the reference takes its
  materialised-probabilities branch for every token count.)
* edlora.pt: the reference's convert_edlora_to_diffusers.py (merge_lora_into_weight, load_new_concept) and
  edlora_util.py (bind_concept_prompt, encode_edlora_prompt) on the small case of toy.lora_case().
Inputs are regenerated from seeds by toy.py; only the reference's OUTPUTS are stored.
"""
import os
import sys

import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import toy  # noqa: E402
from oracle import ref_import  # noqa: E402


def reference_modules():
    p2p = ref_import.load_reference_p2p()
    reg = ref_import._load('videoswap.utils.p2p_utils.attention_register',
                           'videoswap/utils/p2p_utils/attention_register.py')
    edl = sys.modules['videoswap.utils.edlora_util']
    conv = ref_import._load('videoswap.utils.convert_edlora_to_diffusers',
                            'videoswap/utils/convert_edlora_to_diffusers.py')
    return p2p, reg, edl, conv


def oracle_attention(sd, cross):
    from oracle.diffusers_restated import Attention
    a = Attention(query_dim=320, cross_attention_dim=768 if cross else None, heads=8, dim_head=40).eval()
    a.load_state_dict(sd, strict=True)
    return a


def processor_cases(reg, edl):
    inp = toy.attention_inputs()
    self_sd, cross_sd = toy.attention_weights()
    a_self, a_cross = oracle_attention(self_sd, False), oracle_attention(cross_sd, True)
    out = {}
    with torch.no_grad():
        out['edlora_cross_layers_idx3'] = edl.EDLoRA_AttnProcessor(3)(a_cross, inp['hidden'], inp['text_layers'])
        out['edlora_cross_single'] = edl.EDLoRA_AttnProcessor(3)(a_cross, inp['hidden'], inp['text'])
        out['edlora_self'] = edl.EDLoRA_AttnProcessor(0)(a_self, inp['hidden'], None)
        c = toy.ToyController()
        out['control_self_down'] = reg.AttnControlProcessor('down', c)(a_self, inp['hidden'], None)
        out['control_cross_mid'] = reg.AttnControlProcessor('mid', c)(a_cross, inp['hidden'], inp['text'])
        out['edlora_control_cross_up_idx5'] = reg.EDLoRA_AttnControlProcessor(5, 'up', c)(
            a_cross, inp['hidden'], inp['text_layers'])
        out['edlora_control_self_up'] = reg.EDLoRA_AttnControlProcessor(5, 'up', c)(a_self, inp['hidden'], None)
        out['controller_calls'] = list(c.calls)
    return out


class _Tok:
    """tokenizer protocol used by load_new_concept / encode_edlora_prompt (whitespace words, stable ids)"""

    def __init__(self):
        from videoswap_amd.synthetic import WhitespaceTokenizer
        self.t = WhitespaceTokenizer()
        self.model_max_length = 77
        self.added = []

    def add_tokens(self, names):
        self.added.extend(names)
        return self.t.add_tokens(names)

    def convert_tokens_to_ids(self, name):
        return 100 + self.added.index(name)          # ids into the toy embedding table

    def __len__(self):
        return 100 + len(self.added)

    def __call__(self, *a, **k):
        return self.t(*a, **k)


class _TextEncoder(torch.nn.Module):
    def __init__(self, dim=24):
        super().__init__()
        self.emb = torch.nn.Embedding(100, dim)
        with torch.no_grad():
            self.emb.weight.copy_(torch.randn(100, dim, generator=torch.Generator().manual_seed(5)))
        self.dtype = torch.float32

    def resize_token_embeddings(self, n):
        old = self.emb
        self.emb = torch.nn.Embedding(n, old.embedding_dim)
        with torch.no_grad():
            self.emb.weight.zero_()
            self.emb.weight[:old.num_embeddings] = old.weight

    def get_input_embeddings(self):
        return self.emb

    def forward(self, ids):
        g = torch.Generator().manual_seed(6)
        table = torch.randn(50000, self.emb.embedding_dim, generator=g)
        return (table[ids % 50000] + 0.01 * torch.arange(ids.shape[1])[None, :, None],)


class _Pipe:
    def __init__(self, unet_sd, text_sd):
        self.tokenizer = _Tok()
        self.text_encoder = _TextEncoder()
        self.unet = type('U', (), {})()
        self.unet.dtype = torch.float32
        self._unet_sd, self._text_sd = unet_sd, text_sd


def edlora_cases(edl, conv):
    unet, text, ckpt = toy.lora_case()
    out = {'merged_unet': conv.merge_lora_into_weight(unet, ckpt['params']['unet'], 'unet', 0.7),
           'merged_text': conv.merge_lora_into_weight(text, ckpt['params']['text_encoder'], 'text_encoder', 1.0)}
    pipe = _Pipe(unet, text)
    _, cfg = conv.load_new_concept(pipe, ckpt['params']['new_concept_embedding'], enable_edlora=True)
    out['new_concept_cfg'] = cfg
    out['token_table_tail'] = pipe.text_encoder.get_input_embeddings().weight.data[100:].clone()
    prompt = 'a <catA1> <catA2> sitting on a wooden floor'
    out['bound_prompts'] = edl.bind_concept_prompt(prompt, cfg)
    out['prompt_embeds_cfg'] = edl.encode_edlora_prompt(pipe, prompt, cfg, 'cpu', 1, True, 'low quality')
    out['prompt_embeds_nocfg'] = edl.encode_edlora_prompt(pipe, [prompt], cfg, 'cpu', 1, False)
    return out


def main():
    p2p, reg, edl, conv = reference_modules()
    torch.save(dict(cases=processor_cases(reg, edl),
                    generator='tests/golden/make_golden_processors.py on attention_register.py / edlora_util.py verbatim'),
               os.path.join(HERE, 'processors.pt'))
    torch.save(dict(cases=edlora_cases(edl, conv),
                    generator='tests/golden/make_golden_processors.py on convert_edlora_to_diffusers.py / edlora_util.py verbatim'),
               os.path.join(HERE, 'edlora.pt'))
    for n in ('processors.pt', 'edlora.pt'):
        print('wrote', n, os.path.getsize(os.path.join(HERE, n)), 'bytes')


if __name__ == '__main__':
    main()
