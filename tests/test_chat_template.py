"""The DNA chat template (`text_tokenizer.chat_template = CHAT_TEMPLATE`, dna_llm.py:69) against the reference's own
template on the conversation shapes its pipelines build (kegg.py / reason.py), and `prepare_prompt` end to end."""
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bioreason_amd.chat_template import CHAT_TEMPLATE      # noqa: E402

REF = "/root/reference/bioreason/models/dl/chat_template_dl.py"

CONVS = [
    dict(messages=[{"role": "system", "content": "You are a helpful assistant."},
                   {"role": "user", "content": [{"type": "dna", "text": None}, {"type": "dna", "text": None},
                                                {"type": "text", "text": "Which pathway is affected?"}]}],
         add_generation_prompt=True),
    dict(messages=[{"role": "user", "content": [{"type": "dna"}, {"type": "text", "text": "Q?"}]},
                   {"role": "assistant", "content": [{"type": "text", "text": "thinking...\n\nAnswer: ALS"}],
                    "reasoning_content": "step 1\nstep 2\n"}],
         add_generation_prompt=False),
    dict(messages=[{"role": "user", "content": "plain string question"},
                   {"role": "assistant", "content": [{"type": "text", "text": "first answer"}]},
                   {"role": "user", "content": [{"type": "text", "text": "follow up"}, {"type": "dna"}]}],
         add_generation_prompt=True, enable_thinking=False),
    dict(messages=[{"role": "system", "content": "sys"}, {"role": "user", "content": [{"type": "dna"}, {"type": "dna"}]},
                   {"role": "assistant", "content": [{"type": "text", "text": "a"}]},
                   {"role": "system", "content": "later system"}, {"role": "user", "content": "u2"},
                   {"role": "assistant", "content": [{"type": "text", "text": "\nb"}]}],
         add_generation_prompt=False, add_dna_id=True),
]

# renderings of CONVS by the reference's template (recorded in the build container; re-checked live when it is present)
KNOWN = [
    "<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n<|im_start|>user\n<|dna_start|><|dna_pad|><|dna_end|>"
    "<|dna_start|><|dna_pad|><|dna_end|>Which pathway is affected?<|im_end|>\n<|im_start|>assistant\n",
    "<|im_start|>user\n<|dna_start|><|dna_pad|><|dna_end|>Q?<|im_end|>\n<|im_start|>assistant\n<think>\nstep 1\nstep 2\n</think>\n\n"
    "thinking...\n\nAnswer: ALS<|im_end|>\n",
    "<|im_start|>user\nplain string question<|im_end|>\n<|im_start|>assistant\nfirst answer<|im_end|>\n<|im_start|>user\nfollow up"
    "<|dna_start|><|dna_pad|><|dna_end|><|im_end|>\n<|im_start|>assistant\n<think>\n\n</think>\n\n",
    "<|im_start|>system\nsys<|im_end|>\n<|im_start|>user\nDNA Sequence1:<|dna_start|><|dna_pad|><|dna_end|>DNA Sequence2:"
    "<|dna_start|><|dna_pad|><|dna_end|><|im_end|>\n<|im_start|>assistant\na<|im_end|>\n<|im_start|>system\nlater system<|im_end|>\n"
    "<|im_start|>user\nu2<|im_end|>\n<|im_start|>assistant\n<think>\n\n</think>\n\nb<|im_end|>\n",
]


def render(template, conv):
    from jinja2.sandbox import ImmutableSandboxedEnvironment

    def raise_exception(msg):
        raise ValueError(msg)
    env = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True)     # as transformers compiles chat templates
    env.globals["raise_exception"] = raise_exception
    return env.from_string(template).render(**conv)


@pytest.mark.parametrize("i", range(len(CONVS)))
def test_known_renderings(i):
    assert render(CHAT_TEMPLATE, CONVS[i]) == KNOWN[i]


@pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not present")
@pytest.mark.parametrize("i", range(len(CONVS)))
def test_equals_reference_template(i):
    spec = importlib.util.spec_from_file_location("ref_chat_template", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert render(mod.CHAT_TEMPLATE, CONVS[i]) == KNOWN[i]
    assert render(CHAT_TEMPLATE, CONVS[i]) == render(mod.CHAT_TEMPLATE, CONVS[i])


def test_tools_are_rejected():
    with pytest.raises(ValueError):
        render(CHAT_TEMPLATE, dict(messages=[{"role": "user", "content": "x"}], tools=[{"name": "f"}]))


def test_prepare_prompt_renders_dna_placeholders():
    """a list-content DNA prompt through NucleotideDNAModule.prepare_prompt with a real (toy) tokenizer carrying the template"""
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import GPT2TokenizerFast
    from bioreason_amd.dna_modules import NucleotideDNAModule
    from bioreason_amd.processing import DLProcessor
    tk = Tokenizer(models.WordLevel({"<|endoftext|>": 0, "[UNK]": 1}, unk_token="[UNK]"))
    tk.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok = GPT2TokenizerFast(tokenizer_object=tk, eos_token="<|endoftext|>", unk_token="[UNK]")
    tok.chat_template = CHAT_TEMPLATE
    proc = DLProcessor(tokenizer=tok, dna_tokenizer=None)
    out = NucleotideDNAModule().prepare_prompt(proc, [{"prompt": CONVS[0]["messages"]}])
    assert out == [KNOWN[0]]
