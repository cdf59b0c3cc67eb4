"""Chat template of the DNA-LLM tokenizer: what `DNALLMModel.__init__` assigns to `text_tokenizer.chat_template`
(bioreason/models/dna_llm.py:69, bioreason/models/dl/chat_template_dl.py).

Behaviour (Qwen3's ChatML rendering + DNA placeholders), restated as a fresh Jinja program for the message shapes the
reference produces (kegg.py / reason.py conversations: optional leading system message, user messages whose content is a
string or a list of {"type": "dna"} / {"type": "text", "text": ...} items, assistant messages whose content is a list with
one text item, optional `reasoning_content`):
  * leading system message            -> <|im_start|>system\\n{content}<|im_end|>\\n
  * user / later system message       -> <|im_start|>{role}\\n ... <|im_end|>\\n, every DNA item rendered as
                                         <|dna_start|><|dna_pad|><|dna_end|> (prefixed "DNA Sequence{n}:" when
                                         `add_dna_id`), text items verbatim
  * assistant message                 -> after the last user query and (last message, or with reasoning):
                                         <|im_start|>assistant\\n<think>\\n{reasoning}\\n</think>\\n\\n{content}<|im_end|>\\n,
                                         otherwise <|im_start|>assistant\\n{content}<|im_end|>\\n
  * add_generation_prompt             -> <|im_start|>assistant\\n (+ an empty think block when enable_thinking is false)
Tool calling (never used by the reference's DNA pipelines) is rejected with an explicit error instead of being rendered.
`tests/test_chat_template.py` renders the same conversations through this template and the reference's and compares."""

_LINES = [
    "{%- if tools %}{{- raise_exception('the DNA-LLM chat template does not render tool definitions') }}{%- endif %}",
    "{%- set seen = namespace(dna=0, last_query=messages|length - 1, searching=true) %}",
    # index of the last genuine user query (tool responses wrapped in user turns do not count)
    "{%- for m in messages[::-1] %}",
    "{%- set i = (messages|length - 1) - loop.index0 %}",
    "{%- if seen.searching and m.role == 'user' and not (m.content is string and m.content.startswith('<tool_response>') "
    "and m.content.endswith('</tool_response>')) %}",
    "{%- set seen.searching = false %}{%- set seen.last_query = i %}",
    "{%- endif %}",
    "{%- endfor %}",
    "{%- for m in messages %}",
    "{%- if m.role == 'system' and loop.first %}",
    "{{- '<|im_start|>system\\n' + m.content + '<|im_end|>\\n' }}",
    "{%- elif m.role == 'user' or m.role == 'system' %}",
    "{{- '<|im_start|>' + m.role + '\\n' }}",
    "{%- if m.content is string %}",
    "{{- m.content }}",
    "{%- else %}",
    "{%- for item in m.content %}",
    "{%- if item.type == 'dna' or 'dna' in item %}",
    "{%- set seen.dna = seen.dna + 1 %}",
    "{%- if add_dna_id %}{{- 'DNA Sequence' ~ seen.dna ~ ':' }}{%- endif %}",
    "{{- '<|dna_start|><|dna_pad|><|dna_end|>' }}",
    "{%- elif 'text' in item %}",
    "{{- item.text }}",
    "{%- endif %}",
    "{%- endfor %}",
    "{%- endif %}",
    "{{- '<|im_end|>\\n' }}",
    "{%- elif m.role == 'assistant' %}",
    "{%- set body = m.content[0].text %}",
    "{%- set thought = '' %}",
    "{%- if m.reasoning_content is defined and m.reasoning_content is not none %}",
    "{%- set thought = m.reasoning_content %}",
    "{%- elif '</think>' in m.content %}",
    "{%- set thought = body.split('</think>')[0].rstrip('\\n').split('<think>')[-1].lstrip('\\n') %}",
    "{%- set body = body.split('</think>')[-1].lstrip('\\n') %}",
    "{%- endif %}",
    "{%- if loop.index0 > seen.last_query and (loop.last or thought) %}",
    "{{- '<|im_start|>assistant\\n<think>\\n' + thought.strip('\\n') + '\\n</think>\\n\\n' + body.lstrip('\\n') }}",
    "{%- else %}",
    "{{- '<|im_start|>assistant\\n' + body }}",
    "{%- endif %}",
    "{%- if m.tool_calls %}{{- raise_exception('the DNA-LLM chat template does not render tool calls') }}{%- endif %}",
    "{{- '<|im_end|>\\n' }}",
    "{%- elif m.role == 'tool' %}",
    "{{- raise_exception('the DNA-LLM chat template does not render tool messages') }}",
    "{%- endif %}",
    "{%- endfor %}",
    "{%- if add_generation_prompt %}",
    "{{- '<|im_start|>assistant\\n' }}",
    "{%- if enable_thinking is defined and enable_thinking is false %}{{- '<think>\\n\\n</think>\\n\\n' }}{%- endif %}",
    "{%- endif %}",
]
CHAT_TEMPLATE = "\n".join(_LINES)
