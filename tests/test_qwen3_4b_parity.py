"""The same parity module at Qwen3-4B WIDTHS (the LLM half of BASELINE configs 4-5 that needs no Evo2 oracle;
`README.md:84` NT-500M + Qwen3-4B): tests/test_fullsize_parity.py executed a second time under the preset `qwen3_4b` —
hidden 2560, 32 query / 8 kv heads (G = 4), intermediate 9728, 3 layers, against the oracle (HF Qwen3 takes any dims).  New in the
kernels on this path: decode projections with K = 2560 (80 k-steps: the 8-wave x 10-chunk instantiation), 16-column tiles where
N / 8 statistics partials would exceed 256, decode attention with more than 16 query rows per (prompt, kv-head) as virtual prompts."""
import importlib.util
import os

_here = os.path.dirname(os.path.abspath(__file__))
_prev = os.environ.get("BRA_FULLSIZE_PRESET")
os.environ["BRA_FULLSIZE_PRESET"] = "qwen3_4b"
try:
    _spec = importlib.util.spec_from_file_location("fullsize_parity_qwen3_4b", os.path.join(_here, "test_fullsize_parity.py"))
    _m = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(_m)
finally:
    if _prev is None:
        os.environ.pop("BRA_FULLSIZE_PRESET", None)
    else:
        os.environ["BRA_FULLSIZE_PRESET"] = _prev

any_device = _m.any_device
runs = _m.runs
test_forward_backward_4b_widths = _m.test_forward_backward_fullsize
test_per_token_logps_4b_widths = _m.test_per_token_logps_fullsize
test_shared_policy_pass_4b_widths = _m.test_shared_policy_pass_fullsize
test_greedy_decode_fused_shared_prefix_4b_widths = _m.test_greedy_decode_fused_shared_prefix_fullsize
test_greedy_decode_many_rows_4b_widths = _m.test_greedy_decode_many_rows_fullsize
test_zz_dump_ratios_4b_widths = _m.test_zz_dump_ratios
