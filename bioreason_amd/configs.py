"""Architecture configs of the model pairs the reference trains (no weights offline: shapes only).

"Qwen3-1B" in the reference's README is `Qwen/Qwen3-1.7B` in every script (train_dna_qwen.py:1016, sh_reason.sh:46);
"NT-500M" is `InstaDeepAI/nucleotide-transformer-v2-500m-multi-species` (train_dna_qwen.py:1017).  SURVEY §8.
"""
from transformers import EsmConfig, Qwen3Config


def qwen3_config(vocab_size=151936, hidden_size=2048, intermediate_size=6144, num_hidden_layers=28, num_attention_heads=16,
                 num_key_value_heads=8, head_dim=128, rope_theta=1e6, max_position_embeddings=40960, rms_norm_eps=1e-6) -> Qwen3Config:
    return Qwen3Config(vocab_size=vocab_size, hidden_size=hidden_size, intermediate_size=intermediate_size,
                       num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
                       num_key_value_heads=num_key_value_heads, head_dim=head_dim, max_position_embeddings=max_position_embeddings,
                       rms_norm_eps=rms_norm_eps, tie_word_embeddings=True, attention_bias=False,
                       rope_parameters={"rope_type": "default", "rope_theta": rope_theta})


def nt_v2_config(vocab_size=4107, hidden_size=1024, intermediate_size=4096, num_hidden_layers=29, num_attention_heads=16,
                 max_position_embeddings=2050) -> EsmConfig:
    return EsmConfig(vocab_size=vocab_size, hidden_size=hidden_size, intermediate_size=intermediate_size,
                     num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
                     max_position_embeddings=max_position_embeddings, position_embedding_type="rotary", pad_token_id=1,
                     mask_token_id=2, token_dropout=False, emb_layer_norm_before=False, layer_norm_eps=1e-12,
                     hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)


QWEN3_1P7B = dict()                                                  # defaults above
QWEN3_4B = dict(hidden_size=2560, intermediate_size=9728, num_hidden_layers=36, num_attention_heads=32, num_key_value_heads=8)
NT_V2_500M = dict()
