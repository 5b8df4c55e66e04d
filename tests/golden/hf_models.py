"""Seeded construction of the HuggingFace models the drop-in fixtures are defined on (shared by the
fixture generator and the tests; weights come from the seed, a checksum in the fixture pins them)."""
import torch


def wsum(model):
    return float(sum(p.detach().double().abs().sum() for p in model.parameters()))


def build_bert(seed=0, attn="eager"):
    from transformers import BertConfig, BertForSequenceClassification
    torch.manual_seed(seed)
    cfg = BertConfig(num_labels=2, attn_implementation=attn)            # BERT-base defaults: 12 L, H 768, 12 heads, I 3072
    return BertForSequenceClassification(cfg).eval()


def build_gemma3(seed=3, attn="eager"):
    from transformers import Gemma3TextConfig, Gemma3ForCausalLM
    torch.manual_seed(seed)
    cfg = Gemma3TextConfig(vocab_size=512, hidden_size=128, intermediate_size=256, num_hidden_layers=4, num_attention_heads=4,
                           num_key_value_heads=2, head_dim=32, sliding_window=32, max_position_embeddings=512,
                           layer_types=["sliding_attention", "sliding_attention", "sliding_attention", "full_attention"],
                           query_pre_attn_scalar=32, attn_implementation=attn, tie_word_embeddings=False)
    return Gemma3ForCausalLM(cfg).eval()


def build_llama(seed=5, attn="eager"):
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(seed)
    cfg = LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                      head_dim=32, vocab_size=256, max_position_embeddings=512, attn_implementation=attn, tie_word_embeddings=False)
    return LlamaForCausalLM(cfg).eval()


def build_qwen2(seed=6, attn="eager"):
    from transformers import Qwen2Config, Qwen2ForCausalLM
    torch.manual_seed(seed)
    cfg = Qwen2Config(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                      vocab_size=256, max_position_embeddings=512, attn_implementation=attn, tie_word_embeddings=False,
                      use_sliding_window=False)
    return Qwen2ForCausalLM(cfg).eval()


def build_qwen3(seed=7, attn="eager"):
    from transformers import Qwen3Config, Qwen3ForCausalLM
    torch.manual_seed(seed)
    cfg = Qwen3Config(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                      head_dim=32, vocab_size=256, max_position_embeddings=512, attn_implementation=attn, tie_word_embeddings=False,
                      use_sliding_window=False)
    return Qwen3ForCausalLM(cfg).eval()


def build_gpt2(seed=8, attn="eager"):
    from transformers import GPT2Config, GPT2LMHeadModel
    torch.manual_seed(seed)
    cfg = GPT2Config(n_embd=128, n_layer=3, n_head=4, vocab_size=256, n_positions=512, attn_implementation=attn,
                     resid_pdrop=0.0, embd_pdrop=0.0, attn_pdrop=0.0)
    return GPT2LMHeadModel(cfg).eval()


BUILDERS = dict(llama_cp=build_llama, qwen2=build_qwen2, qwen3=build_qwen3, gpt2=build_gpt2)
