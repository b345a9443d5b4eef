"""Write a checkpoint DIRECTORY in the format the reference's `StarVectorForCausalLM.from_pretrained` reads (HF `save_pretrained` of
starvector_arch.py:96-145): `config.json` with StarVectorConfig's fields, sharded `model-0000i-of-0000n.safetensors` +
`model.safetensors.index.json`, tensors under the reference's own state_dict names (`model.image_encoder.…`, `model.image_projection.…`,
`model.svg_transformer.transformer.…`; the tied `lm_head.weight` is not saved: HF drops shared tensors, train/util.py:71 pops it)."""
import json
import os

import torch
from safetensors.torch import save_file

from oracle import starvector_oracle as O


def write_reference_checkpoint(path, cfg: O.OracleConfig, w, n_shards=2, torch_dtype="bfloat16", max_batch=4, max_length=None):
    os.makedirs(path, exist_ok=True)
    v2 = cfg.arch == "v2"
    conf = {
        "architectures": ["StarVectorForCausalLM"], "model_type": "starvector",
        "starcoder_model_name": "bigcode/starcoder2-7b" if v2 else "bigcode/starcoderbase-1b",
        "image_encoder_type": "siglip_384" if v2 else "clip", "adapter_norm": cfg.adapter_norm, "image_size": cfg.image_size,
        "max_length": max_length or cfg.n_positions, "max_length_train": cfg.n_positions, "use_flash_attn": True, "use_cache": True,
        "num_attention_heads": cfg.n_head, "num_hidden_layers": cfg.n_layer, "vocab_size": cfg.vocab - (5 if v2 else 4),
        "hidden_size": cfg.hidden, "num_kv_heads": cfg.n_kv_head, "torch_dtype": torch_dtype, "transformers_version": "4.49.0",
        # not in the reference's config (it instantiates fixed HF / CLIP sub-models): the reduced vision tower of the test model and
        # the engine's batch capacity
        "vit_width": cfg.vit_width, "vit_layers": cfg.vit_layers, "vit_heads": cfg.vit_heads, "patch_size": cfg.patch_size,
        "max_batch": max_batch,
    }
    if v2:
        conf.update(siglip_image_size=cfg.image_size, siglip_patch_size=cfg.patch_size, siglip_layers=cfg.vit_layers,
                    siglip_mlp=cfg.vit_mlp, rope_theta=cfg.rope_theta, sliding_window=cfg.sliding_window or 4096)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(conf, f, indent=2)
    dt = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float32}[torch_dtype]
    names = sorted(k for k in w if k != O.K_LMH)
    per = (len(names) + n_shards - 1) // n_shards
    index = {"metadata": {"total_size": 0}, "weight_map": {}}
    for s in range(n_shards):
        part = {k: w[k].to(dt).contiguous() for k in names[s * per:(s + 1) * per]}
        fn = f"model-{s + 1:05d}-of-{n_shards:05d}.safetensors"
        save_file(part, os.path.join(path, fn), metadata={"format": "pt"})
        for k, v in part.items():
            index["weight_map"][k] = fn
            index["metadata"]["total_size"] += v.numel() * v.element_size()
    with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
        json.dump(index, f, indent=2)
    return conf
