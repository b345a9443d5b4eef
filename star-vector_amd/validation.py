"""`generation_engine: hip` for the reference's validation driver.

The reference picks a validator class by name (validation/validate.py:6-29: 'hf' | 'vllm' | 'vllm-api' -> classes in
`validator_registry`) and every validator contributes one thing to the loop in svg_validator_base.py:304-321,373-377:
`generate_svg(batch, generate_config) -> list[str]`.  This module is that one method over the HIP engine; datasets,
rasterisation, post-processing (`clean_svg`, svgpathtools) and metrics stay the reference's.

The subclass itself lives on the reference's side (it derives from the reference's `SVGValidator` and registers in its
`validator_registry`): INTEGRATION.md shows the dozen lines a maintainer adds.  `generate_svg` below is everything
it calls.
"""
from typing import Dict, List

import torch


def generate_svg(model, task: str, batch: Dict, generate_config: Dict, device=None) -> List[str]:
    """starvector_hf_validator.py:75-88, statement by statement.

    Kept quirk: with temperature == 0 the reference sets temperature = 1.0 and `do_sample = False`, but the generation
    whitelist reads `use_nucleus_sampling`, not `do_sample` (starvector_base.py:228-241), so the call still samples
    unless the config also says `use_nucleus_sampling: false`.  The mirror has the same whitelist, hence the same result.
    """
    generate_config = dict(generate_config)                 # the reference mutates the caller's config; a copy is enough
    if generate_config.get("temperature") == 0:
        generate_config["temperature"] = 1.0
        generate_config["do_sample"] = False
    if device is None:
        device = torch.device("cuda", model.engine.device) if hasattr(model, "engine") else batch["image"].device
    batch = dict(batch)
    batch["image"] = batch["image"].to(device).to(torch.bfloat16)    # :82 (`.to('cuda').to(self.torch_dtype)`)
    if task == "im2svg":
        return model.model.generate_im2svg(batch=batch, **generate_config)
    if task == "text2svg":
        out = model.model.generate_text2svg(batch=batch, **generate_config)
        # the reference returns the token tensor of HF generate here (starvector_base.py:330) and the loop then treats
        # each row as text; decode so that post_process_svg receives strings
        if torch.is_tensor(out):
            out = model.model.svg_transformer.tokenizer.batch_decode(out, skip_special_tokens=True)
        return out
    return []                                                # :78 `outputs = []` for any other task
