"""CPU oracle for the StarVector im2svg hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product path
(``star-vector_amd``) never imports this package and fails loudly when the HIP
library is missing.

Parity status: the reference (joanrod/star-vector) ships NO tests and NO golden
vectors for this path, so the oracle is pinned against
outputs of the reference itself run in the build container:
``oracle/make_golden.py`` imports ``/root/reference``'s own
``clip_model.VisionTransformer`` / ``adapter.Adapter`` and the un-vendored
decoder the reference loads (``transformers.GPTBigCodeForCausalLM`` driven by
``GenerationMixin.generate``), checks this restatement against them and writes
the fixtures under ``tests/golden/``.  The image pre-processing recipe (``oracle/image_preprocess.py``) restates
Pillow's integer arithmetic and is pinned against Pillow itself (``image_preprocess.pin()``).
"""
from .starvector_oracle import (  # noqa: F401
    OracleConfig,
    make_weights,
    iter_weights,
    vit_forward,
    image_encoder_forward,
    adapter_forward,
    decoder_prefill,
    decoder_decode_step,
    decoder_forward_logits,
    prepare_generation_inputs,
    greedy_generate,
    beam_search_generate,
    BeamSearchState,
    top_p_filtered_probs,
    warp_scores,
    generate_im2svg_tokens,
    synthetic_images,
    siglip_forward,
    embed_key,
    fake_quantize_fp8,
)
