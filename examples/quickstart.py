"""scripts/quickstart.py and scripts/quickstart-hf.py of the reference, on the HIP engine: the same calls, the import line and
the checkpoint location are the only differences (there is no hub access here: pass a local checkpoint directory in the
reference's format, config.json + *.safetensors [+ tokenizer files]).

    python examples/quickstart.py /path/to/starvector-1b-im2svg assets/examples/sample-18.png
"""
import sys

import torch
from PIL import Image

from starvector_amd import StarVectorForCausalLM          # reference: from starvector.model.starvector_arch import ...

model_path, image_path = sys.argv[1], sys.argv[2]

starvector = StarVectorForCausalLM.from_pretrained(model_path, torch_dtype="auto")
starvector.cuda()                                           # no-ops: the weights already live repacked inside the engine
starvector.eval()

image_pil = Image.open(image_path).convert("RGB")

# scripts/quickstart.py: the encoder's own pre-processing, the reference's sampling arguments (num_beams defaults to 2 ->
# beam-sample with length_penalty -1 and repetition_penalty 3.1)
image = starvector.process_images([image_pil])[0].to(torch.float16).cuda()
raw_svg = starvector.generate_im2svg({"image": image}, max_length=4000, temperature=1.5, length_penalty=-1,
                                     repetition_penalty=3.1)[0]
print(raw_svg)

# scripts/quickstart-hf.py: the HF-style processor and tokenizer hanging off the model
processor = starvector.model.processor
tokenizer = starvector.model.svg_transformer.tokenizer
image = processor(image_pil, return_tensors="pt")["pixel_values"].cuda()
if not image.shape[0] == 1:
    image = image.squeeze(0)                                # as in the reference; a [3, S, S] image is accepted as a batch of one
raw_svg = starvector.generate_im2svg({"image": image}, max_length=100 + starvector.model.query_length)[0]
print(raw_svg)

# the reference continues with `svg, raster_image = process_and_rasterize_svg(raw_svg)` (starvector/data/util.py:123-136):
# host-side post-processing (bs4 / svgpathtools / cairosvg), unchanged and outside the engine
