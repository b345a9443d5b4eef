"""Model worker over the HIP engine: the caller side of the hot path.

Mirrors the reference's HF worker, starvector/serve/model_worker.py:36-233: same request fields, same wire format
(JSON objects ``{"text", "error_code"}`` each terminated by a NUL byte, the text being the WHOLE svg so far), same two
routes (``/worker_generate_stream``, ``/worker_get_status``), same controller registration + heart beat, so the
reference's controller and Gradio front end (serve/controller.py, serve/gradio_web_server.py) talk to it unchanged.
Those two stay the reference's: nothing of them is rebuilt here.

Differences, all on purpose:
  * The reference hands a ``TextIteratorStreamer`` to ``generate_im2svg``, whose keyword whitelist drops it
    (starvector_base.py:223-241), so its stream only ever ends in the streamer's 15 s timeout.  The mirror keeps the
    keyword (model.py `_get_generation_kwargs`) and the engine calls back every `sync_every` decode steps, so text
    does arrive while the SVG is being written.
  * An exception inside the generation thread is carried to the stream and reported at once as ``error_code: 1``
    (the reference loses it with the thread and reports the timeout instead).
  * The reference skips every chunk equal to " " (:176-177); TextStreamer emits exactly that for the leading space of
    the first word after a flush, so real spaces would be lost ("<svgwidth=").  Only empty chunks are skipped here: the
    last object's text equals what `generate_im2svg` returns.
  * One generation is in flight per engine handle (the C ABI serialises calls on a handle); further requests wait
    on the same asyncio semaphore the reference uses.

FastAPI / uvicorn / requests are imported lazily: `ModelWorker.generate_stream` is plain Python over the mirror.
"""
import argparse
import base64
import json
import threading
import time
import uuid
from io import BytesIO
from queue import Queue
from typing import Iterator, Optional

import torch

WORKER_HEART_BEAT_INTERVAL = 15          # serve/constants.py:2
CLIP_QUERY_LENGTH = 257                  # serve/constants.py:15
server_error_msg = "**NETWORK ERROR DUE TO HIGH TRAFFIC. PLEASE REGENERATE OR REFRESH THIS PAGE.**"   # serve/util.py:11


def load_image_from_base64(image):                                   # serve/util.py:123-124
    from PIL import Image
    return Image.open(BytesIO(base64.b64decode(image)))


def process_images(image, image_processor):                          # serve/util.py:126-129
    out = image_processor(image)
    if not torch.is_tensor(out):                                     # HF-style processor (the v2 tower's): {"pixel_values": ...}
        out = out["pixel_values"]
    return out if out.dim() == 4 else out.unsqueeze(0)


def load_pretrained_model(model_path, device="cuda", **kwargs):      # model/builder.py:6-11
    from .model import StarVectorForCausalLM
    model = StarVectorForCausalLM.from_pretrained(model_path, **kwargs)
    tokenizer = model.model.svg_transformer.tokenizer
    # the reference builds a host ImageTrainProcessor() for every checkpoint; the engine-side processor of the tower gives
    # the same float32 tensor bit for bit (sv_preprocess_image) without leaving the GPU, and is the right recipe for v2
    image_processor = model.model.image_encoder.processor
    context_len = model.model.query_length + model.model.max_length
    return tokenizer, model, image_processor, context_len


class TextQueueStreamer:
    """The subset of HF's TextIteratorStreamer (generation/streamers.py) the worker relies on, without importing
    transformers: `put(ids)` / `end()` on the producer side, iteration over finalized text on the consumer side.
    Text is released the way TextStreamer does it: everything up to the last space, the rest when a newline or the end
    arrives, so a multi-byte character or a word is never cut in the middle.  Batch size 1 only, like HF's."""

    _END = object()

    def __init__(self, tokenizer, skip_prompt: bool = False, timeout: Optional[float] = None, **decode_kwargs):
        self.tokenizer, self.skip_prompt, self.timeout, self.decode_kwargs = tokenizer, skip_prompt, timeout, decode_kwargs
        self.token_cache, self.print_len, self.next_tokens_are_prompt = [], 0, True
        self.text_queue: Queue = Queue()

    def put(self, value):
        if value.dim() > 1 and value.shape[0] > 1:
            raise ValueError("TextStreamer only supports batch size 1")
        if value.dim() > 1:
            value = value[0]
        if self.skip_prompt and self.next_tokens_are_prompt:
            self.next_tokens_are_prompt = False
            return
        self.next_tokens_are_prompt = False
        self.token_cache.extend(value.tolist())
        text = self.tokenizer.decode(self.token_cache, **self.decode_kwargs)
        if text.endswith("\n"):
            out = text[self.print_len:]
            self.token_cache, self.print_len = [], 0
        else:
            out = text[self.print_len:text.rfind(" ") + 1]
            self.print_len += len(out)
        self.text_queue.put(out, timeout=self.timeout)

    def end(self):
        out = ""
        if self.token_cache:
            out = self.tokenizer.decode(self.token_cache, **self.decode_kwargs)[self.print_len:]
            self.token_cache, self.print_len = [], 0
        self.next_tokens_are_prompt = True
        self.text_queue.put(out, timeout=self.timeout)
        self.text_queue.put(self._END, timeout=self.timeout)

    def __iter__(self):
        return self

    def __next__(self):
        v = self.text_queue.get(timeout=self.timeout)
        if v is self._END:
            raise StopIteration
        return v


class ModelWorker:
    """serve/model_worker.py:36-207.  `model` is a `StarVectorForCausalLM` mirror (or anything with the same
    ``.model.generate_im2svg / generate_text2svg`` and ``.config``)."""

    def __init__(self, controller_addr: str, worker_addr: str, worker_id: str, no_register: bool, model_path: str = "",
                 model_name: Optional[str] = None, device="cuda", limit_model_concurrency: int = 5, *,
                 model=None, tokenizer=None, image_processor=None, context_len: Optional[int] = None,
                 continuous_batching: bool = True):
        self.controller_addr, self.worker_addr, self.worker_id = controller_addr, worker_addr, worker_id
        model_path = model_path[:-1] if model_path.endswith("/") else model_path
        if model_name is None:                                         # :46-52
            parts = model_path.split("/")
            model_name = parts[-2] + "_" + parts[-1] if parts[-1].startswith("checkpoint-") and len(parts) > 1 else parts[-1]
        self.model_name = model_name
        # :54-57 leaves `task` unset for any other name (and fails later); im2svg is the path this engine serves
        self.task = "Text2SVG" if "text2svg" in model_name.lower() else "Image2SVG"
        self.device = device
        if model is None:
            tokenizer, model, image_processor, context_len = load_pretrained_model(model_path, device=device)
        self.tokenizer, self.model, self.image_processor, self.context_len = tokenizer, model, image_processor, context_len
        self.is_multimodal = "starvector" in model_name.lower()       # :66
        # The reference admits `limit_model_concurrency` requests at once and lets their generate calls take turns on the GPU
        # (:161-172, 216-229).  Here the admitted requests share ONE decode loop (batching.ContinuousBatcher over the
        # engine's sv_cb_* entry points): each request is a row of the batch with its own sampling parameters and stop.
        self.batcher = None
        lm = getattr(getattr(getattr(self.model, "model", None), "svg_transformer", None), "transformer", None)
        eng = getattr(lm, "_engine", None)
        if continuous_batching and lm is not None and eng is not None and hasattr(eng, "cb_admit"):
            from .batching import ContinuousBatcher
            self.batcher = ContinuousBatcher(eng)
            lm.batcher = self.batcher
        self.limit_model_concurrency = limit_model_concurrency
        self.model_semaphore = None                                    # created on the event loop by the first request
        self.global_counter = 0
        self.heart_beat_thread = None
        if not no_register:
            self.register_to_controller()
            self.heart_beat_thread = threading.Thread(target=self._heart_beat_loop, daemon=True)
            self.heart_beat_thread.start()

    # ---- controller side (:74-117): the client half only -------------------------------------------------------------
    def register_to_controller(self):
        import requests
        r = requests.post(self.controller_addr + "/register_worker", json={
            "worker_name": self.worker_addr, "check_heart_beat": True, "worker_status": self.get_status()})
        assert r.status_code == 200

    def _heart_beat_loop(self):
        while True:
            time.sleep(WORKER_HEART_BEAT_INTERVAL)
            self.send_heart_beat()

    def send_heart_beat(self):
        if self.heart_beat_thread is None:                             # --no-register: there is nobody to tell
            return
        import requests
        while True:
            try:
                ret = requests.post(self.controller_addr + "/receive_heart_beat", json={
                    "worker_name": self.worker_addr, "queue_length": self.get_queue_length()}, timeout=5)
                exist = ret.json()["exist"]
                break
            except requests.exceptions.RequestException:
                time.sleep(5)
        if not exist:
            self.register_to_controller()

    def get_queue_length(self):                                        # :106-111
        sem = self.model_semaphore
        if sem is None:
            return 0
        waiters = getattr(sem, "_waiters", None)
        return self.limit_model_concurrency - sem._value + (len(waiters) if waiters is not None else 0)

    def get_status(self):                                              # :113-118
        return {"model_names": [self.model_name], "speed": 1, "queue_length": self.get_queue_length()}

    # ---- the request (:120-181) -----------------------------------------------------------------------------------------
    def _device(self):
        if isinstance(self.device, torch.device):
            return self.device
        if self.device == "cuda" and hasattr(self.model, "engine"):
            return torch.device("cuda", self.model.engine.device)
        return torch.device(self.device)

    def generate_stream(self, params) -> Iterator[bytes]:
        model, image_processor, task = self.model, self.image_processor, self.task
        num_beams = int(params.get("num_beams", 1))
        temperature = float(params.get("temperature", 1.0))
        len_penalty = float(params.get("len_penalty", 1.0))
        top_p = float(params.get("top_p", 1.0))
        max_context_length = getattr(model.config, "max_position_embeddings", 8192)
        streamer = TextQueueStreamer(self.tokenizer, skip_prompt=False, skip_special_tokens=True, timeout=15)
        prompt = params["prompt"]
        dev = self._device()

        if task == "Image2SVG":
            image = None
            for b64_image in params.get("images", None) or []:         # the last image wins, as in :133-139
                if b64_image is not None and self.is_multimodal:
                    image = process_images(load_image_from_base64(b64_image), image_processor).to(dev, dtype=torch.bfloat16)
                else:
                    image = None
            if image is None:
                raise ValueError("Image2SVG request without an image")
            max_new_tokens = min(int(params.get("max_new_tokens", 256)), 8192)
            max_new_tokens = min(max_new_tokens, max_context_length - CLIP_QUERY_LENGTH)
            pre_pend = prompt
            batch = {"image": image}
            generate_method = model.model.generate_im2svg
        else:
            max_new_tokens = min(int(params.get("max_new_tokens", 128)), 8192)
            pre_pend = ""
            batch = {"caption": [prompt], "image": torch.zeros((3, 256, 256), dtype=torch.bfloat16, device=dev)}
            generate_method = model.model.generate_text2svg

        if max_new_tokens < 1:
            yield json.dumps({"text": prompt + "Exceeds max token length. Please start a new conversation, thanks.",
                              "error_code": 0}).encode() + b"\0"
            return

        failure = []

        def run():
            try:
                with torch.inference_mode():
                    generate_method(batch=batch, prompt=prompt, use_nucleus_sampling=True, num_beams=num_beams,
                                    temperature=temperature, length_penalty=len_penalty, top_p=top_p,
                                    max_length=max_new_tokens, streamer=streamer)
            except BaseException as e:                                 # carried to the consumer below
                failure.append(e)
                streamer.text_queue.put(streamer._END)

        thread = threading.Thread(target=run, daemon=True)
        thread.start()
        generated_text = pre_pend
        for new_text in streamer:
            if not new_text:             # :176-177 skips a chunk that is exactly " ", which is how TextStreamer releases the
                continue                 # leading space of " width" after a flush: that loses real spaces, so only "" is skipped
            generated_text += new_text
            yield json.dumps({"text": generated_text, "error_code": 0}).encode() + b"\0"
        thread.join()
        if failure:
            raise failure[0]

    def generate_stream_gate(self, params) -> Iterator[bytes]:        # :183-207: every failure is one error object
        try:
            for x in self.generate_stream(params):
                yield x
        except Exception as e:                                         # ValueError, engine errors, anything else
            print(f"Caught {type(e).__name__}:", e)
            yield json.dumps({"text": server_error_msg, "error_code": 1}).encode() + b"\0"


def build_app(worker: ModelWorker):
    """The two routes of serve/model_worker.py:209-233 around `worker`."""
    import asyncio
    from functools import partial
    from fastapi import BackgroundTasks, FastAPI, Request
    from fastapi.responses import StreamingResponse

    app = FastAPI()

    def release(fn=None):
        worker.model_semaphore.release()
        if fn is not None:
            fn()

    @app.post("/worker_generate_stream")
    async def generate_stream(request: Request):
        worker.global_counter += 1
        params = await request.json()
        if worker.model_semaphore is None:
            worker.model_semaphore = asyncio.Semaphore(worker.limit_model_concurrency)
        await worker.model_semaphore.acquire()
        worker.send_heart_beat()
        background_tasks = BackgroundTasks()
        background_tasks.add_task(partial(release, fn=worker.send_heart_beat))
        return StreamingResponse(worker.generate_stream_gate(params), background=background_tasks)

    @app.post("/worker_get_status")
    async def get_status(request: Request):
        return worker.get_status()

    return app


def main(argv=None):                                                   # :235-269, same flags
    p = argparse.ArgumentParser()
    p.add_argument("--host", type=str, default="localhost")
    p.add_argument("--port", type=int, default=21002)
    p.add_argument("--worker-address", type=str, default="http://localhost:21002")
    p.add_argument("--controller-address", type=str, default="http://localhost:21001")
    p.add_argument("--model-path", type=str, required=True, help="local checkpoint directory (there is no hub access)")
    p.add_argument("--model-name", type=str)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--limit-model-concurrency", type=int, default=5)
    p.add_argument("--no-register", action="store_true")
    p.add_argument("--no-continuous-batching", action="store_true",
                   help="run concurrent requests one generate call at a time (the reference's behaviour)")
    args = p.parse_args(argv)
    import uvicorn
    worker = ModelWorker(args.controller_address, args.worker_address, str(uuid.uuid4())[:6], args.no_register,
                         args.model_path, args.model_name, args.device, args.limit_model_concurrency,
                         continuous_batching=not args.no_continuous_batching)
    uvicorn.run(build_app(worker), host=args.host, port=args.port, log_level="info")


if __name__ == "__main__":
    main()
