"""Continuous batching on the host side.

The reference's worker lets 5 requests in at a time (serve/model_worker.py:161-172,216-229: an asyncio semaphore around
`generate_stream`) and runs each as its own HF `generate` call on its own thread; on one GPU those calls simply take turns.
Here concurrent requests share ONE decode loop: `ContinuousBatcher` owns a scheduler thread that

  * admits waiting requests into free rows ("slots") of the engine's batch -- a prompt pass for the newcomers only, while the
    running requests keep their KV pages (`sv_cb_admit`),
  * advances every live slot by a few decode steps per iteration (`sv_cb_step`: one captured hipGraph per row bucket),
  * hands each request the tokens that became final since the last look (streaming) and frees the slots of finished requests.

Every slot has its own sampling parameters, budget, EOS, stop sequence and random stream, so a request's tokens are exactly
what a solo `HipEngine.generate` call would return for it (tests/test_gpu_serving.py).  No kernels here: host logic only.
"""
from __future__ import annotations

import threading
from collections import deque
from typing import Callable, Deque, Dict, List, Optional

import torch

from ._lib import StarVectorBusy


class _Request:
    def __init__(self, emb: Optional[torch.Tensor], params: dict, on_tokens: Optional[Callable], exclusive: Optional[Callable] = None):
        self.emb, self.params, self.on_tokens = emb, params, on_tokens
        self.exclusive = exclusive         # a callable that needs the engine to itself (beam search, scoring forward)
        self.value = None
        self.slot: Optional[int] = None
        self.sent = 0                      # tokens already handed over
        self.chunks: List[torch.Tensor] = []
        self.done = threading.Event()
        self.error: Optional[BaseException] = None

    def result(self, timeout: Optional[float] = None) -> torch.Tensor:
        """Blocks until the request has finished; int64 [1, N] new tokens (HF `generate` shape for one sequence)."""
        if not self.done.wait(timeout):
            raise TimeoutError("generation did not finish in time")
        if self.error is not None:
            raise self.error
        if self.exclusive is not None:
            return self.value
        toks = torch.cat(self.chunks) if self.chunks else torch.empty(0, dtype=torch.int64)
        return toks.view(1, -1)


class ContinuousBatcher:
    """engine: a `HipEngine` (or anything with cb_admit / cb_step / cb_poll / cb_read / cb_release / cb_reset and `.cfg`)."""

    def __init__(self, engine, steps_per_poll: int = 8, device: Optional[int] = None):
        self.engine = engine
        self.steps_per_poll = int(steps_per_poll)
        self.device = getattr(engine, "device", 0) if device is None else device
        self._lock = threading.Condition()
        self._pending: Deque[_Request] = deque()
        self._active: Dict[int, _Request] = {}
        self._closing = False
        self.max_concurrent = 0            # high-water mark of requests sharing the decode loop (tests, status)
        self.steps_run = 0
        self._thread = threading.Thread(target=self._loop, name="sv-continuous-batcher", daemon=True)
        self._thread.start()

    # ---- producer side -------------------------------------------------------------------------------------------------
    def submit(self, inputs_embeds: torch.Tensor, params: dict, on_tokens: Optional[Callable] = None) -> _Request:
        """inputs_embeds [1, S0, D]; params: the keys of `HipEngine.cb_admit` (max_new_tokens required);
        on_tokens(tokens int64 [k], first_index) is called from the scheduler thread as tokens become final."""
        if inputs_embeds.dim() != 3 or inputs_embeds.shape[0] != 1:
            raise ValueError("one sequence per request: inputs_embeds must be [1, S0, D]")
        if int(params.get("max_new_tokens", 0)) < 1:
            raise ValueError("max_new_tokens must be >= 1")
        req = _Request(inputs_embeds, dict(params), on_tokens)
        with self._lock:
            if self._closing:
                raise RuntimeError("the batcher is closed")
            self._pending.append(req)
            self._lock.notify_all()
        return req

    def generate(self, inputs_embeds: torch.Tensor, params: dict, on_tokens: Optional[Callable] = None,
                 timeout: Optional[float] = None) -> torch.Tensor:
        return self.submit(inputs_embeds, params, on_tokens).result(timeout)

    def run_exclusive(self, fn: Callable, timeout: Optional[float] = None):
        """Run `fn()` with the engine to itself: what the slots cannot express (beam search keeps its own search state in
        the batch rows, the scoring forward wants the whole KV pool).  FIFO with the generation requests: the ones admitted
        before it finish first, the ones behind it wait; the continuous batch is reset around the call."""
        req = _Request(None, {}, None, exclusive=fn)
        with self._lock:
            if self._closing:
                raise RuntimeError("the batcher is closed")
            self._pending.append(req)
            self._lock.notify_all()
        return req.result(timeout)

    def queue_length(self) -> int:
        with self._lock:
            return len(self._pending) + len(self._active)

    def close(self):
        with self._lock:
            self._closing = True
            self._lock.notify_all()
        self._thread.join(timeout=30)

    # ---- scheduler thread ----------------------------------------------------------------------------------------------
    def _admit(self) -> int:
        """Move waiting requests into free slots, one `cb_admit` per prompt length (the prompt pass is rectangular).
        Returns how many requests left the queue (admitted, or failed alone)."""
        moved = 0
        with self._lock:
            waiting = []
            for r in self._pending:                 # FIFO up to the first exclusive job: nothing overtakes it
                if r.exclusive is not None:
                    break
                waiting.append(r)
        by_len: Dict[int, List[_Request]] = {}
        for r in waiting:
            by_len.setdefault(int(r.emb.shape[1]), []).append(r)
        for S0, group in by_len.items():
            room = self.engine.cfg.max_batch - len(self._active)
            group = group[:room]
            while group:
                try:
                    slots = self.engine.cb_admit(torch.cat([r.emb for r in group], 0), [r.params for r in group])
                except StarVectorBusy:
                    group = group[:-1]              # KV pages are short: try fewer, the rest waits for a release
                    continue
                except BaseException as e:          # a bad request must not take the loop down: it fails alone
                    bad = group if len(group) == 1 else None
                    if bad is None:
                        group = group[:1]
                        continue
                    self._finish(bad[0], e)
                    with self._lock:
                        self._pending.remove(bad[0])
                    moved += 1
                    break
                with self._lock:
                    for r, s in zip(group, slots):
                        r.slot = s
                        self._active[s] = r
                        self._pending.remove(r)
                    self.max_concurrent = max(self.max_concurrent, len(self._active))
                moved += len(group)
                break
        return moved

    def _finish(self, req: _Request, error: Optional[BaseException] = None):
        req.error = error
        req.done.set()

    def _deliver(self):
        live, steps = self.engine.cb_poll()
        for s, req in list(self._active.items()):
            n = steps[s]
            if n > req.sent:
                toks = self.engine.cb_read(s, req.sent, n - req.sent)
                req.chunks.append(toks)
                first, req.sent = req.sent, n
                if req.on_tokens is not None:
                    try:
                        req.on_tokens(toks, first)
                    except BaseException as e:      # the consumer failed: stop this request, keep the others
                        self.engine.cb_release(s)
                        del self._active[s]
                        self._finish(req, e)
                        continue
            if not live[s]:
                self.engine.cb_release(s)
                del self._active[s]
                self._finish(req)

    def _loop(self):
        try:
            if torch.cuda.is_available():
                torch.cuda.set_device(self.device)
        except Exception:
            pass
        try:
            while True:
                with self._lock:
                    while not self._pending and not self._active and not self._closing:
                        self._lock.wait()
                    if self._closing and not self._pending and not self._active:
                        break
                with self._lock:
                    head = self._pending[0] if self._pending else None
                try:
                    if head is not None and head.exclusive is not None:
                        if not self._active:        # the engine is idle: hand it over
                            with self._lock:
                                self._pending.popleft()
                            try:
                                self.engine.cb_reset()
                                head.value = head.exclusive()
                                self._finish(head)
                            except BaseException as e:
                                self._finish(head, e)
                            continue
                    elif self._pending and len(self._active) < self.engine.cfg.max_batch:
                        moved = self._admit()
                        if moved == 0 and not self._active:
                            # nothing runs and the head request still does not fit: no release will ever make room.  Reset the
                            # continuous batch once (recovers anything a failed admit left behind); if it still does not fit,
                            # the request fails with an error instead of the loop spinning on it forever.
                            self.engine.cb_reset()
                            if self._admit() == 0:
                                with self._lock:
                                    head = self._pending.popleft() if self._pending else None
                                if head is not None:
                                    self._finish(head, StarVectorBusy(
                                        "the request does not fit an idle engine (KV pages / slots for prompt + max_new_tokens)"))
                                continue
                        self._deliver()             # first tokens (and one-token requests) right away
                    if self._active:
                        self.engine.cb_step(self.steps_per_poll)
                        self.steps_run += self.steps_per_poll
                        self._deliver()
                except Exception as e:              # the engine failed under this batch (a device error, a row of non-finite
                    with self._lock:                # logits): the requests SHARING the batch learn about it at once, the
                        reqs = list(self._active.values())      # waiting ones stay queued and the loop lives on -- one
                        self._active.clear()                    # poisoned request must not take the worker down
                    for r in reqs:
                        self._finish(r, e)
                    try:
                        self.engine.cb_reset()
                    except Exception:
                        pass
        except BaseException as e:                  # interpreter shutdown and the like: every request learns about it at once
            with self._lock:
                reqs = list(self._active.values()) + list(self._pending)
                self._active.clear()
                self._pending.clear()
                self._closing = True
            for r in reqs:
                self._finish(r, e)
        finally:
            try:
                self.engine.cb_reset()
            except Exception:
                pass
