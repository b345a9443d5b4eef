"""CPU: the continuous-batching scheduler (star-vector_amd/batching.py) over a scripted engine with the sv_cb_* surface --
admission by prompt length, SV_EBUSY back-off, per-request streams, failure isolation, the worker wiring."""
import threading
import time

import pytest
import torch

from starvector_amd._lib import StarVectorBusy
from starvector_amd.batching import ContinuousBatcher
from starvector_amd.engine import EngineConfig


class _CbEngine:
    """Rows are requests; request with prompt mean m emits m, m+1, ... for its budget (EOS = 999 cuts it)."""
    device = 0

    def __init__(self, max_batch=4, page_budget=10 ** 9, delay=0.0):
        self.delay = delay
        self.cfg = EngineConfig(image_size=28, patch_size=14, vit_width=4, hidden=8, vocab=1000, max_batch=max_batch)
        self.slots = {}
        self.page_budget = page_budget
        self.admits, self.steps, self.lock = [], 0, threading.Lock()

    def cb_admit(self, emb, reqs):
        with self.lock:
            free = [s for s in range(self.cfg.max_batch) if s not in self.slots]
            need = sum(r["max_new_tokens"] for r in reqs)
            used = sum(v["budget"] for v in self.slots.values())
            if len(free) < len(reqs) or used + need > self.page_budget:
                raise StarVectorBusy("busy")
            if any(r["max_new_tokens"] > 500 for r in reqs):
                raise ValueError("max_new_tokens out of range")
            out = []
            for i, r in enumerate(reqs):
                s = free[i]
                base = int(round(float(emb[i].float().mean())))
                self.slots[s] = dict(base=base, budget=r["max_new_tokens"], toks=[base], live=r["max_new_tokens"] > 1,
                                     eos=r.get("eos_token_id", -1))
                out.append(s)
            self.admits.append((emb.shape[1], len(reqs)))
            return out

    def cb_step(self, n):
        if self.delay:
            time.sleep(self.delay)
        with self.lock:
            for _ in range(n):
                self.steps += 1
                for v in self.slots.values():
                    if v["live"]:
                        t = v["base"] + len(v["toks"])
                        v["toks"].append(t)
                        if len(v["toks"]) >= v["budget"] or t == v["eos"]:
                            v["live"] = False
            return sum(v["live"] for v in self.slots.values())

    def cb_poll(self):
        with self.lock:
            n = self.cfg.max_batch
            return ([int(self.slots[s]["live"]) if s in self.slots else 0 for s in range(n)],
                    [len(self.slots[s]["toks"]) if s in self.slots else 0 for s in range(n)])

    def cb_read(self, slot, first, count):
        with self.lock:
            return torch.tensor(self.slots[slot]["toks"][first:first + count], dtype=torch.int64)

    def cb_release(self, slot):
        with self.lock:
            del self.slots[slot]

    def cb_reset(self):
        with self.lock:
            self.slots.clear()


def _emb(m, S0=3):
    return torch.full((1, S0, 8), float(m))


def test_concurrent_requests_share_one_loop_and_keep_their_own_streams():
    eng = _CbEngine(max_batch=4, delay=0.01)           # a step takes time: requests arriving meanwhile join the live batch
    b = ContinuousBatcher(eng, steps_per_poll=2)
    got, chunks = {}, {i: [] for i in range(6)}

    def run(i):
        got[i] = b.generate(_emb(10 * i, S0=3 + i % 2), dict(max_new_tokens=5 + i), lambda t, f, i=i: chunks[i].append((f, t.tolist())))

    th = [threading.Thread(target=run, args=(i,)) for i in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=30)
    for i in range(6):
        assert got[i].shape == (1, 5 + i) and got[i][0].tolist() == [10 * i + k for k in range(5 + i)]
        flat = [x for _, c in chunks[i] for x in c]
        assert flat == got[i][0].tolist() and [f for f, _ in chunks[i]] == sorted(f for f, _ in chunks[i])   # contiguous bursts
    assert b.max_concurrent >= 3                        # they overlapped inside the engine's batch
    assert all(S0 in (3, 4) for S0, _ in eng.admits)    # one prompt pass per prompt length
    assert not eng.slots                                # every slot was released
    b.close()


def test_busy_engine_makes_requests_wait_not_fail():
    eng = _CbEngine(max_batch=4, page_budget=12)        # room for two 6-token requests at a time
    b = ContinuousBatcher(eng, steps_per_poll=1)
    reqs = [b.submit(_emb(i), dict(max_new_tokens=6)) for i in range(5)]
    for i, r in enumerate(reqs):
        assert r.result(timeout=30)[0].tolist() == [i + k for k in range(6)]
    assert b.max_concurrent <= 2
    b.close()


def test_a_bad_request_fails_alone_and_consumer_errors_do_not_stop_the_others():
    eng = _CbEngine(max_batch=4)
    b = ContinuousBatcher(eng, steps_per_poll=1)
    ok1 = b.submit(_emb(1), dict(max_new_tokens=8))
    bad = b.submit(_emb(2), dict(max_new_tokens=900))   # the engine refuses it (ValueError)
    def boom(t, f):
        raise RuntimeError("consumer went away")
    cons = b.submit(_emb(3), dict(max_new_tokens=8), boom)
    ok2 = b.submit(_emb(4), dict(max_new_tokens=3, eos_token_id=5))        # EOS = its second token
    assert ok1.result(timeout=30)[0].tolist() == list(range(1, 9))
    with pytest.raises(ValueError):
        bad.result(timeout=30)
    with pytest.raises(RuntimeError):
        cons.result(timeout=30)
    assert ok2.result(timeout=30)[0].tolist() == [4, 5]
    with pytest.raises(ValueError):
        b.submit(torch.zeros(2, 3, 8), dict(max_new_tokens=4))             # one sequence per request
    b.close()
    with pytest.raises(RuntimeError):
        b.submit(_emb(1), dict(max_new_tokens=2))


def test_engine_failure_reaches_every_request():
    eng = _CbEngine(max_batch=4, delay=0.01)

    def broken(n):
        raise RuntimeError("device fault (scripted)")
    b = ContinuousBatcher(eng, steps_per_poll=1)
    r1 = b.submit(_emb(1), dict(max_new_tokens=400))
    time.sleep(0.05)
    eng.cb_step = broken
    r2 = b.submit(_emb(2), dict(max_new_tokens=400))
    for r in (r1, r2):
        with pytest.raises(RuntimeError):
            r.result(timeout=30)
    b.close()


def test_the_loop_survives_an_engine_failure_and_serves_the_next_request():
    """A failure under one batch (e.g. a request whose logits went non-finite) fails the requests sharing that batch; the
    scheduler resets the slots and keeps serving."""
    eng = _CbEngine(max_batch=4, delay=0.005)
    real_step = eng.cb_step
    state = {"fail": True}

    def flaky(n):
        if state["fail"]:
            state["fail"] = False
            raise RuntimeError("a row of logits had no finite value (scripted)")
        return real_step(n)
    eng.cb_step = flaky
    b = ContinuousBatcher(eng, steps_per_poll=2)
    r1 = b.submit(_emb(1), dict(max_new_tokens=12))
    with pytest.raises(RuntimeError):
        r1.result(timeout=30)
    r2 = b.submit(_emb(2), dict(max_new_tokens=12))           # the same batcher, afterwards
    assert r2.result(timeout=30).shape == (1, 12)
    b.close()


def test_mirror_generate_routes_single_sequences_through_the_batcher():
    """HipCausalLM.generate with a batcher attached: B = 1, num_beams = 1 calls become slots; the HF kwargs are mapped."""
    from starvector_amd.model import HipCausalLM
    eng = _CbEngine(max_batch=4, delay=0.01)
    lm = HipCausalLM.__new__(HipCausalLM)
    torch.nn.Module.__init__(lm)
    object.__setattr__(lm, "_engine", eng)
    lm.eos_token_id, lm.pad_token_id, lm.seed = 999, 998, None
    lm.batcher = ContinuousBatcher(eng, steps_per_poll=2)
    out = [None] * 4

    def run(i):
        out[i] = lm.generate(inputs_embeds=_emb(20 * i), max_length=3 + 6, do_sample=(i % 2 == 1), top_p=0.9, temperature=0.7)

    th = [threading.Thread(target=run, args=(i,)) for i in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=30)
    for i in range(4):
        assert out[i].tolist() == [[20 * i + k for k in range(6)]]
    assert lm.batcher.max_concurrent >= 2
    lm.batcher.close()


def test_exclusive_jobs_take_the_engine_in_turn():
    """Beam search / multi-row HF batches cannot be slots: `run_exclusive` gives them the engine between the requests that
    were admitted before and those that came after (FIFO), with the continuous batch reset around the call."""
    eng = _CbEngine(max_batch=4, delay=0.005)
    b = ContinuousBatcher(eng, steps_per_poll=1)
    order = []
    r1 = b.submit(_emb(1), dict(max_new_tokens=20), lambda t, f: order.append("r1"))

    def job():
        assert not eng.slots                                # nothing is live while the job owns the engine
        order.append("job")
        return 42
    res = {}
    th = threading.Thread(target=lambda: res.setdefault("v", b.run_exclusive(job)))
    th.start()
    time.sleep(0.02)
    r2 = b.submit(_emb(2), dict(max_new_tokens=4), lambda t, f: order.append("r2"))
    assert r1.result(timeout=30)[0].tolist() == list(range(1, 21))
    th.join(timeout=30)
    assert res["v"] == 42 and r2.result(timeout=30)[0].tolist() == [2, 3, 4, 5]
    j = order.index("job")
    assert "r1" in order[:j] and "r1" not in order[j:] and "r2" not in order[:j]       # r1 finished, then the job, then r2
    with pytest.raises(ZeroDivisionError):
        b.run_exclusive(lambda: 1 / 0)                      # the job's exception reaches its caller, the loop lives on
    assert b.generate(_emb(7), dict(max_new_tokens=3))[0].tolist() == [7, 8, 9]
    b.close()


def test_request_that_never_fits_fails_instead_of_spinning_the_scheduler():
    """A request whose budget exceeds what an IDLE engine can hold gets SV_EBUSY from every admit: with nothing running no
    release will ever make room, so it must fail with an error (and the ones behind it be served), not keep the scheduler
    thread in a busy loop while its caller hangs."""
    eng = _CbEngine(max_batch=4, page_budget=12)
    b = ContinuousBatcher(eng, steps_per_poll=1)
    big = b.submit(_emb(1), dict(max_new_tokens=40))            # 40 > the 12-token page budget of the whole engine
    ok = b.submit(_emb(7), dict(max_new_tokens=4))
    with pytest.raises(StarVectorBusy):
        big.result(timeout=20)
    assert ok.result(timeout=20)[0].tolist() == [7, 8, 9, 10]
    assert not eng.slots
    b.close()


def test_padded_batches_from_two_threads_do_not_reset_each_other():
    """The mirror runs a padded multi-row batch as slots: cb_reset, cb_admit per length group, cb_step ..., cb_read, cb_reset.
    Two host threads doing that on one engine must take turns for the WHOLE sequence (the engine's call_lock), otherwise one
    thread's cb_reset releases the other's slots."""
    from starvector_amd.model import HipCausalLM

    class _Eng(_CbEngine):
        def __init__(self):
            super().__init__(max_batch=4, delay=0.005)
            self.call_lock = threading.RLock()
            self.resets_while_live = 0

        def cb_reset(self):
            with self.lock:
                if any(v["live"] for v in self.slots.values()):
                    self.resets_while_live += 1
            super().cb_reset()

    eng = _Eng()
    lm = HipCausalLM.__new__(HipCausalLM)
    torch.nn.Module.__init__(lm)
    object.__setattr__(lm, "_engine", eng)
    lm.eos_token_id, lm.pad_token_id, lm.seed, lm.batcher = -1, 0, None, None
    out, err = {}, []

    def run(i):
        try:
            emb = torch.stack([torch.full((4, 8), float(10 * i)), torch.full((4, 8), float(10 * i + 5))])
            mask = torch.tensor([[1, 1, 1, 1], [0, 1, 1, 1]])          # row 1 is left-padded: two length groups
            out[i] = lm.generate(inputs_embeds=emb, attention_mask=mask, max_length=4 + 6, eos_token_id=-1, pad_token_id=0)
        except BaseException as e:       # noqa: BLE001
            err.append(e)

    th = [threading.Thread(target=run, args=(i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=60)
    assert not err, err
    assert eng.resets_while_live == 0
    for i in range(3):
        assert out[i].shape == (2, 6)
        assert out[i][0].tolist() == [10 * i + k for k in range(6)] and out[i][1].tolist() == [10 * i + 5 + k for k in range(6)]
