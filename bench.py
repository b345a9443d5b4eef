#!/usr/bin/env python3
"""Headline benchmark: StarVector-1B im2svg, batch 32 per GPU, bf16, greedy, 224x224 synthetic images,
random-init weights (BASELINE.json configs[1]; configs[2] = the same shard on each of N GPUs).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch already resident in HBM:
image encoder -> adapter -> prompt embedding -> decoder prefill -> N_NEW greedy decode steps -> token ids
(rows a1-a11 of SURVEY.md section 8a; detokenisation is host Python in the reference too and is outside
the timed region).  Multi-GPU: the global batch is sharded by rank, no collective inside the path, ONE
all_gather of the token streams per step (RCCL), weak scaling (32 images per GPU).

Prints ONE JSON line (rank 0).  value = generated SVG tokens / second over ALL GPUs.
"""
from __future__ import annotations

import os

# Before ANYTHING touches the HIP / HSA runtime (it reads its flags once, at initialisation): the host driver of this pool only
# supports dmabuf IPC, and without this RCCL's hipIpcGetMemHandle fails (a setdefault after torch.cuda.* is too late).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import argparse   # noqa: E402
import json       # noqa: E402
import socket     # noqa: E402
import statistics  # noqa: E402
import sys        # noqa: E402
import time       # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROMPT_IDS = [7, 11]          # '<svg' is 2 ids under the (gated, offline) StarCoder tokenizer: synthetic stand-ins
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8.0 TB/s spec


def decoder_weight_bytes(cfg) -> int:
    """Algorithmic bytes of the dominant kernel per decode step: every decoder Linear weight + the tied lm_head,
    bf16, streamed once (2,240,876,544 B for StarVector-1B)."""
    head_dim = cfg.hidden // cfg.n_head
    qkv = cfg.n_head * head_dim + 2 * cfg.n_kv_head * head_dim
    per_layer = cfg.hidden * (qkv + cfg.n_head * head_dim + 2 * cfg.n_inner)
    return 2 * (cfg.n_layer * per_layer + cfg.vocab * cfg.hidden)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--new-tokens", type=int, default=1024)
    ap.add_argument("--model", choices=["1b", "8b"], default="1b",
                    help="1b: BASELINE config 2 (batch 32, greedy); 8b: config 4 (StarVector-8B, batch 16, top-p 0.95)")
    ap.add_argument("--weights", choices=["bf16", "fp8"], default="bf16",
                    help="fp8: decoder weights + lm_head quantised to e4m3 at load (BASELINE config 5's weight format); "
                         "NOT the reference precision -- never the headline line")
    ap.add_argument("--task", choices=["im2svg", "text2svg"], default="im2svg",
                    help="text2svg: BASELINE config 5's workload (no image encoder; the prompt is 32 caption ids + <svg-start>, "
                         "batch 64 per GPU with --model 8b) -- a secondary line, never the headline")
    ap.add_argument("--beams", type=int, default=1,
                    help="num_beams of the decode (1: BASELINE's greedy line).  2 with --sample = the reference's DEFAULT generate_im2svg call "
                         "(starvector_base.py:231-239: beam-sample, num_beams 2, top-p 0.9): batch x beams rows per decode step -- a secondary line")
    ap.add_argument("--sample", action="store_true", help="with --beams > 1: HF beam-sample (do_sample, top-p 0.9, temperature 1.0)")
    ap.add_argument("--top-k", type=int, default=50,
                    help="with --sample: TopKLogitsWarper in front of top-p.  50 = what the reference's call really runs (it never passes top_k, and its pinned "
                         "transformers 4.49 defaults GenerationConfig.top_k to 50); 0 = top-p over the whole vocabulary (the line filed before this flag existed)")
    ap.add_argument("--batch", type=int, default=0, help="requests per GPU (default: the BASELINE configuration's: 32 / 16 / 64); other values are sweep points, not the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ttft-requests", type=int, default=20)
    return ap.parse_args()


def cpu_baseline():
    """The oracle (CPU float32 restatement of the reference modules, BASELINE config 1) timed on this box's host cores, bounded
    sample: one image, prefill + 12 greedy decode steps.  The ONLY place bench.py touches oracle/ (test infrastructure): its own
    N(0, 0.02) weights of the same architecture -- values do not change a dense float32 forward's time."""
    import torch
    from oracle import starvector_oracle as O
    from oracle.hostinfo import host_cores
    cfg = O.OracleConfig()
    torch.set_num_threads(host_cores())
    cores = torch.get_num_threads()
    w = O.make_weights(cfg, seed=1234, init="std002")
    img = O.synthetic_images(1, cfg.image_size, seed=3)
    prompt = torch.tensor([PROMPT_IDS], dtype=torch.long)
    t0 = time.perf_counter()
    emb = O.prepare_generation_inputs(w, cfg, img, prompt)
    t1 = time.perf_counter()
    logits, cache = O.decoder_prefill(w, cfg, emb)
    tok = logits.argmax(-1)
    t2 = time.perf_counter()
    n = 12
    for _ in range(n):
        logits, cache = O.decoder_decode_step(w, cfg, tok, cache)
        tok = logits.argmax(-1)
    t3 = time.perf_counter()
    return {"value": round(n / (t3 - t2), 3), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"oracle fp32, batch 1, one 224x224 image: encoder+adapter {t1 - t0:.2f}s, prefill(259)+first "
                      f"token {t2 - t1:.2f}s (TTFT {t2 - t0:.2f}s), then {n} greedy decode steps at context 260-271",
            "ttft_s": round(t2 - t0, 3)}


_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)      # data/util.py:33-38 (ImageTrainProcessor's Normalize)
_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def synthetic_images(torch, n, size, seed):
    """Random-pixel images through the reference's normalisation: row i depends on (seed, i) only."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 3, size, size, generator=g)
    mean, std = torch.tensor(_CLIP_MEAN).view(1, 3, 1, 1), torch.tensor(_CLIP_STD).view(1, 3, 1, 1)
    return ((x - mean) / std).to(torch.bfloat16)


def self_launch_command(gpus: int, argv, port=None):
    """`python bench.py --gpus N` without a launcher: the command line this process replaces itself with -- one rank per GPU under
    torch.distributed.run, rendezvous on 127.0.0.1 (the container hostname may not resolve), a free port."""
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # called bare (`python bench.py --gpus 8`): become the launcher the contract names instead of giving up
        cmd = self_launch_command(args.gpus, sys.argv[1:])
        print("[bench] no launcher in the environment: re-executing as " + " ".join(cmd), file=sys.stderr, flush=True)
        os.execvpe(cmd[0], cmd, os.environ)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s): they must agree (one process per GPU)")
    # One process per GPU.  With fewer GPUs than ranks (the 1-GPU development box) the ranks share devices and rendezvous
    # over gloo: a FUNCTIONAL run of the very same sharded path (full replica per rank, rank shard of the global batch, one
    # all_gather of token streams); its rate is not a scaling number and the line says so ("dist_backend").
    n_dev = torch.cuda.device_count()
    shared = world > n_dev or os.environ.get("SV_DIST_BACKEND", "") == "gloo"
    dev_index = local_rank % max(n_dev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import datetime
    import torch.distributed as dist
    rccl_version = None
    if world > 1:
        # Fail LOUDLY, never hang: a bounded rendezvous / collective timeout, and a first tiny all_reduce right away so that a
        # broken xGMI / RCCL setup (e.g. a missing HSA_ENABLE_IPC_MODE_LEGACY=0: hipIpcGetMemHandle fails) surfaces here with a
        # message instead of inside the timed region.
        tmo = datetime.timedelta(seconds=int(os.environ.get("SV_DIST_TIMEOUT_S", "180")))
        try:
            if shared:
                dist.init_process_group("gloo", timeout=tmo)
            else:
                dist.init_process_group("nccl", device_id=dev, timeout=tmo)       # "nccl" is RCCL on ROCm
                try:
                    rccl_version = ".".join(str(x) for x in torch.cuda.nccl.version())
                except Exception:
                    rccl_version = "unknown"
            probe = torch.ones(1, device="cpu" if shared else dev)
            dist.all_reduce(probe)
            if not shared:
                torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError(f"all_reduce probe returned {probe.item()} for world size {world}")
        except Exception as e:
            print(f"[bench rank {rank}] distributed setup FAILED ({'gloo' if shared else 'nccl/RCCL'}, world {world}, "
                  f"MASTER_ADDR={os.environ.get('MASTER_ADDR')}:{os.environ.get('MASTER_PORT')}): {type(e).__name__}: {e}",
                  file=sys.stderr, flush=True)
            raise SystemExit(3)

    import starvector_amd as sva
    from starvector_amd.parallel import all_gather_token_streams

    is8b = args.model == "8b"
    t2s = args.task == "text2svg"
    B_PER_GPU = (64 if t2s else 16) if is8b else 32
    if args.batch > 0:
        B_PER_GPU = args.batch
    n_new = args.new_tokens
    CAPTION_TOKENS = 32
    PAD_ID = 0 if is8b else 49152            # llm/starcoder2.py:47 / llm/starcoder.py:40-53 ([PAD] appended to the 49152-entry vocabulary)
    NB = max(int(args.beams), 1)
    ROWS = B_PER_GPU * NB                     # rows of a decode step: every beam is a row of the engine's batch
    ec = sva.EngineConfig.starvector_8b(max_batch=ROWS, max_seq_len=16) if is8b else sva.EngineConfig(max_batch=ROWS)
    S0 = CAPTION_TOKENS + 1 if t2s else ec.query_length + len(PROMPT_IDS)
    ec.max_seq_len = S0 + n_new
    cfg = ec                                  # the shapes of the path come from the product's own config (StarVectorConfig's defaults)
    W_BYTES_PER_STEP = decoder_weight_bytes(cfg) // (2 if args.weights == "fp8" else 1)
    t_setup = time.time()
    if args.weights == "fp8":
        ec.weight_dtype = "fp8_e4m3"
    # one process per GPU is the deployment this benchmark measures: the engine owns its device (all-blocks-resident fused launches on);
    # ranks SHARING a GPU (the functional gloo run on a 1-GPU box) must not claim that
    ec.exclusive_device = not shared and os.environ.get("SV_SHARED_GPU", "") != "1"
    eng = sva.HipEngine(ec, device=dev_index)
    keep_cpu = (world == 1 and rank == 0 and not args.no_cpu_baseline and not is8b and not t2s and NB == 1)   # 8B fp32 on CPU: 29 GB, skipped
    eng.load_random_weights(seed=1234, std=0.02)   # drawn on the GPU, one tensor at a time, same values on every rank
    # this rank's shard of the global batch (seeded per global row -> identical under any sharding)
    images = synthetic_images(torch, B_PER_GPU * world, cfg.image_size, seed=0)[rank * B_PER_GPU:(rank + 1) * B_PER_GPU].to(dev)
    prompt = torch.tensor([PROMPT_IDS] * B_PER_GPU, dtype=torch.long, device=dev)
    if t2s:
        # starvector_base.py:297-330: caption ids + <svg-start>; seeded per global row like the images
        g = torch.Generator().manual_seed(0)
        caps = torch.randint(1, 49152, (B_PER_GPU * world, CAPTION_TOKENS), generator=g)[rank * B_PER_GPU:(rank + 1) * B_PER_GPU]
        prompt = torch.cat([caps, torch.full((B_PER_GPU, 1), 49152 + 1, dtype=torch.long)], 1).to(dev)
    t_setup = time.time() - t_setup

    def step(max_new=n_new):
        if t2s:
            emb = eng.embed_tokens(prompt)                     # text2svg: no image encoder, no adapter
        else:
            enc = eng.encode_image(images)                     # a2-a5
            emb = eng.prepare_inputs(enc, prompt)              # a6 + a1 / a7: adapter rows and prompt rows written into one buffer
        if NB > 1:     # the reference's default decode (starvector_base.py:231-239): beam search / beam-sample over B x num_beams rows
            new = eng.generate(emb, max_length=S0 + max_new, eos_token_id=-1, pad_token_id=PAD_ID, num_beams=NB,
                               do_sample=bool(args.sample), temperature=1.0, top_p=0.9 if args.sample else 1.0,
                               top_k=int(args.top_k) if args.sample else 0, seed=1)
        else:
            new = eng.generate(emb, max_length=S0 + max_new, eos_token_id=-1,      # EOS disabled:
                               pad_token_id=PAD_ID,                                # fixed-length workload
                               do_sample=is8b, temperature=1.0, top_p=0.95, top_k=50 if is8b else 0,
                               seed=1)       # config 4 samples: top-p 0.95 after HF 4.49's implicit top-k 50
        out = new if t2s else torch.cat([prompt, new], 1)      # starvector_base.py:256 (text2svg returns the new ids, :329-330)
        if world > 1:
            # ONE collective: int32 [B, 1 + width] per rank (column 0 = the row's length); the width is known up front
            out = all_gather_token_streams(out.cpu() if shared else out, PAD_ID, B_PER_GPU * world,
                                           width=(0 if t2s else len(PROMPT_IDS)) + max_new)
        return out, new.shape[1]

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    n_tok = 0
    decode_ms, decode_steps, graph, graph_steps = 0.0, 0.0, True, 0
    for _ in range(args.steps):
        _, n = step()
        n_tok += n
        tm = eng.last_timing()
        decode_ms += tm["decode_ms"]; decode_steps += tm["decode_steps"]; graph = graph and tm["graph"]
        graph_steps = tm.get("graph_steps", 0)
    sync_all()
    dt = time.perf_counter() - t0
    per_rank_tps = None
    if world > 1:
        # every rank's own clock over the same K steps (a straggler is visible in the line); the job's time is the MAX
        mine = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared else dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_tps = [round(n_tok * B_PER_GPU / float(x.item()), 1) for x in every]
        dt = max(float(x.item()) for x in every)
    total_tokens = n_tok * B_PER_GPU * world
    value = total_tokens / dt

    # time-to-first-token: encoder + adapter + prefill + first sampled token, batch 32, p50 over requests
    ttfts = []
    for _ in range(args.ttft_requests):
        torch.cuda.synchronize()
        a = time.perf_counter()
        step(max_new=1)
        torch.cuda.synchronize()
        ttfts.append((time.perf_counter() - a) * 1e3)
    ttft_p50 = statistics.median(ttfts) if ttfts else None
    # ... and where it goes: one pass with a HIP event in front of every launch group (sv_profile_ttft).  The events open small gaps
    # between launches, so the stages sum to a little more than the unprofiled figure above -- both are printed.
    ttft_stages = None
    try:
        tp = eng.profile_ttft(None if t2s else images, prompt, iters=3)
        ttft_stages = {k: round(tp[k], 3) for k in eng.TTFT_STAGES}
        ttft_stages["sum_of_stages"] = round(sum(tp[k] for k in eng.TTFT_STAGES), 3)
        ttft_stages["first_to_last_event"] = round(tp["first_to_last_event_ms"], 3)
        ttft_stages["launch_groups"] = tp["launches"]
        ttft_stages["note"] = ("HIP-event deltas minus the empty event-pair time (%.4f ms), one batch of %d, mean of 3 passes; gemm_remainder_rows = "
                               "the peeled remainder-row launches of the big-M GEMMs of all towers" % (tp["event_pair_overhead_ms"], B_PER_GPU))
    except Exception as ex:                      # a measurement extra: never the reason a bench line is missing
        ttft_stages = {"error": f"{type(ex).__name__}: {ex}"}

    # dominant kernel = skinny weight-streaming GEMM (97 launches / decode step): HIP-event time per step
    # the decode step's real shape, asked of the engine BEFORE the profiling legs run their own (unfused) steps: kernel nodes of the captured
    # step graph and which fused launches were on (sv_debug_step_plan; ADVICE r05: not re-derived from the configuration)
    if NB > 1:
        step(max_new=8)      # the one-token TTFT requests left B rows in the cache: the profiling legs below need all B x num_beams rows live
    plan = eng.step_plan()
    prof = eng.profile_decode_step(ROWS, iters=5)
    sk = prof["skinny_gemm"]
    launches = max(sk["launches_per_step"], 1.0)
    # average launch duration of the dominant kernel: its 97 launches of one step enqueued back to back between
    # one HIP event pair on the engine stream (dispatch to dispatch, the interval rocprofv3 reports per kernel)
    sk_chain_ms = prof.get("skinny_chain_ms_per_step", 0.0) or sk["ms_per_step"]
    sk_exec_ms = sk["ms_per_step"]                 # per-launch event deltas minus the empty event-pair time
    # IN SITU the same launches sit between attention / row-update launches (cold L2, a different predecessor): what the step
    # spends on them is the measured step minus the chain of everything else, also enqueued back to back between one event
    # pair.  rocprofv3's per-kernel average (profiles/) lies between the two figures; `frac` uses the in-situ (larger) one.
    step_ms = decode_ms / max(decode_steps, 1)
    others_ms = prof.get("others_chain_ms_per_step", 0.0)
    # An engine that owns its GPU runs the row update of every layer INSIDE the c_attn launch (rowln_cattn_kernel); the profiling legs time the
    # unfused launches, so the 24 row updates that now belong to the family are taken out of the "others" chain (their event-delta share)
    # (StarVector-8B, bf16, <= 32 rows: the same for the ln_1 row update of its 7-launch layer -- rowln_cattn_kernel<9, true>; ln_2's stays a launch)
    rc_on = plan["rowln_cattn_fused"]
    if rc_on and others_ms > 0:
        ru = prof.get("row_update_ln", {})
        n_ru = max(ru.get("launches_per_step", 0.0), 1.0)
        others_ms = max(others_ms - ru.get("ms_per_step", 0.0) * min(float(cfg.n_layer), n_ru - 1.0) / n_ru, 0.0)
    sk_ms = max(sk_chain_ms, step_ms - others_ms) if others_ms > 0 and step_ms > others_ms else sk_chain_ms
    achieved = W_BYTES_PER_STEP / (sk_ms * 1e-3) / 1e9 if sk_ms > 0 else None
    # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside this process, so the figure is
    # the one a separate `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` pass of this command left in profiles/ -- labelled
    # as such (source + the kernel it was taken on); null when it does not apply to this configuration
    traffic, traffic_source = None, None
    tfile = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tfile) and not is8b and args.weights == "bf16" and not t2s:
        try:
            tj = json.load(open(tfile))
            # the figure belongs to ONE launch family (73 launches per step with the fused MLP launch, 97 without): quoted only for that one
            if int(tj.get("launches_per_step", round(launches))) == int(round(launches)):
                traffic = tj.get("skinny_gemm_bytes_per_launch")
                traffic_source = ("static: profiles/hbm_traffic.json (" + tj.get("measured", "rocprofv3 --pmc pass, separate run") + ")")
            else:
                traffic_source = (f"profiles/hbm_traffic.json was measured on the {tj.get('launches_per_step')}-launch family; this engine runs "
                                  f"{int(round(launches))} launches per step")
        except Exception:
            traffic = None

    # the profiler's own figure for the same family (rocprofv3 --kernel-trace of this command, tools/rocprof_family.py -> profiles/rocprof_family.json):
    # quoted beside the in-situ one, which is a subtraction of chains
    rocprof_us, rocprof_src = None, None
    rfile = os.path.join(ROOT, "profiles", "rocprof_family.json")
    if os.path.exists(rfile) and not is8b and args.weights == "bf16" and not t2s and NB == 1:
        try:
            rj = json.load(open(rfile))
            rocprof_us = rj.get("avg_launch_us")
            rocprof_src = f"static: profiles/rocprof_family.json ({rj.get('measured', '')}; {rj.get('launches')} launches of the family, {rj.get('source')})"
        except Exception:
            rocprof_us = None

    # secondary, MFMA-bound kernels (TTFT path): the four decoder GEMMs of one prefill layer at M = B * S0 rows, live,
    # through the same dispatch the engine uses (128^2 / 256^2 tile kernel + tail kernel)
    from starvector_amd.engine import bench_linear, set_linear_seq_rows, gemm_seq_form
    Mp = B_PER_GPU * S0
    set_linear_seq_rows(S0)          # the prompt pass's own form: rows of S0-row sequences (per-sequence remainder where gemm_seq_form holds)
    D, F = cfg.hidden, cfg.n_inner
    qkv = D + 2 * (D // cfg.n_head) * cfg.n_kv_head
    head_dim = D // cfg.n_head
    gemms = [("c_attn", qkv, D, "none", False), ("c_proj", D, D, "none", True),
             ("c_fc", F, D, "gelu_tanh", False), ("down_proj", D, F, "none", True)]
    pf_us, pf_flop, per_gemm = 0.0, 0.0, {}
    for gname, n_, k_, act_, res_ in gemms:
        us_ = bench_linear(Mp, n_, k_, act=act_, residual=res_, iters=5 if is8b else 10)
        fl_ = 2.0 * Mp * n_ * k_
        pf_us += us_
        pf_flop += fl_
        per_gemm[gname] = {"shape": [Mp, n_, k_], "us": round(us_, 1), "tflops": round(fl_ / us_ / 1e6, 1),
                           "per_sequence_remainder": bool(gemm_seq_form(S0, n_, k_, act_))}
    set_linear_seq_rows(0)
    tf_fc = pf_flop / pf_us / 1e6

    if rank == 0:
        res = {
            "metric": f"SVG tokens/sec (whole job) + p50 time-to-first-token, StarVector-{args.model.upper()} {args.task} batch{B_PER_GPU}/GPU",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.weights == "bf16" else "bf16 activations / fp8-e4m3 decoder weights (not the reference precision)",
            "data": f"synthetic: random-pixel {cfg.image_size}x{cfg.image_size} images (CLIP-normalised), random-init weights N(0,0.02) seed 1234 drawn on the GPU",
            "config": {"workload": (f"StarVector-{args.model.upper()} text2svg (no image encoder), batch {B_PER_GPU}/GPU, {args.weights} decoder weights, "
                                    f"{'top-k 50 + top-p 0.95' if is8b else 'greedy'}, prompt rows {S0} ({CAPTION_TOKENS} caption ids + <svg-start>), "
                                    f"{n_new} new tokens/seq, EOS disabled") if t2s else
                                   (f"StarVector-8B im2svg, batch {B_PER_GPU}/GPU, bf16, top-k 50 + top-p 0.95, 384x384, prompt rows "
                                    f"{S0} (576 visual + {len(PROMPT_IDS)}), {n_new} new tokens/seq, EOS disabled") if is8b else
                                   (f"StarVector-1B im2svg, batch {B_PER_GPU}/GPU, bf16, " +
                                    (f"{('beam-sample (top-k ' + str(int(args.top_k)) + ' + top-p 0.9)' if int(args.top_k) > 0 else 'beam-sample (top-p 0.9, no top-k)') if args.sample else 'beam search'} num_beams {NB} = {ROWS} rows per decode step "
                                     "(the reference's default generate_im2svg call), " if NB > 1 else "greedy, ") +
                                    f"224x224, prompt rows {S0} (257 visual + {len(PROMPT_IDS)}), {n_new} new tokens/seq, EOS disabled"),
                       "global_batch": B_PER_GPU * world, "new_tokens": n_new,
                       "parallelism": f"dp{world}" if world > 1 else "single",
                       **({"dist_backend": ("gloo, ranks SHARE GPUs (functional run of the sharded path on a box with fewer GPUs than "
                                            "ranks; not a scaling number)") if shared else f"nccl (RCCL {rccl_version})",
                           "collectives_per_step": 1, "rccl_version": rccl_version,
                           "tokens_per_s_by_rank": per_rank_tps} if world > 1 else {}),
                       "hipgraph_decode": bool(graph), "decode_steps_per_graph_launch": graph_steps, "exclusive_device": bool(ec.exclusive_device)},
            "tokens_per_s_per_gpu": round(value / world, 1),
            "ttft_p50_ms": round(ttft_p50, 2) if ttft_p50 is not None else None,
            "ttft_breakdown_ms": ttft_stages,
            "decode_us_per_step": round(decode_ms / max(decode_steps, 1) * 1e3, 1),
            "roofline": {"bound": "hbm", "kernel": "decoder weight-streaming GEMMs: " + (
                             "gemm_skinny_mt2_kernel (two row tiles, up to three column tiles per block)" if ROWS > 32 else
                             "rowln_cattn_kernel<9, true> (ln_1 row update + c_attn in one launch) + gemm_skinny_kernel" if (is8b and rc_on) else
                             "gemm_skinny_kernel" if (is8b or args.weights != "bf16") else
                             "rowln_cattn_kernel (row update + c_attn in one launch) + gemm_cols_resid_kernel (attention output projection) + mlp_fused_kernel "
                             "(c_fc and down projection in one launch) + gemm_head_persist_kernel (lm_head: one round of blocks over six column tiles each; its last block also runs the greedy step's bookkeeping)"
                             if ec.exclusive_device else
                             "gemm_skinny_kernel (+ gemm_cols_resid_kernel for the attention output projection)") + f", {int(launches)} launches/step",
                         "achieved": round(achieved, 1) if achieved else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4) if achieved else None, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "algorithmic_bytes_per_launch": round(W_BYTES_PER_STEP / launches),
                         "avg_launch_us": round(sk_ms * 1e3 / launches, 2),
                         "avg_launch_us_source": ("in situ: (decode step - back-to-back chain of the step's other kernels) / launches" +
                                                  (f"; the {cfg.n_layer} row updates that run inside the c_attn launch count with the family" if rc_on else "")),
                         "avg_launch_us_rocprof": rocprof_us, "avg_launch_us_rocprof_source": rocprof_src,
                         "frac_rocprof": round(W_BYTES_PER_STEP / launches / (rocprof_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if rocprof_us else None,
                         "avg_launch_us_gemm_chain": round(sk_chain_ms * 1e3 / launches, 2),
                         "frac_gemm_chain": round(W_BYTES_PER_STEP / (sk_chain_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if sk_chain_ms > 0 else None,
                         "avg_exec_us_event_deltas": round(sk_exec_ms * 1e3 / launches, 2)},
            "roofline_prefill_gemm": {"bound": "mfma",
                                      "kernel": "gemm_bf16_kernel / gemm256_kernel + gemm_tail_kernel / gemm_tailk_kernel for the remainder rows (the 4 decoder GEMMs of a prefill layer, dispatched as the prompt pass dispatches them)",
                                      "achieved": round(tf_fc, 1), "peak": 2500.0, "unit": "TFLOP/s",
                                      "frac": round(tf_fc / 2500.0, 4), "us_per_layer": round(pf_us, 1), "gemms": per_gemm},
            # the whole decode step against the same roof: every byte a step must move (decoder weights once + the KV cache of
            # every sequence at the mean context of the run) over the measured time of a step
            "roofline_whole_step": (lambda kvb, us: {
                "bound": "hbm", "bytes_per_step": int(W_BYTES_PER_STEP + kvb), "us_per_step": round(us, 1),
                "achieved": round((W_BYTES_PER_STEP + kvb) / (us * 1e-6) / 1e9, 1) if us > 0 else None, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round((W_BYTES_PER_STEP + kvb) / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if us > 0 else None,
                # the profiling legs time the UNFUSED launches; the captured step of an engine that owns its GPU runs the layer's row update inside
                # the c_attn launch (n_layer launches less) and a plain greedy step selects inside the lm_head launch (no argmax launch)
                "launches_per_step": plan["graph_kernel_nodes"] or (
                    int(sum(v["launches_per_step"] for v in prof.values() if isinstance(v, dict))) + 2
                    - (cfg.n_layer if rc_on else 0) - (1 if plan["greedy_in_lm_head"] else 0)),
                "launches_per_step_source": "kernel nodes of the captured decode-step graph (sv_debug_step_plan)" if plan["graph_kernel_nodes"]
                                            else "profiling legs' launch counts, corrected by the engine's fused-launch decisions",
                "fused_launches": {k: plan[k] for k in ("rowln_cattn_fused", "greedy_in_lm_head", "mlp_fused")}})(
                    ROWS * (S0 + n_new / 2.0) * cfg.n_layer * 2 * cfg.n_kv_head * head_dim * 2,
                    decode_ms / max(decode_steps, 1) * 1e3),
            "decode_step_profile_ms": {k: round(v["ms_per_step"], 4) for k, v in prof.items() if isinstance(v, dict)},
            "setup_s": round(t_setup, 1),
        }
        if keep_cpu:
            res["cpu_baseline"] = cpu_baseline()
        print(json.dumps(res), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
