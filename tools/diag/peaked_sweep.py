"""diag (round 5): how peaked can the attention of the long-context e2e case be before bf16 rounding differences (not bugs) dominate?"""
import dataclasses, gc, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import starvector_oracle as O
from tests.gpu_util import build_engine, dev
from tests.test_gpu_e2e import LOGIT_TOL
from tests.test_gpu_long_context import _oracle_long, _page_swap_sensitivity, _synthetic_prompt

B, S0, n_new = 4, 7780, 4
for gain in [float(x) for x in sys.argv[1:]] or [1.0, 2.0, 3.0, 4.0]:
    cfg = dataclasses.replace(O.OracleConfig(), eos_token_id=-1)
    w = O.make_weights(cfg, seed=1234)
    D = cfg.hidden
    for i in range(cfg.n_layer):
        p = f"{O.P_DEC}h.{i}.attn.c_attn."
        w[p + "weight"][:D] = (w[p + "weight"][:D].float() * gain).to(torch.bfloat16).to(w[p + "weight"].dtype)
        w[p + "bias"][:D] = (w[p + "bias"][:D].float() * gain).to(torch.bfloat16).to(w[p + "bias"].dtype)
    eng = build_engine(cfg, w, max_batch=B, max_seq_len=7808)
    w_dev = {k: v.to(dev()) for k, v in w.items() if "image_encoder" not in k and "image_projection" not in k}
    del w; gc.collect()
    emb = _synthetic_prompt(B, S0, cfg.hidden, 8780)
    o_toks, o_lg, cache = _oracle_long(w_dev, cfg, emb, n_new, 2)
    scale = float(o_lg.abs().max())
    errs = []
    for t in range(n_new):
        lg = (eng.prefill(emb) if t == 0 else eng.decode_step(o_toks[:, t - 1].contiguous())).float()
        errs.append(float((lg - o_lg[:, t]).abs().max()) / scale)
    with torch.no_grad():
        ref_next, _ = O.decoder_decode_step(w_dev, cfg, o_toks[:, -1], cache, "bf16")
        # the oracle against ITSELF in fp32 mode: how much of the error is bf16 rounding of a chaotic function
        lg32, _ = O.decoder_prefill(w_dev, cfg, emb[:2].float(), "fp32")
    tol_abs = LOGIT_TOL * scale
    last_page = (S0 + n_new - 2) // 64
    sens = [_page_swap_sensitivity(w_dev, cfg, cache, o_toks[:, -1], ref_next.float(), pg, 0, tol_abs) for pg in (last_page - 1, 61, 7)]
    o32 = float((lg32.float() - o_lg[:2, 0]).abs().max()) / scale
    print(f"gain {gain:g}: engine err / scale per step {[f'{e:.2e}' for e in errs]} (tol {LOGIT_TOL:.1e}); oracle bf16 vs fp32 {o32:.2e}; "
          f"page-swap sensitivity {[f'{x:.1f}' for x in sens]} x tol; scale {scale:.2f}", flush=True)
    eng.close(); del w_dev, cache; gc.collect(); torch.cuda.empty_cache()
