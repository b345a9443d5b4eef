"""diag (round 5): what keeps a poisoned request's NaN alive across calls?  tiny engine, classic entry points."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from safetensors.torch import load_file
from oracle import starvector_oracle as O
from tests.gpu_util import bf, build_engine, dev
import starvector_amd as sva

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
g = load_file(os.path.join(root, "tests", "golden", "tiny_b3.safetensors"))
cfg = O.OracleConfig.tiny()
w = O.apply_fixture_weights(O.make_weights(cfg, seed=int(g["meta"][0])), cfg, g)

def fresh():
    eng = build_engine(cfg, w, max_batch=8, max_seq_len=120)
    img = bf(g["image"])
    prompt = torch.tensor([[7, 11]] * 3, device=dev())
    emb = torch.cat([eng.adapter(eng.encode_image(img)), eng.embed_tokens(prompt)], 1)
    bad = emb.clone(); bad[1, 2, 5] = float("nan")
    return eng, emb.contiguous(), bad.contiguous()

def walk(eng, emb, n=6, tag=""):
    lg = eng.prefill(emb)
    out = [int(torch.isnan(lg.float()).any(-1).sum())]
    tok = torch.nan_to_num(lg.float()).argmax(-1)
    for _ in range(n):
        try:
            lg = eng.decode_step(tok)
        except Exception as e:
            out.append("ERR:" + str(e)[:60]); break
        out.append(int(torch.isnan(lg.float()).any(-1).sum()))
        tok = torch.nan_to_num(lg.float()).argmax(-1)
    print(tag, "rows with NaN logits per step:", out, flush=True)

eng, emb, bad = fresh()
walk(eng, emb, tag="[clean engine]      ")
walk(eng, bad, tag="[poisoned prompt]   ")
walk(eng, emb, tag="[clean, right after]")
walk(eng, emb, tag="[clean, once more]  ")
x = torch.empty(1 << 30, dtype=torch.uint8, device=dev()); x.fill_(1); y = x.clone(); torch.cuda.synchronize(); del x, y
walk(eng, emb, tag="[clean, after 2 GiB of cache traffic]")
eng.close()

eng, emb, bad = fresh()
kw = dict(max_length=emb.shape[1] + 24, eos_token_id=-1)
try:
    eng.generate(bad, **kw); print("poisoned generate: NO error")
except Exception as e:
    print("poisoned generate:", str(e)[:90])
for i in range(3):
    try:
        t = eng.generate(emb, **kw).cpu(); print(f"clean generate #{i}: ok, equals golden: {torch.equal(t, g['tokens'])}")
    except Exception as e:
        print(f"clean generate #{i}:", str(e)[:90])
walk(eng, emb, tag="[clean walk after generate path]")
eng.close()
