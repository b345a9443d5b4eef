"""The polled activation buffer (`xp_a`) of the fused row-update + c_attn launch is consumed by blocks that POLL it while other blocks of the
same launch -- and the launches before it -- write it.  That is only correct when every access bypasses the non-coherent caches: write-through
(`sc1`) stores and L1-bypassing (`sc1`) loads, i.e. the `raw_buffer_store_b128 / raw_buffer_load_b128(..., aux = 16)` builtins.  Round 5's NaN
logits (a stale line of the buffer in one XCD's L2 after a plain store) were found by the bench, not by a test; this file turns the rule into
one: it reads the SOURCES and fails when

  * the host hands `e->xp_a` to anything but the sanctioned arguments on an engine with the fused launch, or
  * a kernel that receives the buffer dereferences it other than through a buffer resource with aux = 16.

Source-level on purpose: it runs in the CPU suite, before anything is launched."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "star-vector_amd", "csrc")


def _code(path):
    """file text with // comments and /* */ blocks removed (line structure kept)"""
    s = open(os.path.join(CSRC, path)).read()
    s = re.sub(r"/\*.*?\*/", lambda m: "\n" * m.group(0).count("\n"), s, flags=re.S)
    return "\n".join(line.split("//")[0] for line in s.split("\n"))


def _body(src, header_regex):
    """text of the function whose header matches `header_regex`: from the header to its closing brace"""
    m = re.search(header_regex, src)
    assert m, f"function not found: {header_regex}"
    i = src.index("{", m.end())
    depth, j = 0, i
    while True:
        depth += {"{": 1, "}": -1}.get(src[j], 0)
        if depth == 0:
            return src[m.start():j + 1]
        j += 1


def test_host_hands_the_polled_buffer_only_to_sanctioned_arguments():
    src = _code("engine_forward.hip")
    sanctioned = [
        r"a\.poison = e->xp_a",                       # lm_head launch: arms the buffer for the next step's layer 0 (write-through stores)
        r"ad\.poison2 = e->xp_a",                     # attention launch of layer i: arms it for layer i + 1 (write-through stores)
        r"ca\.poison2 = e->xp_a",                     # ... or, where the attention grid is too small (batches below 10 rows), the output projection's launch
        r"a\.xp = e->xp_a",                           # the fused launch's GEMM role polls it (sc1 loads); its row role writes ru.xp_out = the same buffer
        r"\? e->xp_f : e->xp_a",                      # every other producer / consumer takes xp_f on an engine with the fused launch
        r"xp != e->xp_a",
    ]
    has = lambda ln: re.search(r"e->xp_a\b", ln) is not None                 # (not xp_attn)
    uses = [ln for ln in src.split("\n") if has(ln)]
    assert len(uses) >= 5, "engine_forward.hip no longer mentions e->xp_a: update this test with the new name"
    for ln in uses:
        rest = ln
        for pat in sanctioned:
            rest = re.sub(pat, "", rest)
        assert not has(rest), f"unsanctioned use of the polled buffer: {ln.strip()}"
    # the selector lines really are guarded by rc_enabled (xp_a only where the fused launch cannot be on)
    for ln in uses:
        if "? e->xp_f : e->xp_a" in ln:
            assert "rc_enabled(e)" in ln, f"xp_a chosen without asking rc_enabled: {ln.strip()}"
    # the fused launch gets the buffer only inside the `if (rc)` block
    blk = src[src.index("if (rc) {"):]
    blk = blk[:blk.index("if (!rc_done)")]
    assert "a.xp = e->xp_a" in blk


def _only_through_resource(body, name, what):
    """every mention of `name` in `body` is a null test, a byte count, or the pointer argument of make_buffer_rsrc"""
    for ln in body.split("\n"):
        if not re.search(rf"\b{name}\b(?!_bytes)", ln):
            continue
        rest = re.sub(rf"__builtin_amdgcn_make_buffer_rsrc\(\s*\w+\.{name}\s*,", "", ln)
        rest = re.sub(rf"\w+\.{name}\s*&&", "", rest)
        rest = re.sub(rf"if \(\s*\w+\.{name}\s*\)", "", rest)
        rest = re.sub(rf"&&\s*\w+\.{name}\b", "", rest)
        assert not re.search(rf"\b{name}\b(?!_bytes)", rest), f"{what}: `{name}` used outside a buffer resource: {ln.strip()}"


def _aux16_everywhere(body, what, min_calls):
    calls = re.findall(r"__builtin_amdgcn_raw_buffer_(?:store|load)_b(?:128|32)\(([^;]*)\);", body)
    assert len(calls) >= min_calls, f"{what}: expected >= {min_calls} buffer accesses, found {len(calls)}"
    for c in calls:
        assert re.search(r",\s*16\s*\)?\s*$", c.strip()), f"{what}: buffer access without aux = 16 (sc1): {c.strip()}"


def test_attention_launch_arms_the_buffer_with_write_through_stores():
    body = _body(_code("attention.hip"), r"void attn_decode_kernel\(")
    _only_through_resource(body, "poison2", "attn_decode_kernel")
    store = _body(body, r"auto poison2_store = ")
    _aux16_everywhere(store, "attn_decode_kernel poison2_store", 1)


def test_output_projection_launch_arms_the_buffer_with_write_through_stores():
    body = _body(_code("decode_cols.hip"), r"void gemm_cols_resid_kernel\(")
    _only_through_resource(body, "poison2", "gemm_cols_resid_kernel")
    i = body.index("pl.poison2)")
    _aux16_everywhere(body[i:i + 900], "gemm_cols_resid_kernel poison2 store", 1)


def test_lm_head_launch_arms_the_buffer_with_write_through_stores():
    body = _body(_code("gemm.hip"), r"void gemm_skinny_kernel\(")
    _only_through_resource(body, "poison", "gemm_skinny_kernel")
    i = body.index("p.poison)")
    _aux16_everywhere(body[i:i + 900], "gemm_skinny_kernel poison store", 1)


def test_fused_row_update_launch_touches_the_buffer_write_through_only():
    src = _code("rowops.hip")
    for header, what, n in ((r"void rowln_cattn_kernel\(", "rowln_cattn_kernel", 3), (r"void rowln_wide_role\(", "rowln_wide_role", 1)):
        body = _body(src, header)
        _only_through_resource(body, "xp_out", what)
        # inside these bodies every buffer access is on the polled buffer or on partial slabs: all of them carry aux = 16 or are plain slab stores
        for c in re.findall(r"__builtin_amdgcn_raw_buffer_(?:store|load)_b128\(([^;]*)\);", body):
            assert re.search(r",\s*16\s*\)?\s*$", c.strip()), f"{what}: buffer access without aux = 16: {c.strip()}"
        assert len(re.findall(r"__builtin_amdgcn_raw_buffer_(?:store|load)_b128\(", body)) >= n
    # the plain (two-launch) row update keeps its plain stores -- and must therefore never be given xp_a on a fused engine (host test above)
    plain = _body(src, r"void row_update_ln_kernel\(")
    assert "p.xp_out + xp_index" in plain
