#!/usr/bin/env bash
# ONE same-box A/B driver for the GPU box (replaces the fourteen one-off r05_*.sh / rc_*.sh / r06_*.sh scripts of rounds 5-6).
# Every arm runs in its own process on the SAME box, arms interleaved `reps` times; a bench.py JSON line is summarised, other output is passed
# through for the lines that carry a number (us_per_step, ttft_ms, TF, passed / failed).  From the repo root:
#   bash tools/diag/bench_ab.sh env   VAR "v1 v2 ..." [reps] -- <command ...>     the command under VAR=v1, VAR=v2, ...   (SV_EXP masks, SV_RC_DELAY, SV_TAIL_RING ...)
#   bash tools/diag/bench_ab.sh tree  DIR [reps] -- <bench.py arguments>          another checkout (DIR, built) against this one, python bench.py <arguments>
#   bash tools/diag/bench_ab.sh build "FLAGS_A" "FLAGS_B" [reps] -- <command ...> two builds of the library (SV_HIPCC_FLAGS), rebuilt on the box before each arm
#   bash tools/diag/bench_ab.sh swap  FILE ALT [reps] -- <command ...>            the tree's FILE against the file ALT (rebuilt before each arm): a source-level A/B
# Examples (round 6):  env SV_EXP "131072 0 262144 524288" -- python bench.py --model 8b --weights fp8 --task text2svg --new-tokens 256 --steps 2 --no-cpu-baseline
#                      swap star-vector_amd/csrc/attention.hip /tmp/attention_before.hip 2 -- python tools/ab_exp.py --new-tokens 1024 --reps 2 0 0
set -u
mode="${1:?mode: env | tree | build | swap}"; shift
summ() {   # stdin: a command's output
  python3 -c '
import json, sys
for ln in sys.stdin:
    ln = ln.strip()
    if ln.startswith("{\"metric\""):
        d = json.loads(ln)
        print("   value %.1f %s | ttft_p50 %s ms | decode %s us/step | gemm family frac %s | whole step frac %s" % (
            d["value"], d["unit"], d.get("ttft_p50_ms"), d.get("decode_us_per_step"), d["roofline"].get("frac"), d.get("roofline_whole_step", {}).get("frac")))
    elif any(k in ln for k in ("us_per_step", "ttft_ms", " TF", "passed", "failed", "Error")):
        print("   " + ln[:260])
'
}
rebuild() { env "$@" python star-vector_amd/build.py > /dev/null 2>&1 || echo "   BUILD FAILED"; }
case "$mode" in
  env)
    var="$1"; vals="$2"; shift 2; reps=1; [ "${1:-}" != "--" ] && { reps="$1"; shift; }; shift
    for r in $(seq "$reps"); do for v in $vals; do echo "== $var=$v (rep $r)"; env "$var=$v" timeout 900 "$@" 2>/dev/null | summ; done; done ;;
  tree)
    dir="$1"; shift; reps=1; [ "${1:-}" != "--" ] && { reps="$1"; shift; }; shift
    for r in $(seq "$reps"); do for t in "$dir" .; do echo "== tree $t (rep $r)"; ( cd "$t" && timeout 900 python bench.py "$@" 2>/dev/null ) | summ; done; done ;;
  build)
    fa="$1"; fb="$2"; shift 2; reps=1; [ "${1:-}" != "--" ] && { reps="$1"; shift; }; shift
    for r in $(seq "$reps"); do for f in "$fa" "$fb"; do echo "== SV_HIPCC_FLAGS='$f' (rep $r)"; rebuild SV_HIPCC_FLAGS="$f"; timeout 900 "$@" 2>/dev/null | summ; done; done
    rebuild SV_HIPCC_FLAGS= ;;
  swap)
    file="$1"; alt="$2"; shift 2; reps=1; [ "${1:-}" != "--" ] && { reps="$1"; shift; }; shift
    cp "$file" /tmp/_bench_ab_tree_version
    for r in $(seq "$reps"); do
      echo "== tree version of $file (rep $r)"; cp /tmp/_bench_ab_tree_version "$file"; rebuild X=1; timeout 900 "$@" 2>/dev/null | summ
      echo "== $alt (rep $r)"; cp "$alt" "$file"; rebuild X=1; timeout 900 "$@" 2>/dev/null | summ
    done
    cp /tmp/_bench_ab_tree_version "$file"; rebuild X=1 ;;
  *) echo "unknown mode $mode"; exit 2 ;;
esac
