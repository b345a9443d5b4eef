#!/usr/bin/env bash
# time to first token of another (built) checkout against this tree, same box, interleaved:  bash tools/diag/ttft_tree_ab.sh DIR [reps]
DIR="${1:?other checkout}"; REPS="${2:-3}"
for r in $(seq "$REPS"); do
  for t in "$DIR" .; do echo "== tree $t (rep $r)"; ( cd "$t" && timeout 600 python tools/ttft_ab.py --reps 15 0 2>/dev/null | tail -1 | cut -c1-160 ); done
done
