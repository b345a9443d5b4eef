#!/usr/bin/env python3
"""profiles/hbm_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; tools/pmc_summary.py output), for the
decoder's weight-streaming GEMM family of the decode step (gemm_skinny_kernel<*, false|true> + gemm_cols_resid_kernel + mlp_fused_kernel +
rowln_cattn_kernel, the round-5 launch that carries the c_attn projection behind its row update):
    python tools/hbm_traffic.py <pmc_FETCH_SIZE.json> <pmc_WRITE_SIZE.json> <algorithmic bytes per launch> "<where measured>"
Rule (MI355X_MICROARCH.md, HBM section): both counters are KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced
streaming read (16 B per lane) -> doubled; WRITE_SIZE taken as is; separate passes, --kernel-trace only.  bench.py reads the result."""
import json
import sys

FAMILY = ("gemm_skinny_kernel", "gemm_head_persist_kernel", "gemm_cols_resid_kernel", "mlp_fused_kernel", "rowln_cattn_kernel")


def family(path, counter):
    rows = {k: v[counter] for k, v in json.load(open(path)).items() if any(f in k for f in FAMILY) and counter in v}
    calls = sum(r["calls"] for r in rows.values())
    return rows, calls, sum(r["calls"] * r["avg"] for r in rows.values()) / max(calls, 1)


def main():
    fetch_json, write_json, alg, where = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fr, fcalls, fkib = family(fetch_json, "FETCH_SIZE")
    wr, wcalls, wkib = family(write_json, "WRITE_SIZE")
    fetch_b, write_b = int(fkib * 1024 * 2), int(wkib * 1024)
    out = {
        "skinny_gemm_bytes_per_launch": fetch_b + write_b,
        "derivation": {
            "kernels": {k: {"launches": v["calls"], "FETCH_SIZE_avg_KiB": round(v["avg"], 2),
                            "WRITE_SIZE_avg_KiB": round(wr.get(k, {}).get("avg", 0.0), 2)} for k, v in sorted(fr.items())},
            "launches_sampled": fcalls,
            "FETCH_SIZE_avg_KiB": fkib, "WRITE_SIZE_avg_KiB": wkib,
            "rule": "MI355X_MICROARCH.md section HBM: FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a "
                    "wide coalesced streaming read (16 B/lane) -> doubled; WRITE_SIZE uncalibrated, taken as is; collected in separate "
                    "--pmc passes with --kernel-trace only; launch-weighted mean over the family",
            "fetch_bytes_corrected": fetch_b, "write_bytes": write_b,
            "algorithmic_bytes_per_launch": alg,
            "traffic_over_algorithmic": round((fetch_b + write_b) / alg, 3),
            "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python bench.py --steps 1 "
                       "--warmup 0 --new-tokens 64 --no-cpu-baseline --ttft-requests 1",
        },
        "measured": where,
    }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
