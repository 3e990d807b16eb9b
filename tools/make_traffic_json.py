#!/usr/bin/env python
"""Build profiles/traffic_<cfg>.json (HBM bytes per launch of the dominant kernels) from the two
pmc_traffic.py summaries of separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes.

usage: make_traffic_json.py FETCH.csv WRITE.csv out.json [layernorm rows per launch = 262144]

FETCH_SIZE is doubled: on gfx950 it counts half of the bytes of 16 B/lane streams (see
MI355X_MICROARCH.md; calibrated here on the C = 320 LayerNorm kernel, which reads 327,680 KB per launch at the
128^2 level and reports 163,9xx KB).  Both counters are in KB."""
import csv
import json
import sys


def load(path):
    rows = {}
    with open(path) as fh:
        next(fh)                       # "counter,NAME"
        for r in csv.DictReader(fh):
            rows[r["kernel"]] = (int(r["dispatches"]), float(r["sum"]))
    return rows


def family(rows, prefix):
    n = sum(d for k, (d, s) in rows.items() if k.startswith(prefix))
    s = sum(s for k, (d, s) in rows.items() if k.startswith(prefix))
    return n, s


def main():
    fetch, write, out = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    res = {}
    for fam in ("gemm_bf16_kernel", "attn_d64_kernel", "ffn2_geglu_c320_kernel"):
        nf, sf = family(fetch, fam)
        nw, sw = family(write, fam)
        n = max(nf, nw)
        res[fam] = {
            "bytes_per_launch": int((2.0 * sf + sw) * 1024 / n),
            "unit": "B",
            "launches_profiled": n,
            "fetch_KB_x2_corrected": int(2.0 * sf),
            "write_KB": int(sw),
            "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, kernel trace only) on bench.py --steps 1 --warmup 1 "
                      "--no-cpu-baseline --no-profile --no-legs with HI3D_STEP_GRAPH=0: 4 eager sampler steps and nothing else, so "
                      "every profiled launch belongs to a step; FETCH_SIZE doubled (gfx950 half-count for 16 B/lane streams, "
                      "calibrated on layernorm)",
        }
    # calibration: the C = 320 LayerNorm reads ln_rows x 320 bf16 per launch at the 128^2 level -- ln_rows = 262144 in the default
    # two-stream step (half-batch launches; round 4's json carried the full-batch 327,680 KB here by mistake), 524288 with
    # HI3D_TWO_STREAM=0; optional 4th argument
    ln_rows = int(sys.argv[4]) if len(sys.argv) > 4 else 262144
    cal = [v for k, v in fetch.items() if k.startswith("layernorm_packed_kernel<8>") or k.startswith("layernorm_kernel<1>")]
    if cal:
        rep = sum(c[1] for c in cal) / sum(c[0] for c in cal)
        res["calibration"] = {"layernorm_C320_fetch_KB_per_launch_reported": rep, "KB_actually_read_per_launch": ln_rows * 640 // 1024,
                              "x2_corrected_over_actual": round(2.0 * rep / (ln_rows * 640 / 1024), 4)}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
