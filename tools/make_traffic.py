#!/usr/bin/env python3
"""profiles/<tag>_pmc.json -> profiles/traffic.json (the HBM bytes per launch that bench.py reports as roofline.traffic).

FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts the 128-byte requests of 16 B/lane streaming reads as
64 bytes, so it is doubled.  Usage: tools/make_traffic.py profiles/r01_pmc.json profiles/traffic.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_stamp  # noqa: E402  (sha256 of the kernel sources the counters were taken on)

KEYS = {  # pmc.json prefix -> workload key of bench.py (workload_name)
    "c3": "B64_H32_KV32_D128_S32768_C2048_k32_f16",
    "c5": "B64_H32_KV8_D128_S131072_C2048_k64_f16",
    "c4": "B64_H32_KV32_D128_S65536_C2048_k32_int8",
}
METHOD = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum, one counter set per pass "
          "(tools/profile_round.sh), mean over the attn_kernel dispatches; FETCH_SIZE doubled (gfx950 counts the "
          "128-B requests of 16 B/lane streaming reads as 64 B)")


def attn(entries, counter):
    for e in entries:
        if "attn_kernel" in e["kernel"] and e["counter"] == counter:
            return e
    return None


def main(src, dst):
    d = json.load(open(src))
    out = {}
    stamp = kernel_source_stamp()
    for pre, key in KEYS.items():
        f = attn(d.get(pre + "_fetch", []), "FETCH_SIZE")
        if f is None:
            continue
        raw = {"FETCH_SIZE_KiB": f["mean"]}
        fetch = f["mean"] * 1024 * 2
        write = 0.0
        w = attn(d.get(pre + "_write", []), "WRITE_SIZE")
        if w is not None:
            raw["WRITE_SIZE_KiB"] = w["mean"]
            write = w["mean"] * 1024
        for c in ("TCC_HIT_sum", "TCC_MISS_sum"):
            e = attn(d.get(pre + "_l2", []), c)
            if e is not None:
                raw[c] = e["mean"]
        out[key] = {"fetch_bytes_corrected": fetch, "raw": raw,
                    "kernel_mean_duration_ns_under_pmc": f["mean_duration_ns"], "write_bytes": write,
                    "hbm_bytes_per_launch": fetch + write, "method": METHOD,
                    "kernel_source_stamp": stamp}
    json.dump(out, open(dst, "w"), indent=1)
    for k, v in out.items():
        print(k, "%.3f GB" % (v["hbm_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
