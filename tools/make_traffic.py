"""profiles/r02_pmc_{fetch,write}.csv (summarize_pmc.py tables of separate --pmc FETCH_SIZE / WRITE_SIZE passes over
`bench.py --steps 2 --warmup 1 --no-overlap --no-cpu-baseline --no-workloads`) -> profiles/traffic.json, the per-frame HBM-side
bytes of the kernels bench.py reports.  usage: python tools/make_traffic.py <fetch.csv> <write.csv> <frames_per_launch>"""
import csv
import json
import os
import sys

KEYS = {   # bench.py's kernel key -> substring of the rocprof kernel name
    "knn_query_multi<16, true>": "knn_query_multi<16, true",
    "lfa_attn_mfma16<1>": "lfa_attn_mfma16<1,",
    "lfa_attn_wave<64,2>": "lfa_attn_wave<64, 2,",
}


def table(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[(r["Kernel"], r["Counter"])] = float(r["MeanPerDispatch"])
    return out


OPS = {   # bench_models.py's roofline key -> (kernel-name substrings that make up the op, units per launch); tables from
          # `rocprofv3 --pmc ... -- python tools/roofline_ops.py kp|pp` (only that op runs there)
    "kpconv_block_32_32": (("kp_agg_gemm32",), 64),
    "pp_conv3x3_64": (("gemm_tile",), 16),
}


def add_op(key, fetch_csv, write_csv):
    """python tools/make_traffic.py --op <key> <fetch.csv> <write.csv>: add one op's entry to profiles/traffic.json"""
    f, w = table(fetch_csv), table(write_csv)
    subs, units = OPS[key]
    fetch_kib = sum(v for (k, c), v in f.items() if c == "FETCH_SIZE" and any(s_ in k for s_ in subs))
    write_kib = sum(v for (k, c), v in w.items() if c == "WRITE_SIZE" and any(s_ in k for s_ in subs))
    names = sorted({k for (k, c) in f if c == "FETCH_SIZE" and any(s_ in k for s_ in subs)})
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    out = json.load(open(path))
    out["kernels"][key] = {"rocprof_names": names, "fetch_size_kib_per_launch": fetch_kib, "write_size_kib_per_launch": write_kib,
                           "bytes_per_launch_per_frame": (2.0 * fetch_kib + write_kib) * 1024.0 / units,
                           "source": "%s + %s (tools/roofline_ops.py: the op alone, %d units per launch; sum over its kernels)"
                                     % (os.path.basename(fetch_csv), os.path.basename(write_csv), units)}
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out["kernels"][key], indent=1))


def main():
    if sys.argv[1] == "--op":
        return add_op(sys.argv[2], sys.argv[3], sys.argv[4])
    f, w, frames = table(sys.argv[1]), table(sys.argv[2]), int(sys.argv[3])
    kernels = {}
    for key, sub in KEYS.items():
        fk = [k for k in f if sub in k[0] and k[1] == "FETCH_SIZE"]
        wk = [k for k in w if sub in k[0] and k[1] == "WRITE_SIZE"]
        if not fk or not wk:
            continue
        fetch_kib, write_kib = f[fk[0]], w[wk[0]]
        kernels[key] = {"rocprof_name": fk[0][0], "fetch_size_kib_per_launch": fetch_kib, "write_size_kib_per_launch": write_kib,
                        "bytes_per_launch_per_frame": (2.0 * fetch_kib + write_kib) * 1024.0 / frames}
    out = {"source": "%s + %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bench.py --steps 2 --warmup 1 "
                     "--no-overlap --no-cpu-baseline --no-workloads, %d frames per launch)"
                     % (os.path.basename(sys.argv[1]), os.path.basename(sys.argv[2]), frames),
           "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md §HBM); bytes = (2 * FETCH_SIZE "
                         "+ WRITE_SIZE) * 1024; WRITE_SIZE uncalibrated",
           "kernels": kernels}
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
