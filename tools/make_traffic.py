"""profiles/r02_pmc_{fetch,write}.csv (summarize_pmc.py tables of separate --pmc FETCH_SIZE / WRITE_SIZE passes over
`bench.py --steps 2 --warmup 1 --no-overlap --no-cpu-baseline --no-workloads`) -> profiles/traffic.json, the per-frame HBM-side
bytes of the kernels bench.py reports.  usage: python tools/make_traffic.py <fetch.csv> <write.csv> <frames_per_launch>"""
import csv
import json
import os
import sys

KEYS = {   # bench.py's kernel key -> substring of the rocprof kernel name
    "knn_query_multi<16, true>": "knn_query_multi<16, true>",
    "lfa_attn_mfma16<1>": "lfa_attn_mfma16<1,",
    "lfa_attn_wave<64,2>": "lfa_attn_wave<64, 2,",
}


def table(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[(r["Kernel"], r["Counter"])] = float(r["MeanPerDispatch"])
    return out


def main():
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
