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
    "lfa_attn_wave_b3<64,2>": "lfa_attn_wave_b3<64, 2,",
}


def table(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[(r["Kernel"], r["Counter"])] = float(r["MeanPerDispatch"])
    return out


OPS = {   # bench_models.py's roofline key -> (kernel-name substrings that make up the op, units per launch); tables from
          # `rocprofv3 --pmc ... -- python tools/roofline_ops.py kp|pp` (only that op runs there)
    "kpconv_block_32_32": (("kp_agg_gemm32",), 64),
    "pp_conv3x3_64": (("gemm_tile", "conv3x3s1_bf3"), 16),      # (round 5: the window-staged bf16x3 kernel on the default path)
}
PRIMS = {  # HBM-bound primitives (tools/roofline_ops.py radius|subsample|voxelize|pillars): EVERY kernel the op launches counts
           # -- grid build, scans, sorts, fills, torch's own helper kernels -- except the names excluded here (the H2D copies of the
           # inputs; for `pillars` the one model forward that produced the call's arguments is excluded by listing the op's kernels)
    "kp_radius_dense": (None, ("copyBuffer",), 64),
    "kp_subsample": (None, ("copyBuffer",), 64),
    "pp_voxelize": (None, ("copyBuffer", "CatArray"), 8),       # (torch.cat of the sweeps happens before the op, once)
    "pp_pillar_features": (("pillar_pfn", "pfn_next", "fillBufferAligned", "grid_zero"), (), 8),
}


def counts(path):
    out = {}
    for r in csv.DictReader(open(path)):
        out[(r["Kernel"], r["Counter"])] = (int(r["Dispatches"]), float(r["MeanPerDispatch"]))
    return out


def add_prim(key, fetch_csv, write_csv, launches):
    """python tools/make_traffic.py --prim <key> <fetch.csv> <write.csv> <launches>: the op ran `launches` times ALONE in the
    profiled process; its traffic per launch = sum over its kernels of dispatches x mean / launches"""
    inc, exc, units = PRIMS[key]
    tot, names = {}, set()
    for ctr, path in (("FETCH_SIZE", fetch_csv), ("WRITE_SIZE", write_csv)):
        t = 0.0
        for (k, c), (n, mean) in counts(path).items():
            if c != ctr or any(x in k for x in exc) or (inc is not None and not any(x in k for x in inc)):
                continue
            t += n * mean
            names.add(k.split("(")[0].replace("void ", ""))
        tot[ctr] = t / launches
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    out = json.load(open(path))
    out["kernels"][key] = {"rocprof_names": sorted(names), "fetch_size_kib_per_launch": tot["FETCH_SIZE"],
                           "write_size_kib_per_launch": tot["WRITE_SIZE"],
                           "bytes_per_launch_per_frame": (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0 / units,
                           "source": "%s + %s (tools/roofline_ops.py: the op alone, %d launches of %d units; sum over ALL its kernels)"
                                     % (os.path.basename(fetch_csv), os.path.basename(write_csv), launches, units)}
    json.dump(out, open(path, "w"), indent=1)
    print(key, json.dumps({k: v for k, v in out["kernels"][key].items() if k != "rocprof_names"}))


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


def add_sq(key, sq_csv):
    """python tools/make_traffic.py --sq <kernel key> <sq.csv>: the kernel's VALU-issue roofline from an SQ counter pass
    (SQ_ACTIVE_INST_VALU counts quad-cycles: x 4 = SIMD-cycles the vector ALU was occupied).  Two denominators: the launch's mean
    duration in that pass x 2.4 GHz x 1024 SIMDs (the most cycles the chip could have offered: a LOWER bound of the utilisation,
    the clock under load is 1.9-2.1 GHz), and SQ_BUSY_CYCLES / 32 shader engines x 1024 SIMDs (the cycles it did offer)."""
    t = counts(sq_csv)
    sub = KEYS.get(key, key)
    row = lambda ctr: next((v for (k, c), v in t.items() if sub in k and c == ctr), None)
    act, ins, busy, dur = row("SQ_ACTIVE_INST_VALU"), row("SQ_INSTS_VALU"), row("SQ_BUSY_CYCLES"), row("DURATION_NS")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
    out = json.load(open(path))
    e = {"source": os.path.basename(sq_csv), "insts_valu_per_launch": ins and ins[1], "active_inst_valu_quadcycles": act and act[1],
         "busy_cycles_sum_over_32_se": busy and busy[1], "duration_ns_in_pass": dur and dur[1]}
    if act and dur:
        e["valu_frac_at_2p4ghz"] = 4.0 * act[1] / (dur[1] * 2.4 * 1024)
    if act and busy:
        e["valu_frac_of_busy_cycles"] = 4.0 * act[1] / (busy[1] / 32.0 * 1024)
    out.setdefault("sq", {})[key] = e
    json.dump(out, open(path, "w"), indent=1)
    print(key, json.dumps(e))


def main():
    if sys.argv[1] == "--sq":
        return add_sq(sys.argv[2], sys.argv[3])
    if sys.argv[1] == "--op":
        return add_op(sys.argv[2], sys.argv[3], sys.argv[4])
    if sys.argv[1] == "--prim":
        return add_prim(sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]))
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
    try:       # the entries of the ops / primitives profiled alone survive a rebuild of the step's table (they are re-added after)
        prev = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")))["kernels"]
        for k, v in prev.items():
            if k not in kernels and k not in KEYS:
                kernels[k] = v
    except Exception:
        pass
    sq_prev = {}
    try:
        sq_prev = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"))).get("sq", {})
    except Exception:
        pass
    out = {"sq": sq_prev, "source": "%s + %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bench.py --steps 2 --warmup 1 "
                     "--no-overlap --no-cpu-baseline --no-workloads, %d frames per launch)"
                     % (os.path.basename(sys.argv[1]), os.path.basename(sys.argv[2]), frames),
           "correction": "FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md §HBM); bytes = (2 * FETCH_SIZE "
                         "+ WRITE_SIZE) * 1024; WRITE_SIZE uncalibrated",
           "kernels": kernels}
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
