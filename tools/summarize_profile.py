#!/usr/bin/env python3
"""Condense what tools/gpu_profile.sh left under gpurun_out/<tag>/ into the files kept under profiles/:

    python tools/summarize_profile.py gpurun_out/<tag> [--round 1] [--name bench] [--no-copy]

  profiles/rNN_<name>_kernel_stats.csv       rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/rNN_<name>_under_rocprof.json     bench.py's JSON line from that same run
  profiles/rNN_pmc_{fetch,write}_size_by_kernel.tsv   kernel, launches, sum KB, average KB per launch
  profiles/pmc_latest.json                   HBM bytes per launch of the kernel bench.py names as dominant
                                             (FETCH_SIZE doubled: gfx950 correction, MI355X_MICROARCH.md HBM section)
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
csv.field_size_limit(1 << 30)


def find(d, pat):
    hits = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return hits[0] if hits else None


def pmc_by_kernel(path, counter):
    """counter_collection.csv -> {kernel: [launches, sum]}; a dispatch may span several rows (one per counter/dimension)."""
    per_dispatch = collections.defaultdict(float)
    name_of = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            key = (row.get("Process_Id", ""), row.get("Dispatch_Id") or row.get("Correlation_Id"))
            per_dispatch[key] += float(row["Counter_Value"])
            name_of[key] = row["Kernel_Name"]
    out = collections.defaultdict(lambda: [0, 0.0])
    for key, v in per_dispatch.items():
        o = out[name_of[key]]
        o[0] += 1
        o[1] += v
    return out


def norm(name):
    return name.replace(" ", "").replace("void", "").replace("(anonymousnamespace)::", "")


def residency(trace_csv, fam, kernels):
    ev = []; rows = []
    with open(trace_csv, newline="") as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), norm(r["Kernel_Name"])))
    # the replays only: from the first launch of the first replayed step's first kernel (the capture's eager passes come before)
    rows.sort()
    firsts = [s for s, e, n in rows if "image_to_nhwc4" in n]
    opt = [e for s, e, n in rows if "rmsprop_kernel" in n]
    if len(firsts) < 3 or not opt:
        return None
    t0, t1 = firsts[-3], max(opt)                 # the last three steps of the run are replays (bench.py --steps 3)
    rows = [r for r in rows if r[0] >= t0 and r[1] <= t1]
    for i, (s, e, n) in enumerate(rows):
        ev.append((s, 1, i)); ev.append((e, 0, i))
    ev.sort()
    active = set(); last = ev[0][0]
    conc = {1: 0.0, 2: 0.0, 3: 0.0}
    shared = [0.0] * len(rows)
    for t, kind, i in ev:
        dt = t - last
        if dt > 0 and active:
            conc[min(len(active), 3)] += dt
            if len(active) > 1:
                for j in active:
                    shared[j] += dt
        last = t
        if kind:
            active.add(i)
        else:
            active.discard(i)
    steps = 3
    res = {"steps": steps, "ms_per_step_with_1_2_3plus_kernels_resident": [round(conc[k] / steps / 1e6, 2) for k in (1, 2, 3)], "families": {}}
    for k_, pats in fam.items():
        tot = sh = 0.0; solo_n = 0; solo_t = 0.0
        for (s, e, n), sh_ in zip(rows, shared):
            if any(p_ in n for p_ in pats):
                tot += e - s; sh += sh_
                if sh_ < 0.05 * (e - s):
                    solo_n += 1; solo_t += e - s
        if tot:
            res["families"][k_] = {"shared_frac_of_time": round(sh / tot, 3), "launches_alone": solo_n,
                                   "avg_launch_ms_alone": round(solo_t / solo_n / 1e6, 4) if solo_n else None}
            if k_ in kernels:
                kernels[k_]["shared_frac_of_time"] = round(sh / tot, 3)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--round", type=int, default=1)
    ap.add_argument("--name", default="bench")
    ap.add_argument("--no-copy", action="store_true", help="print only, leave profiles/ alone")
    a = ap.parse_args()
    prof = os.path.join(ROOT, "profiles")
    tag = f"r{a.round:02d}"
    stats = find(os.path.join(a.dir, "stats"), "*kernel_stats.csv")
    bench_json = os.path.join(a.dir, "bench_under_rocprof.json")
    dominant = None
    match = None
    eager_json = os.path.join(a.dir, "bench_eager_under_rocprof.json")
    src_hash = None
    for cand in (bench_json, eager_json, os.path.join(a.dir, "bench.json")):      # (the graph-replay run carries no live roofline)
        if os.path.exists(cand) and os.path.getsize(cand):
            with open(cand) as f:
                line = [l for l in f if l.startswith("{")][-1]
            rec_ = json.loads(line)
            src_hash = src_hash or rec_.get("src")      # the sources the profiled runs used: bench.py quotes a profile only on the same ones
            rf = rec_.get("roofline")
            if rf and rf.get("kernel") and dominant is None:
                dominant = rf["kernel"]
                match = rf.get("rocprof_match")
    full_json = os.path.join(a.dir, "bench_full.json")
    if match is None and os.path.exists(full_json):      # (the compact line no longer carries the match strings)
        with open(full_json) as f:
            match = ((json.load(f).get("bench_line") or {}).get("roofline") or {}).get("rocprof_match")
    if stats:
        with open(stats, newline="") as f:
            rows = list(csv.DictReader(f))
        print(f"{'kernel':90s} {'calls':>6s} {'avg us':>10s} {'%':>6s}")
        for r in rows[:14]:
            print(f"{r['Name'][:90]:90s} {r['Calls']:>6s} {float(r['AverageNs']) / 1e3:10.1f} {r['Percentage']:>6s}")
        if not a.no_copy:
            shutil.copy(stats, os.path.join(prof, f"{tag}_{a.name}_kernel_stats.csv"))
            if os.path.exists(bench_json):
                shutil.copy(bench_json, os.path.join(prof, f"{tag}_{a.name}_under_rocprof.json"))
            est = find(os.path.join(a.dir, "stats_eager"), "*kernel_stats.csv")
            if est:     # the eager, side-streams-off run: the regime of bench.py's live per-launch event pairs
                shutil.copy(est, os.path.join(prof, f"{tag}_{a.name}_eager_kernel_stats.csv"))
                if os.path.exists(eager_json):
                    shutil.copy(eager_json, os.path.join(prof, f"{tag}_{a.name}_eager_under_rocprof.json"))
            for extra in ("bench.json", "bench_full.json", "bench_bf16s.json", "bench_full_bf16s.json", "bench_fp8s.json", "bench_full_fp8s.json",
                          "bench_fp8s_clips32.json", "bench_bf16s_clips32.json", "precision_criterion.json"):
                if os.path.exists(os.path.join(a.dir, extra)):
                    shutil.copy(os.path.join(a.dir, extra), os.path.join(prof, f"{tag}_{extra}"))
            if os.path.exists(os.path.join(a.dir, "precision_criterion.json")):      # (bench.py quotes the modes' criterion from this copy)
                shutil.copy(os.path.join(a.dir, "precision_criterion.json"), os.path.join(prof, "precision_criterion_latest.json"))
            b16 = find(os.path.join(a.dir, "stats_bf16s"), "*kernel_stats.csv")      # the bf16-storage mode's replays under the tracer
            if b16:
                shutil.copy(b16, os.path.join(prof, f"{tag}_{a.name}_bf16s_kernel_stats.csv"))
                if os.path.exists(os.path.join(a.dir, "bench_bf16s_under_rocprof.json")):
                    shutil.copy(os.path.join(a.dir, "bench_bf16s_under_rocprof.json"), os.path.join(prof, f"{tag}_{a.name}_bf16s_under_rocprof.json"))
            f8 = find(os.path.join(a.dir, "stats_fp8s"), "*kernel_stats.csv")       # the fp8-storage mode's replays under the tracer
            if f8:
                shutil.copy(f8, os.path.join(prof, f"{tag}_{a.name}_fp8s_kernel_stats.csv"))
    # in-step launch durations of the replayed graph (real stream concurrency), per kernel family: bench.py cannot time launches
    # inside a hipGraph (events recorded in a captured graph cannot be read: tools/graph_event_probe.hip), so it quotes these beside
    # its live event-pair numbers
    if stats and not a.no_copy:
        fam = {"conv3": ["conv3_kernel<4,2,4,2,", "conv3_kernel<2,2,4,2,", "conv3x_kernel<"], "conv3x": ["conv3x_kernel<"], "conv3<4,2,4>": ["conv3_kernel<4,2,4,2,"], "conv3<2,2,4>": ["conv3_kernel<2,2,4,2,"],
               "wgrad3": ["wgrad3_kernel<2,false>"], "conv1": ["conv1_kernel<"], "wgrad<128,128>": ["wgrad_kernel<128,128,16,true,0,2,false>", "wgrad1x_kernel<"],
               "igemm<128,128> NT": ["igemm_kernel<128,128,2,2,0,false,16,true,0,2,"], "scale_act": ["scale_act_kernel", "scale_act_pc_kernel"],
               "bn_act_bwd_apply": ["bn_act_bwd_apply_kernel", "bn_act_bwd_apply_pc_kernel"], "channel_partials": ["channel_partials_kernel"],
               "l2norm_score_fwd": ["l2norm_score_fwd_kernel"], "dgrad2": ["dgrad2_kernel<"], "nconv1": ["nconv1_kernel<"],
               "wgrad9": ["wgrad9_kernel<"], "stem_wgrad_bn": ["stem_wgrad_bn_kernel"]}
        out = {"round": a.round, "src_hash": src_hash, "source": f"profiles/{tag}_{a.name}_kernel_stats.csv", "command": "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 "
               "--warmup 1 --no-cpu-baseline --alt-steps 0 --profile-steps 0 (graph replays only)", "kernels": {}}
        for k_, pats in fam.items():
            calls = 0; tot = 0.0
            for r in rows:
                if any(p_ in norm(r["Name"]) for p_ in pats):
                    calls += int(r["Calls"]); tot += float(r["TotalDurationNs"])
            if calls:
                out["kernels"][k_] = {"calls": calls, "avg_launch_ms": tot / calls / 1e6}
        # who shares the chip: from the kernel trace of the same run, per family the part of its time during which a kernel of
        # another graph queue was resident too (a launch beside the weight-gradient queue is longer than the same launch alone —
        # the step is shorter for it), and the replays' time with one / two / three+ kernels resident
        trace = find(os.path.join(a.dir, "stats"), "*kernel_trace.csv")
        if trace:
            out["residency"] = residency(trace, fam, out["kernels"])
        with open(os.path.join(prof, "in_step_latest.json"), "w") as f:
            json.dump(out, f, indent=1)
    tables = {}
    for which, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        cc = find(os.path.join(a.dir, f"pmc_{which}"), "*counter_collection.csv")
        if not cc:
            continue
        t = pmc_by_kernel(cc, counter)
        tables[which] = t
        lines = [f"{k[:90]}\t{n}\t{s:.3f}\t{s / n:.4f}" for k, (n, s) in sorted(t.items(), key=lambda kv: -kv[1][1])]
        if not a.no_copy:
            with open(os.path.join(prof, f"{tag}_pmc_{which}_size_by_kernel.tsv"), "w") as f:
                f.write("\n".join(lines) + "\n")
        print(f"-- {counter} (KB): kernel, launches, sum, avg/launch")
        print("\n".join(lines[:6]))
    if dominant and len(tables) == 2:
        wants = [w.replace(" ", "") for w in match] if match else [norm(dominant).rstrip(">")]      # rocprof prints defaulted template arguments too
        hit = [k for k in tables["fetch"] if any(w in norm(k) for w in wants)]
        if hit:
            k = " + ".join(h[:80] for h in hit)
            nf = sum(tables["fetch"][h][0] for h in hit); sf = sum(tables["fetch"][h][1] for h in hit)
            nw = sum(tables["write"].get(h, [0, 0.0])[0] for h in hit); sw = sum(tables["write"].get(h, [0, 0.0])[1] for h in hit)
            fetch_kb, write_kb = sf / nf, sw / max(nw, 1)
            rec = {"round": a.round, "src_hash": src_hash, "kernel": dominant, "rocprof_name": k, "launches_in_pass": nf,
                   "FETCH_SIZE_KB_avg_per_launch": fetch_kb, "WRITE_SIZE_KB_avg_per_launch": write_kb,
                   "gfx950_correction": "FETCH_SIZE doubled (16-B/lane coalesced reads are tallied at half their bytes, "
                                        "MI355X_MICROARCH.md HBM section); WRITE_SIZE exact",
                   "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024,
                   "command": "tools/gpu_profile.sh: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 "
                              "bench.py --steps 1 --warmup 0 --no-cpu-baseline --alt-steps 0 --profile-steps 1 (second pass: --pmc WRITE_SIZE)",
                   "note": "Infinity-Cache hits are counted in FETCH_SIZE, so this is an upper bound on HBM bytes"}
            print(json.dumps(rec, indent=1))
            if not a.no_copy:
                with open(os.path.join(prof, "pmc_latest.json"), "w") as f:
                    json.dump(rec, f, indent=1)
        else:
            print("dominant kernel", dominant, "not found in the PMC pass", file=sys.stderr)


if __name__ == "__main__":
    main()
