#!/usr/bin/env python3
"""Turn the rocprofv3 outputs merged under gpurun_out/ into the committed summaries under profiles/.
usage: python scripts/collect_profiles.py <tag> <stats_dir> <fetch_dir> <write_dir> <sq_dir> <bench_json>"""
import csv, collections, json, shutil, sys
tag, stats, fetch, write, sq, bench = sys.argv[1:7]
import glob
shutil.copy(glob.glob(stats + "/*_kernel_stats.csv")[0], "profiles/%s_kernel_stats.csv" % tag)
import os
for cfg, name in (("c2", "config2"), ("c4", "config4"), ("clat", "latency_n64"), ("cmid", "n16384")):          # BASELINE configs[1], configs[3], the small-batch path
    g = glob.glob(os.path.dirname(stats.rstrip("/")) + "/stats_%s/*_kernel_stats.csv" % cfg)
    if g:
        shutil.copy(g[0], "profiles/%s_%s_kernel_stats.csv" % (tag, name))
for extra in ("latency.json", "kstats.txt", "bench_line_131072.json", "keyops.json", "throughput_vs_n.json", "icache_footprint.txt"):
    src = os.path.join(os.path.dirname(stats.rstrip("/")), extra)
    if os.path.exists(src):
        shutil.copy(src, "profiles/%s_%s" % (tag, {"kstats.txt": "kernel_resources.txt"}.get(extra, extra)))
PHASE = {"k_aggregate_raw_d": "aggregate", "k_aggregate": "aggregate", "k_sig": "sig", "k_hash": "hash", "k_miller": "miller", "k_final": "final"}
shutil.copy(bench, "profiles/%s_bench_line.json" % tag)
out = {"_units": "FETCH_SIZE/WRITE_SIZE raw counter values are KB per dispatch (rocprofv3); hbm bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per the "
                 "gfx950 correction for 16-byte-per-lane streams (MI355X_MICROARCH.md, HBM section)", "raw": {}}
for name, f in (("FETCH_SIZE", fetch + "/f_counter_collection.csv"), ("WRITE_SIZE", write + "/w_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_"):
            agg[k].append(float(r["Counter_Value"]))
    out["raw"][name] = {k: sum(v) / len(v) for k, v in agg.items()}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from milagro_bls_amd import build as _build
# the build the counters belong to: scripts/profile_round.sh writes the hash of the library it ran next to its outputs (source_hash.txt); bench.py attaches the
# figures to roofline.traffic only for that build
_hf = os.path.join(os.path.dirname(stats.rstrip("/")), "source_hash.txt")
tr = {"_collected": "round %s, scripts/profile_round.sh + scripts/collect_profiles.py" % tag,
      "_source_hash": open(_hf).read().strip() if os.path.exists(_hf) else _build.source_hash()}
for k in ("k_aggregate_raw_d", "k_sig", "k_hash", "k_miller", "k_final"):
    if k not in out["raw"]["FETCH_SIZE"]:
        continue
    tr[PHASE[k]] = (2 * out["raw"]["FETCH_SIZE"][k] + out["raw"]["WRITE_SIZE"][k]) * 1024
    print(k, "%.2f GB per launch" % (tr[PHASE[k]] / 1e9))
json.dump(out, open("profiles/%s_pmc_raw.json" % tag, "w"), indent=1)
json.dump(tr, open("profiles/hbm_traffic.json", "w"), indent=1)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sq + "/s_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0]
    if k in PHASE:
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
sqo = {}
for k, v in agg.items():
    m = {c: sum(x) / len(x) for c, x in v.items()}
    sqo[k] = m
    print(k, {c: "%.0f%%" % (100 * val / m["SQ_WAVE_CYCLES"]) for c, val in m.items() if c != "SQ_WAVE_CYCLES"})
json.dump(sqo, open("profiles/%s_sq_counters.json" % tag, "w"), indent=1)
