#!/usr/bin/env python
"""HBM traffic per launch of the dominant kernel(s) from a tools/pmc.sh output directory, corrected as calibrated in
profiles/r02_calibration.txt: on this gfx950 / rocprofv3, FETCH_SIZE counts 128-B fabric requests at 64 B each (every read
pattern: x2), WRITE_SIZE is exact.  usage: tools/make_traffic_json.py <pmc dir> [kernel regex ...]  ->  JSON on stdout; without
a regex the kernels of the bench line's dominant form (<pmc dir>/bench_line.json, roofline.form) are taken
(commit it as profiles/rNN_traffic.json: bench.py quotes it only while the kernel sources still hash to `source_hash`)."""
import collections, csv, glob, json, os, re, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (source_hash)

d, subs = sys.argv[1], sys.argv[2:]
if subs and subs[0] == "--config":
    # the next-tier configurations: <pmc dir> --config kmeans|ransac <kernel regex>  ->  profiles/rNN_traffic_<config>.json
    config, pat = subs[1], subs[2]
    line = json.loads(open(f"{d}/bench_line.json").read())
    c = collections.defaultdict(list)
    for f in glob.glob(f"{d}/*/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if re.search(pat, r["Kernel_Name"]) and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                c[r["Counter_Name"]].append(float(r["Counter_Value"]))
    skip = 3 if len(c["FETCH_SIZE"]) > 6 else 0      # (warm-up launches)
    fb = sum(c["FETCH_SIZE"][skip:]) / max(len(c["FETCH_SIZE"][skip:]), 1) * 1024.0
    wb = sum(c["WRITE_SIZE"][skip:]) / max(len(c["WRITE_SIZE"][skip:]), 1) * 1024.0
    cfg = line.get("config", {})
    print(json.dumps({"config": config, "kernel_pattern": pat, "traffic_bytes_per_launch": 2.0 * fb + wb, "FETCH_SIZE_bytes_reported": fb, "WRITE_SIZE_bytes": wb,
                      "launches": len(c["FETCH_SIZE"]), "source_hash": bench.source_hash(bench.KERNEL_SOURCES[config]),
                      "workload": {k: cfg.get(k) for k in (("n_points", "k") if config == "kmeans" else ("n_points",))},
                      "algorithmic_bytes_per_launch": (line.get("roofline") or {}).get("algorithmic_bytes_per_launch"),
                      "method": f"separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python bench.py --config {config} --no-cpu-baseline`; HBM bytes = "
                                "2 x FETCH_SIZE + WRITE_SIZE (profiles/r02_calibration.txt)"}, indent=1))
    sys.exit(0)
line = json.loads(open(f"{d}/bench_line.json").read()) if os.path.exists(f"{d}/bench_line.json") else {}
form = (line.get("roofline") or {}).get("form")
if not subs:
    subs = {0: [r"k_search_tiled<0", r"k_search_deferred<0", r"k_iter<\d+, true"], 1: [r"k_search_tiled<[1-4]", r"k_search_deferred<[1-4]"],
            2: [r"k_warm<\d+, 1(, \w+)?>"], 3: [r"k_warm<\d+, 2(, \w+)?>"], 4: [r"k_iter<\d+, true"]}[form]
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{d}/*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(re.search(s, r["Kernel_Name"]) for s in subs) and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
per = {}
tot = 0.0
for k, c in vals.items():
    f = sum(c["FETCH_SIZE"][3:]) / max(len(c["FETCH_SIZE"][3:]), 1) * 1024.0      # KB -> B, warm-up launches dropped
    w = sum(c["WRITE_SIZE"][3:]) / max(len(c["WRITE_SIZE"][3:]), 1) * 1024.0
    per[k] = {"FETCH_SIZE_bytes_reported": f, "WRITE_SIZE_bytes": w, "hbm_bytes": 2.0 * f + w, "launches": len(c["FETCH_SIZE"])}
    tot += 2.0 * f + w
# the cold tile kernels of the same passes (bench.py: roofline_cold.plain_tile / .record_writing_tile): launch 0 of a pass is the warm-up
# run's, every launch moves the same bytes
cold = {}
for f in glob.glob(f"{d}/*/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"k_search_tiled<([1-4]), false, (true|false)>", r["Kernel_Name"])
        if m and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            cold.setdefault("record_writing_tile" if m.group(2) == "true" else "plain_tile", collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
cold_forms = {k: 2.0 * (sum(c["FETCH_SIZE"]) / max(len(c["FETCH_SIZE"]), 1)) * 1024.0 + (sum(c["WRITE_SIZE"]) / max(len(c["WRITE_SIZE"]), 1)) * 1024.0 for k, c in cold.items()}
cfg = line.get("config", {})
print(json.dumps({"traffic_bytes_per_launch": tot, "kernels": per, "cold_forms": cold_forms, "source_hash": bench.source_hash(),
                  "form": form, "kernel_patterns": list(subs),
                  "workload": {"n_target": cfg.get("n_target"), "n_source_per_gpu": cfg.get("n_source_per_gpu"),
                               "metric": "p2plane" if "point-to-plane" in cfg.get("workload", "") else ("p2p" if "point-to-point" in cfg.get("workload", "") else "combined")},
                  "method": "separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of `python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline`; "
                            "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (profiles/r02_calibration.txt: FETCH_SIZE reports half of the 128-B-line bytes for every read pattern, WRITE_SIZE is exact)"},
                 indent=1))
