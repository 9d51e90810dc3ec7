"""Copy the judged summaries of a gpu_profile.sh / gpu_sq_counters.sh / gpu_lane_util.sh run from gpurun_out/ into
profiles/<name>/ and refresh profiles/hbm_traffic.json.  usage: python scripts/collect_profile.py r02 profiles/r02_final [workload key]

All three scripts run `bench.py --no-split` (the frame as ONE pipeline: no two kernels share the GPU), so a kernel's row is
the launch the roofline talks about.  Launches of the STATS instantiations (one untimed counter pass) are left out."""
import collections, csv, json, os, shutil, sys

tag, dst = sys.argv[1], sys.argv[2]
workload = sys.argv[3] if len(sys.argv) > 3 else "bvh_1000000_4096_100"     # key of profiles/hbm_traffic.json: mode_tris_res_shadows[_bN]
src = "gpurun_out/prof_%s" % tag
sq = "gpurun_out/sq_%s" % tag
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_trace.json"), os.path.join(dst, "bench_under_rocprof.json"))

FAMILIES = ("k_primary", "k_cam_cones", "k_order_nodes", "k_ray_keys", "k_shaft_pkt", "k_shaft", "k_shadow_cls", "k_shadow_test", "k_shadow_wave", "k_shadow_rays",
            "k_fb_expand", "k_fb_resolve", "k_bounce_prep", "k_bounce_walk", "k_bounce_finish", "k_bounce", "k_fold", "k_resolve", "k_post_process", "k_anti_alias")
STATS_ARG = {"k_primary": 2, "k_shaft_pkt": 0, "k_shaft": 0, "k_shadow_cls": 1, "k_shadow_test": 1, "k_shadow_rays": 1, "k_shadow_wave": 1, "k_bounce": 1,
             "k_bounce_walk": 0, "k_bounce_finish": 1}


BOUNCE = ("k_bounce_prep", "k_bounce_walk", "k_bounce_finish", "k_bounce")     # a level: prepare / walk / finish (or the one-kernel form)


def family(name):
    for k in FAMILIES:
        if k in name:
            return k
    return None


def is_stats(name):
    """the STATS template argument of the instantiation is `true`"""
    head = name.split("(")[0]
    fam = family(name)
    if "<" not in head or fam not in STATS_ARG:
        return False
    args = [a.strip() for a in head[head.index("<") + 1:head.rindex(">")].split(",")]
    i = STATS_ARG[fam]
    return i < len(args) and args[i] == "true"


def ours(row):
    return "sr::" in row["Kernel_Name"]


for pmc, out in (("pmc_fetch", "pmc_fetch_kernels.csv"), ("pmc_write", "pmc_write_kernels.csv"), ("pmc_l2", "pmc_l2_kernels.csv")):
    rows = [r for r in csv.DictReader(open(os.path.join(src, pmc, "pmc_counter_collection.csv"))) if ours(r)]
    with open(os.path.join(dst, out), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)

# ---- HBM traffic per LAUNCH and kernel: (2 * FETCH_SIZE + WRITE_SIZE) * 1024 summed over the timed launches / their number
traffic = collections.defaultdict(float)
launches = collections.defaultdict(set)
for pmc, mul in (("pmc_fetch", 2.0), ("pmc_write", 1.0)):
    for r in csv.DictReader(open(os.path.join(src, pmc, "pmc_counter_collection.csv"))):
        fam = family(r["Kernel_Name"])
        if fam and not is_stats(r["Kernel_Name"]) and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            traffic[fam] += mul * float(r["Counter_Value"]) * 1024.0
            if pmc == "pmc_fetch":
                launches[fam].add(r["Dispatch_Id"])
per_launch = {k: traffic[k] / max(1, len(launches[k])) for k in traffic}
# names as bench.py's kernel table has them; k_shadow_cls runs twice per frame (rounds 1 and 2): per frame = 2 launches
table = {"k_primary": per_launch.get("k_primary"), "k_shaft": per_launch.get("k_shaft_pkt"), "k_shaft_round2": per_launch.get("k_shaft"),
         "k_shadow": (2.0 * per_launch["k_shadow_cls"]) if "k_shadow_cls" in per_launch else None,
         "k_shadow_fallback": sum(per_launch.get(k, 0.0) for k in ("k_fb_expand", "k_shadow_rays", "k_fb_resolve", "k_shadow_wave")),
         "k_bounce_per_level": (sum(per_launch.get(k, 0.0) for k in BOUNCE) or None),
         "k_bounce": (sum(traffic.get(k, 0.0) for k in BOUNCE + ("k_ray_keys", "k_fold")) / max(1, len(launches.get("k_fold", ())))) if any(k in traffic for k in BOUNCE) else None,
         "_launches_counted": {k: len(v) for k, v in launches.items()}}
hb = {}
if os.path.exists("profiles/hbm_traffic.json"):
    try:
        hb = json.load(open("profiles/hbm_traffic.json"))
    except Exception:
        hb = {}
hb[workload] = table
hb["_how"] = ("HBM bytes per LAUNCH of each kernel = sum over its timed launches of (2*FETCH_SIZE + WRITE_SIZE)*1024 / number of launches, rocprofv3 --pmc "
              "FETCH_SIZE / --pmc WRITE_SIZE in separate passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-B-per-lane loads on gfx950; "
              "bench.py --no-split: one launch per frame of k_primary / k_shaft (packet walk, round 1) / k_shaft_round2; `k_shadow` = the two launches of "
              "k_shadow_cls of a frame (rounds 1 and 2) together; `k_bounce` = all levels' k_ray_keys / k_bounce_prep / k_bounce_walk / k_bounce_finish launches + k_fold of one frame (what the K_BOUNCE event pair brackets, less the device sort's own kernels)")
hb.setdefault("_sources", {})[workload] = {"dir": dst, "command": "bash scripts/gpu_profile.sh %s ...   (bench.py --steps 2 --warmup 1 --no-split)" % tag}
json.dump(hb, open("profiles/hbm_traffic.json", "w"), indent=1)

# ---- SQ issue / wait summary per kernel (timed launches)
if os.path.isdir(sq):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    owner = {}
    for d in ("pmc_sq", "pmc_sq2", "pmc_sq3", "lanes"):
        p = os.path.join(sq, d, "pmc_counter_collection.csv") if d != "lanes" else "gpurun_out/lanes_%s/pmc/pmc_counter_collection.csv" % tag
        if not os.path.exists(p):
            continue
        seen = set()
        for r in csv.DictReader(open(p)):
            fam = family(r["Kernel_Name"])
            if not fam or is_stats(r["Kernel_Name"]):
                continue
            owner.setdefault(r["Counter_Name"], d)               # a counter collected in several passes counts once
            if owner[r["Counter_Name"]] == d:
                agg[fam][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (d, r["Dispatch_Id"])
            if d == "pmc_sq" and key not in seen:
                seen.add(key)
                agg[fam]["_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                agg[fam]["_launches"] += 1
    with open(os.path.join(dst, "sq_counters_summary.csv"), "w", newline="") as f:
        names = sorted({c for v in agg.values() for c in v})
        w = csv.writer(f)
        w.writerow(["kernel"] + names + ["valu_busy_frac_at_2.4GHz_1024_SIMDs", "active_lanes_per_valu_inst"])
        for fam, v in agg.items():
            busy = v.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / 1024.0 / 2.4e9 / (v.get("_ns", 1.0) * 1e-9) if v.get("_ns") else 0.0
            lanes = v.get("SQ_THREAD_CYCLES_VALU", 0.0) / v["SQ_INSTS_VALU"] if v.get("SQ_INSTS_VALU") else 0.0
            w.writerow([fam] + ["%.6g" % v.get(c, 0.0) for c in names] + ["%.3f" % busy, "%.1f" % lanes])
print(json.dumps(table))
