"""Copy the judged summaries of a gpu_profile.sh / gpu_sq_counters.sh run from gpurun_out/ into profiles/<name>/ and
refresh profiles/hbm_traffic.json.  usage: python scripts/collect_profile.py r01d profiles/r01d_pipeline"""
import collections, csv, json, os, shutil, sys

tag, dst = sys.argv[1], sys.argv[2]
src = "gpurun_out/prof_%s" % tag
sq = "gpurun_out/sq_%s" % tag
os.makedirs(dst, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
shutil.copy(os.path.join(src, "bench_trace.json"), os.path.join(dst, "bench_under_rocprof.json"))


def ours(row):
    return "sr::" in row["Kernel_Name"]


def family(name):
    for k in ("k_primary", "k_cam_cones", "k_shaft_pkt", "k_shaft", "k_shadow_cls", "k_shadow_test", "k_shadow_wave", "k_shadow_rays", "k_fb_expand", "k_fb_resolve", "k_post_process", "k_anti_alias"):
        if k in name:
            return k
    return None


for pmc, out in (("pmc_fetch", "pmc_fetch_kernels.csv"), ("pmc_write", "pmc_write_kernels.csv"), ("pmc_l2", "pmc_l2_kernels.csv")):
    rows = [r for r in csv.DictReader(open(os.path.join(src, pmc, "pmc_counter_collection.csv"))) if ours(r)]
    with open(os.path.join(dst, out), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)

# ---- HBM traffic per frame and kernel family: (2 * FETCH_SIZE + WRITE_SIZE) * 1024, timed (STATS = false) launches only
frames = 6.0                                    # bench --steps 2 --warmup 1 --no-split: 3 timed-loop frames + 3 of the kernel-timing pass
traffic = collections.defaultdict(float)
for pmc, mul in (("pmc_fetch", 2.0), ("pmc_write", 1.0)):
    for r in csv.DictReader(open(os.path.join(src, pmc, "pmc_counter_collection.csv"))):
        fam = family(r["Kernel_Name"])
        if fam in ("k_primary", "k_cam_cones", "k_shaft_pkt", "k_shaft", "k_shadow_cls", "k_shadow_test", "k_shadow_wave", "k_shadow_rays", "k_fb_expand", "k_fb_resolve") and "true>" not in r["Kernel_Name"].split("(")[0]:
            traffic[fam] += mul * float(r["Counter_Value"]) * 1024.0
# k_primary also runs 3 primary-only frames at the end of bench.py: 9 launches in total
per_frame = {"k_primary": traffic["k_primary"] / 9.0, "k_shaft": (traffic["k_shaft"] + traffic["k_shaft_pkt"]) / frames,
             "k_shadow": (traffic["k_shadow_cls"] + traffic["k_shadow_test"] + traffic["k_shadow_wave"] + traffic["k_shadow_rays"] + traffic["k_fb_expand"] + traffic["k_fb_resolve"]) / frames}
hb = {"bvh_1000000_4096_100": per_frame,
      "_how": "HBM bytes per FRAME of each kernel family = sum over its launches of (2*FETCH_SIZE + WRITE_SIZE)*1024, rocprofv3 --pmc "
              "FETCH_SIZE / --pmc WRITE_SIZE in separate passes (%s), FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for "
              "16-B-per-lane loads on gfx950; one band per frame: k_primary 1, k_shaft 2, k_shadow_test 2, k_fb_expand / k_shadow_rays / k_fb_resolve / k_shadow_wave 1 launch each" % dst,
      "_command": "bash scripts/gpu_profile.sh %s   (bench.py default workload, --steps 2 --warmup 1)" % tag}
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
            if not fam or "true>" in r["Kernel_Name"].split("(")[0]:
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
print(json.dumps(per_frame))
