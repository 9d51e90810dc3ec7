#!/usr/bin/env python3
"""Oracle-generated full-frame fixtures for the configurations whose frames are too large to commit as pixels.

Runs the CPU ORACLE (oracle/liboracle.so: the C++ restatement of the reference, reference tree 15 / 25) over whole frames of
BASELINE.json's configs C3 / C4 / C5 and writes, per 16-row strip, the CRC-32 (zlib) of the strip's pixels (int32 ARGB, row-major,
little-endian, exactly what Renderer.Render() leaves in the caller's surface) to tests/golden/frames/<name>.json.  A `-m gpu` test
(tests/test_gpu_frames.py) renders the same frame through the C ABI and compares every strip.

This is test infrastructure (it imports oracle/); nothing under softray_amd/ uses it.  It needs neither a GPU nor /root/reference.
The run is resumable: strips already in the output file are kept (`--force` starts over), and the file is rewritten after
every batch of strips.

    python scripts/make_frame_fixtures.py c3                    # 2048^2, 1 M triangles, shading + 100-sample shadows, all 128 strips
    python scripts/make_frame_fixtures.py c4                    # 4096^2, same scene, all 256 strips
    python scripts/make_frame_fixtures.py c5 --strips 100:104   # 4096^2, 10 M triangles, 4 mirror bounces, the strips named
"""
import argparse
import json
import os
import subprocess
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from helpers import make_frame, orc, random_triangles  # noqa: E402

STRIP = 16
OUT_DIR = os.path.join(ROOT, "tests", "golden", "frames")

# name -> (triangles, extent, resolution, frame keywords, (max_bounces, reflectivity))
CONFIGS = {
    "c2": dict(model="obj.3ds", res=1024, frame=dict(depth=1.0), bounce=(0, 0.0)),
    "c2_shadows": dict(model="obj.3ds", res=1024, frame=dict(depth=1.0, shadows=True), bounce=(0, 0.0)),
    "c3": dict(n=1_000_000, extent=0.05, res=2048, frame=dict(depth=1.5, shadows=True), bounce=(0, 0.0)),
    "c4": dict(n=1_000_000, extent=0.05, res=4096, frame=dict(depth=1.5, shadows=True), bounce=(0, 0.0)),
    "c5": dict(n=10_000_000, extent=0.02, res=4096, frame=dict(depth=1.5), bounce=(4, 0.5)),
    "c5_shadows": dict(n=10_000_000, extent=0.02, res=4096, frame=dict(depth=1.5, shadows=True), bounce=(0, 0.0)),
}


def scene_of(cfg):
    if "model" in cfg:
        from helpers import load_obj3ds
        return load_obj3ds(cfg["model"])
    v9, argb, _ = random_triangles(cfg["n"], 12345, space=1.0 - cfg["extent"], extent=cfg["extent"], origin=-0.5, mask_color=True)
    return v9, argb, np.array([-0.5] * 3), np.array([0.5] * 3)


def frame_of(cfg, **kw):
    f = make_frame(cfg["res"], **cfg["frame"], **kw)
    f.max_bounces, f.reflectivity = cfg["bounce"]
    return f


def strip_crcs(pixels, width, first_row, rows):
    """CRC-32 per 16-row strip of rows [first_row, first_row + rows) of a whole-frame pixel array."""
    px = np.ascontiguousarray(pixels).view(np.uint32).reshape(-1, width)
    return {(r // STRIP): zlib.crc32(px[r:r + STRIP].astype("<u4").tobytes()) & 0xFFFFFFFF for r in range(first_row, first_row + rows, STRIP)}


def parse_strips(spec, total):
    if not spec:
        return list(range(total))
    out = []
    for part in spec.split(","):
        if ":" in part:
            a, b = part.split(":")
            out.extend(range(int(a), int(b)))
        else:
            out.append(int(part))
    return sorted(set(s for s in out if 0 <= s < total))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=sorted(CONFIGS))
    ap.add_argument("--strips", default="", help="a:b,c,... strip indices (16 rows each); default: the whole frame")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--batch", type=int, default=4, help="strips per oracle call (the file is rewritten after each)")
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    cfg = CONFIGS[a.config]
    res = cfg["res"]
    total = res // STRIP
    path = os.path.join(OUT_DIR, a.config + ".json")
    os.makedirs(OUT_DIR, exist_ok=True)
    doc = None
    if os.path.exists(path) and not a.force:
        doc = json.load(open(path))
    if doc is None:
        doc = {"strips": {}}
    try:
        commit = subprocess.check_output(["git", "-C", ROOT, "log", "-1", "--format=%H", "--", "oracle/softray_oracle.cpp"], text=True).strip()
    except Exception:
        commit = "unknown"
    doc.update({
        "what": "CRC-32 (zlib) of each 16-row strip of the frame the CPU oracle renders: int32 ARGB pixels, row-major, little-endian",
        "config": a.config, "width": res, "height": res, "strip_rows": STRIP, "strips_total": total,
        "scene": ({"model": cfg["model"]} if "model" in cfg else
                  {"triangles": cfg["n"], "extent": cfg["extent"], "seed": 12345, "generator": "SpatialSubdivisionTests.cs:397-411 scaled to the unit cube (SURVEY 8d)"}),
        "frame": dict(cfg["frame"], max_bounces=cfg["bounce"][0], reflectivity=cfg["bounce"][1], pose="RendererTests yaw 135 pitch -22", shading=True,
                      shadow_samples=100 if cfg["frame"].get("shadows") else 0, tree="reference 15/25"),
        "oracle_commit": commit,
        "command": "python scripts/make_frame_fixtures.py " + a.config + "   (strips may be made in several runs: see invocations)",
    })
    inv = "python scripts/make_frame_fixtures.py " + " ".join(sys.argv[1:])
    doc.setdefault("invocations", [])
    if inv not in doc["invocations"]:
        doc["invocations"].append(inv)
    want = [s for s in parse_strips(a.strips, total) if str(s) not in doc["strips"]]
    print("%s: %d strips to render (%d already in %s)" % (a.config, len(want), len(doc["strips"]), os.path.relpath(path, ROOT)), flush=True)
    if not want:
        return
    t0 = time.time()
    v9, argb, bmin, bmax = scene_of(cfg)
    o = orc.Scene()
    o.set_triangles(v9, argb, bmin, bmax)
    assert o.build_tree() == 0
    print("scene + reference tree: %.1f s" % (time.time() - t0), flush=True)
    buf = np.zeros(res * res, dtype=np.int32)
    done = 0
    t0 = time.time()
    # consecutive strips go into one oracle call
    i = 0
    while i < len(want):
        j = i
        while j + 1 < len(want) and want[j + 1] == want[j] + 1 and j + 1 - i < a.batch:
            j += 1
        s0, s1 = want[i], want[j]
        f = frame_of(cfg, start_row=s0 * STRIP, end_row=(s1 + 1) * STRIP - 1)
        o.render(f, threads=a.threads, out=buf)
        for s, crc in strip_crcs(buf, res, s0 * STRIP, (s1 - s0 + 1) * STRIP).items():
            doc["strips"][str(s)] = crc
        done += s1 - s0 + 1
        i = j + 1
        doc["strips"] = dict(sorted(doc["strips"].items(), key=lambda kv: int(kv[0])))
        doc["coverage"] = "%d of %d strips = %.1f %% of the frame's rows" % (len(doc["strips"]), total, 100.0 * len(doc["strips"]) / total)
        tmp = path + ".tmp"
        with open(tmp, "w") as fh:
            json.dump(doc, fh, indent=0, separators=(",", ":"))
            fh.write("\n")
        os.replace(tmp, path)
        el = time.time() - t0
        print("strips %d..%d done; %d / %d, %.0f s elapsed, ~%.0f s left" % (s0, s1, done, len(want), el, el / done * (len(want) - done)), flush=True)


if __name__ == "__main__":
    main()
