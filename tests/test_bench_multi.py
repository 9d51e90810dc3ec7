"""bench.py's N > 1 launch paths.  A plain `python bench.py --gpus N` must fan out to N ranks by itself (child processes,
started before anything touches the GPU) and print ONE line whose frame is verified against the single-GPU frame.
* one-GPU box: 2 ranks share the GPU over gloo (RCCL refuses two ranks on one device);
* a box with >= 2 GPUs: the real thing, 2 ranks over nccl (= RCCL over xGMI) -- skipped otherwise, so the first multi-GPU
  machine that runs the suite validates the RCCL gather by itself (Engine3D/Renderer.cs:1655-1680 is the split it replaces)."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tris", "100000", "--res", "1024", "--steps", "3", "--warmup", "1",
                        "--prelude-s", "0", "--no-cpu-baseline"] + list(extra),
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_plain_gpus2_fans_out_over_gloo_on_one_gpu():
    one = _bench("--gpus", "1", "--no-extras", "--verify")
    two = _bench("--gpus", "2", "--backend", "gloo")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["verify"]["full_frame_equal"]
    assert two["frame_crc"] == two["verify"]["crc"] == one["verify"]["crc"]
    m = two["multi_gpu"]
    assert m["world_size"] == 2 and m["backend"] == "gloo" and m["rccl_ranks"] is None
    assert len(m["ranks"]) == 2 and {r["rank"] for r in m["ranks"]} == {0, 1} and len({r["pid"] for r in m["ranks"]}) == 2
    assert all(r["render_ms"] > 0 for r in m["ranks"]) and m["gather_ms"] > 0


@pytest.mark.gpu
@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs: 2 ranks over nccl (RCCL)")
def test_two_ranks_over_rccl_match_single_gpu_frame():
    two = _bench("--gpus", "2")
    assert two["n_gpus"] == 2 and two["verify"]["full_frame_equal"]
    m = two["multi_gpu"]
    assert m["backend"] == "nccl" and m["rccl_ranks"] == 2 and m["distinct_devices"] == 2
