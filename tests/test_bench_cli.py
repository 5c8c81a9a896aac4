"""bench.py's command line, as far as it can be checked without a GPU: the multi-GPU entry and the per-N expectation."""
import os
import subprocess
import sys

import oracle_util as ou

sys.path.insert(0, ou.ROOT)


def test_bare_multi_gpu_start_launches_its_own_ranks():
    """`python3 bench.py --gpus 2` with no rank environment (the driver's command) must start the ranks itself - a child
    `torch.distributed.run`, never an exec - and hand the ranks' verdict on.  Without GPUs both ranks say so and the exit code is theirs."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "RT_BENCH_REHEARSE")}
    r = subprocess.run([sys.executable, os.path.join(ou.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    import torch
    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0, r.stderr[-2000:]
        return
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.stderr.count("--gpus 2 but this node shows") == 2            # both ranks were started and both spoke
    assert "WORLD_SIZE=1" not in r.stderr                                   # (round 3's exit: the bare start never got this far)


def test_predicted_single_frame_is_link_bound_where_the_links_are_slower_than_the_render():
    import bench
    # the headline frame: 0.069 ms on one GPU, RGBA8 peer stores, half of the blocks sky
    p = {n: bench.predicted_single_frame(n, 3840, 2160, 0.069, 4, 0.51) for n in (2, 4, 8)}
    for n in (2, 4, 8):
        assert p[n]["bound"] == "link into rank 0"
        share = 3840 * 2160 * 4 * 0.49 / n
        assert abs(p[n]["link_ms"] - share / (bench.XGMI_GBS_PER_DIRECTION * 1e9) * 1e3) < 1e-3
        assert abs(p[n]["efficiency_vs_n1"] - 0.069 / p[n]["ms_per_step"] / n) < 2e-3
    # a frame that takes long to render and is small on the wire is render-bound and scales
    q = bench.predicted_single_frame(8, 1024, 1024, 4.0, 3, 0.0)
    assert q["bound"] == "render" and q["efficiency_vs_n1"] > 0.9
