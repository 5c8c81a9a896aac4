"""bench.py's N>1 control flow, rehearsed on the ONE-GPU box (run with -m gpu): two ranks share GPU 0 and talk over gloo through
host memory (RT_BENCH_REHEARSE=1), started as fresh child processes by torch.distributed.run before anything touches the GPU.
Never a measurement - the JSON line says so - but every plan's set-up, pre-flight, fallback, calibration, steady-state loop,
pipelining and parity check runs exactly as on a multi-GPU node, and the peer-store plan really crosses a process boundary
(IPC handles between the two ranks).  The first real xGMI run is then not a debugging session."""
import json
import os
import socket
import subprocess
import sys

import pytest

import oracle_util as ou

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_bench(world, extra_args=(), bare=False, **env_extra):
    env = dict(os.environ, RT_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra)
    tail = [os.path.join(ou.ROOT, "bench.py"), "--gpus", str(world), "--steps", "10", "--warmup", "2", *extra_args]
    if bare:       # the way the driver starts it: no launcher, no rank environment - bench.py starts its ranks itself (a child process)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
            env.pop(k, None)
        cmd = [sys.executable, *tail]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), *tail]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # ONE JSON line, rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 10 and out["warmup"] == 2
    assert out["parity_ok"] is True and out["max_lsb_vs_reference_rows"] <= 1
    assert "REHEARSAL" in out["config"]
    assert out["config"]["steady_state_warmup"]["trains"] >= 2
    return out, r.stderr


def test_bare_invocation_starts_its_own_ranks_and_reports_both_forms():
    """`python3 bench.py --gpus 2` with no launcher and no rank environment (the driver's command): bench.py starts the two ranks as a
    child process and relays ONE JSON line whose headline is north_star's form - one 3840x2160 frame row-tiled over the ranks, whole
    on rank 0 - with the batch form beside it and a stated expectation per N."""
    out, _ = run_bench(2, bare=True)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["frames_per_step"] == 1
    assert "ONE frame" in out["config"]["workload"] and out["value"] > 0 and out["roofline"]["kernel_ms"] > 0
    assert abs(out["efficiency_vs_n1"] - out["value"] / (2 * out["n1_reference"]["mpixel_per_s"])) < 1e-3
    b = out["batch_mode"]
    assert b["scaling"] == "weak" and b["frames_per_step"] == 2 and b["parity_ok"] is True and b["value"] > 0
    assert abs(b["efficiency_vs_n1"] - b["value"] / (2 * b["n1_reference"]["mpixel_per_s"])) < 1e-3
    pred = out["predicted"]["per_n"]
    assert set(pred) == {"2", "4", "8"} and all(0 < pred[k]["efficiency_vs_n1"] <= 1.0 for k in pred)
    assert "measured_over_predicted_at_this_n" not in out["predicted"]        # a rehearsal measures nothing


def test_a_bare_multi_gpu_start_on_a_one_gpu_box_says_what_is_missing():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "RT_BENCH_REHEARSE")}
    r = subprocess.run([sys.executable, os.path.join(ou.ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    # (torch is not imported into THIS process: it brings its own HIP runtime, and a process that has loaded it first hands out IPC
    # handles the system runtime of a child cannot open - test_ipc_peer_process_renders_into_our_frame runs later in this process)
    n = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stdout=subprocess.PIPE, text=True, env=env).stdout.strip() or 0)
    if n >= 2:
        pytest.skip("a multi-GPU box: the bare start is a real run here")
    assert r.returncode != 0 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert "--gpus 2 but this node shows 1 GPU" in r.stderr


def test_exchange_plan():
    """cfg3 at N=2, exchange plan forced.  Headline: ONE frame per step, RGB24 bands gathered to rank 0 (one gather per 4 steps).
    `batch_mode`: a batch of 2 frames per step, one all-to-all per 4 steps, frame f whole on rank f."""
    out, _ = run_bench(2, RT_BENCH_P2P="0")
    assert out["scaling"] == "strong" and out["config"]["frames_per_step"] == 1
    assert out["exchange"]["collective"] == "gather to rank 0" and out["exchange"]["bytes_per_pixel_on_the_link"] == 3
    assert out["exchange"]["steps_per_collective"] == 4
    assert "plan_calibration_ms_per_step" not in out["config"]
    # the gathered bands are COMPACT: the sky's blocks - about half of the headline frame - never cross a link in this plan either
    ex = out["exchange"]
    assert ex["bands"].startswith("compact") and 0.4 < ex["sky_fraction_left_out"] < 0.6
    assert ex["bytes_sent_per_rank_per_step"] < 0.62 * ex["bytes_sent_per_rank_per_step_with_the_sky"]
    assert abs(out["predicted"]["sky_fraction_left_out"] - ex["sky_fraction_left_out"]) < 1e-3 and out["predicted"]["bytes_per_pixel_on_the_link"] == 3
    plain, _ = run_bench(2, RT_BENCH_P2P="0", RT_BENCH_NO_COMPACT="1", RT_BENCH_NO_BATCH="1")
    assert "bands" not in plain["exchange"] and plain["exchange"]["bytes_sent_per_rank_per_step"] == ex["bytes_sent_per_rank_per_step_with_the_sky"]
    b = out["batch_mode"]
    assert b["scaling"] == "weak" and b["exchange"]["collective"] == "all_to_all_single" and b["exchange"]["steps_per_collective"] == 4
    assert out["n1_reference"]["mpixel_per_s"] > 0 and b["n1_reference"]["mpixel_per_s"] > 0


def test_peer_store_plan():
    """RT_BENCH_P2P=1: every rank's kernel stores its tiles straight into the owning rank's frame buffer, opened through an IPC
    handle from the OTHER process; the pre-flight frame matches the reference's rows."""
    out, _ = run_bench(2, RT_BENCH_P2P="1")
    assert out["exchange"]["plan"].startswith("peer stores")
    assert "plan_note" not in out["config"]
    ex = out["exchange"]
    assert ex["bytes_stored_remotely_per_rank_per_step"] > 0
    # the senders leave the constant-background blocks out: about half of the headline frame never crosses a link
    assert 0.4 < ex["sky_fraction_of_this_ranks_blocks"] < 0.6
    assert ex["bytes_stored_remotely_per_rank_per_step"] < 0.62 * ex["bytes_stored_remotely_per_rank_per_step_with_the_sky"]
    assert out["batch_mode"]["exchange"]["plan"].startswith("peer stores")
    assert abs(out["predicted"]["sky_fraction_left_out"] - ex["sky_fraction_of_this_ranks_blocks"]) < 1e-3
    sent, _ = run_bench(2, RT_BENCH_P2P="1", RT_BENCH_SEND_SKY="1", RT_BENCH_NO_BATCH="1")
    assert "batch_mode" not in sent
    assert sent["exchange"]["bytes_stored_remotely_per_rank_per_step"] == ex["bytes_stored_remotely_per_rank_per_step_with_the_sky"]


def test_peer_store_failure_falls_back_to_the_exchange_plan():
    out, err = run_bench(2, RT_BENCH_P2P="1", RT_BENCH_P2P_INJECT_FAILURE="1")
    assert "injected failure on rank 1" in out["config"]["plan_note"]
    assert out["exchange"]["collective"] == "gather to rank 0"
    assert out["batch_mode"]["exchange"]["collective"] == "all_to_all_single" and "injected failure" in out["batch_mode"]["plan_note"]
    assert "peer-store plan not usable" in err


def test_auto_calibration_times_both_plans():
    out, _ = run_bench(2, RT_BENCH_NO_BATCH="1")               # the default: both plans set up, the faster one measured
    cal = out["config"]["plan_calibration_ms_per_step"]
    assert set(cal) == {"exchange", "peer_stores"} and all(v > 0 for v in cal.values())
    chosen_p2p = out["exchange"]["plan"].startswith("peer stores")
    assert chosen_p2p == (cal["peer_stores"] <= cal["exchange"])


@pytest.mark.parametrize("p2p", ["0", None])
def test_cfg4_one_frame_row_tiled_over_the_ranks(p2p):
    """BASELINE configs[3] as stated: ONE 7680x4320 frame per step, row-tiled over the ranks, whole on rank 0 after one gather
    (RT_BENCH_P2P=0) or - default - after whichever of the two plans the calibration finds faster; checked against the rows
    the reference itself rendered (tests/golden/h8_7680x4320_rows)."""
    env = {} if p2p is None else {"RT_BENCH_P2P": p2p}
    out, _ = run_bench(2, ("--config", "cfg4"), **env)
    assert out["scaling"] == "strong" and out["config"]["frames_per_step"] == 1
    assert out["config"]["workload"].startswith("cfg4: h8 scene") and "7680x4320" in out["config"]["workload"]
    assert "h8_7680x4320_rows" in out["parity_checked_against"]
    if p2p == "0":
        assert out["exchange"]["collective"] == "gather to rank 0"
    else:
        assert set(out["config"]["plan_calibration_ms_per_step"]) == {"exchange", "peer_stores"}


def test_cfg5_scene_single_frame_mode_at_reduced_size():
    """BASELINE configs[4]'s scene and mode (one supersampled frame per step, 64 spheres, depth 5, whole on rank 0) at 2048x2048
    so that the rehearsal stays short; a frame size without golden rows is checked against the C restatement's rows."""
    out, _ = run_bench(3, ("--config", "cfg5", "--width", "2048", "--height", "2048"), RT_BENCH_P2P="auto")
    assert out["n_gpus"] == 3 and out["scaling"] == "strong"
    assert "oracle/rt_oracle.c rows" in out["parity_checked_against"]
    assert "supersample 2" in out["config"]["workload"]
