"""bench.py's launch contract: `python bench.py --gpus N` (no launcher around it, exactly how the driver's SCALE runs start it)
must itself bring up N ranks -- before anything touches the GPU -- and report n_gpus = N."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert lines, text[-3000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("gpus", [1, 2, 4])
def test_bare_invocation_spawns_ranks(gpus):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--probe-ranks"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = _last_json(res.stdout)
    assert out["n_gpus"] == gpus and out["ranks_counted"] == gpus


@pytest.mark.parametrize("gpus,batch", [(2, 1024), (4, 1024), (4, 10), (8, 1024)])
def test_strong_scaling_shards_the_job_total(gpus, batch):
    """`--scaling strong` = BASELINE configs[3] as written (1024 transforms in the whole job, 1024/G per GPU); the
    default stays weak (--batch on every GPU), which is what the driver's contract asks for."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    for scaling in ("strong", "weak"):
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--probe-ranks", "--batch", str(batch)] +
                             (["--scaling", "strong"] if scaling == "strong" else []),
                             capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        out = _last_json(res.stdout)
        per = [r["transforms_per_step"] // 2 for r in sorted(out["ranks"], key=lambda r: r["rank"])]
        assert out["scaling"] == scaling and len(per) == gpus
        if scaling == "strong":
            assert sum(per) == batch == out["batch_total"] and max(per) - min(per) <= 1 and per == sorted(per, reverse=True)
        else:
            assert per == [batch] * gpus and out["batch_total"] == batch * gpus


def test_under_a_launcher_no_second_spawn():
    # the contract's other form: the driver starts torch.distributed.run itself; bench.py must then NOT spawn again
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--probe-ranks"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, OMP_NUM_THREADS="1"), cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    outs = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(outs) == 1 and json.loads(outs[0])["ranks_counted"] == 2


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_rehearsal():
    """world = 2 through the real bench body on the one GPU of the box (gloo carries the collectives, both ranks share cuda:0):
    the batch workload and the slab workload (one all-to-all per transform)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TOYNI_BENCH_BACKEND"] = "gloo"
    for extra in (["--batch", "8"], ["--workload", "slab", "--log-n", "22"], ["--workload", "fourstep", "--log-n", "20"]):
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                              "--no-extras", "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
        out = _last_json(res.stdout)
        assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2 and out["value"] > 0
        if "--workload" in extra:    # the forward half was gathered and compared with the single-device transform
            assert out["exchange_verified"] is True
            # the one-transform lines carry a roofline object (HBM + exchange, per-phase times from events)
            roof = out["roofline"]
            assert roof["bound"] == "hbm" and 0 < roof["frac"] < 1 and roof["algorithmic_bytes_per_step"] == 16 * (1 << int(extra[-1]))
            assert roof["phases_ms_per_step"]["exchange"] > 0 and roof["exchange"]["bytes_sent_per_rank_per_transform"] > 0
            assert ("slab_pass" in roof["phases_ms_per_step"]) == (extra[1] == "slab")
        if "--batch" in extra:       # the per-rank table: which device every rank used and its own step time
            assert [r["rank"] for r in out["ranks"]] == [0, 1] and all(r["ms_per_step"] > 0 and r["transforms_per_step"] == 16 for r in out["ranks"])
            assert "rank  device  pci" in res.stderr
    # strong scaling through the real body: 9 transforms over 2 ranks = 5 + 4
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extras",
                          "--no-cpu-baseline", "--batch", "9", "--scaling", "strong"], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = _last_json(res.stdout)
    assert out["scaling"] == "strong" and [r["transforms_per_step"] for r in out["ranks"]] == [10, 8] and out["config"]["batch_total"] == 9


@pytest.mark.gpu
def test_single_process_slab_line_carries_roofline_and_baseline():
    """The configs[4] line as an 8-GPU node would print it, rehearsed with 8 lanes on the one device: `roofline` (algorithmic bytes vs
    the HBM of the devices in use, exchange bytes vs the xGMI links) and `cpu_baseline` (the oracle at n, or extrapolated and labelled)."""
    env = dict(os.environ, TOYNI_BENCH_LANES_ON_ONE_DEVICE="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "slab-sp", "--gpus", "8", "--log-n", "25", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    out = _last_json(res.stdout)
    assert out["exchange_verified"] is True and out["lanes"] == 8
    roof, base = out["roofline"], out["cpu_baseline"]
    assert roof["bound"] == "hbm" and roof["peak"] == 8000.0 and 0 < roof["frac"] < 1
    ph = roof["phases_ms_per_step"]                      # lane 0's stages from the measurement build's events
    assert set(ph) == {"slab_pass", "exchange", "relayout", "row_transforms"} and all(v > 0 for v in ph.values())
    assert sum(ph.values()) < 3 * out["ms_per_step"]     # one lane's stages: the same order of magnitude as a step (eight lanes share the device)
    assert roof["exchange"]["bytes_sent_per_rank_per_transform"] == 4 * (1 << 25) / 8 * 7 / 8 and "ONE device" in roof["exchange"]["note"]
    assert base["kind"] == "port" and base["cores"] == 1 and base["value"] > 0 and base["extrapolated"] is True and "EXTRAPOLATED" in base["sample"]
