"""bench.py's start-up contract: `python bench.py --gpus N` must start as the driver calls it -- no launcher, no
RANK / WORLD_SIZE -- and a run that cannot start prints ONE JSON line carrying "error", never a traceback.  (The run itself is
rehearsed on a GPU in tests/test_gpu_multi_device.py::test_c4_rehearsal_eight_ranks_full_batch_on_one_gpu.)"""
import json
import os
import subprocess
import sys

from conftest import ROOT

CLEAN = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def _one_line(proc):
    lines = [ln for ln in proc.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (proc.stdout[-500:], proc.stderr[-1500:])
    assert "Traceback" not in proc.stderr, proc.stderr[-1500:]
    return json.loads(lines[0])


def test_more_gpus_than_devices_is_a_json_error_line():
    import torch
    have = torch.cuda.device_count()
    proc = subprocess.run([sys.executable, "bench.py", "--gpus", str(have + 2), "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                          cwd=ROOT, env=CLEAN, timeout=300)
    d = _one_line(proc)
    assert proc.returncode == 3 and d["value"] is None and d["n_gpus"] == have + 2 and "device(s) visible" in d["error"]


def test_launcher_width_must_equal_gpus():
    env = dict(CLEAN, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    proc = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                          cwd=ROOT, env=env, timeout=300)
    d = _one_line(proc)
    assert proc.returncode == 2 and "WORLD_SIZE=1" in d["error"]


def test_self_launch_command_is_the_drivers(monkeypatch):
    """self_launch() re-runs the same arguments under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr
    127.0.0.1 --master-port P bench.py ...` -- the form the driver itself uses for N > 1."""
    import argparse
    import bench
    seen = {}

    class Done:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])

    class FakeCuda:
        @staticmethod
        def device_count():
            return 8

    class FakeTorch:
        cuda = FakeCuda

    args = argparse.Namespace(gpus=4, all_on_device=None, steps=7, warmup=3, scaling="strong", workload="c2")
    assert bench.self_launch(args, FakeTorch) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and "MASTER_PORT" not in seen["env"]


def test_stdout_digest_stays_under_8_kb():
    """The driver keeps the last 8 KB of stdout: the ONE line bench.py prints there (slim_line) must carry the contract's fields, `roofline`,
    `cpu_baseline` and every workload's digest inside that -- checked on the full result of a real default run (profiles/r06_bench_full.json)."""
    import bench
    full = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench_full.json")).read())
    line = bench.slim_line(full)
    text = json.dumps(line)
    assert len(text) < 8000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and line["cpu_baseline"]["kind"] in ("port", "reference")
    assert set(full["workloads"]) == set(line["workloads"]) and "ragged" in line and line["config"]["workload"].startswith("c2")
