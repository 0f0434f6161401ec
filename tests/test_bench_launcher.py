"""CPU: `python bench.py --gpus N` must start its own ranks (one process per GPU through
torch.distributed.run) before the parent touches a GPU, so that the plain command works on a
multi-GPU node."""
import importlib.util
import os
import subprocess
import sys

from cases import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_launch_command_shape():
    b = _bench()
    cmd = b.launch_command(8, ["--gpus", "8", "--steps", "64"], 29999)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29999"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "64"] and cmd[-5].endswith("bench.py")
    assert b.round_up(20, 8) == 24 and b.round_up(16, 8) == 16 and b.round_up(0, 8) == 0


def test_plain_gpus_2_starts_two_ranks_without_touching_a_gpu_in_the_parent():
    """On this GPU-less machine every rank stops with the "needs a GPU" message; what matters is that the
    parent got as far as starting both ranks and handed their failure back as its exit code."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300, env=env)
    import torch
    if torch.cuda.is_available():
        return   # on a GPU box this is a real two-rank run (or a refusal for lack of a second GPU); nothing to assert here
    assert out.returncode != 0
    assert out.stderr.count("bench.py needs a GPU") >= 2, out.stderr[-2000:]
