"""`python bench.py --gpus N` starts its N ranks itself (a child torch.distributed.run, before anything touches the
GPU) and passes their exit code on.  Here, with every GPU hidden, both ranks must come up and stop at the device
check with a non-zero exit."""
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_gpus_2_self_launches_two_ranks_up_to_the_device_check():
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=240)
    out = p.stdout + p.stderr
    assert p.returncode != 0, out[-2000:]
    assert "bench.py rank 0/2: needs GPU 0, this box exposes 0 GPU(s)" in out, out[-2000:]
    assert "bench.py rank 1/2: needs GPU 1, this box exposes 0 GPU(s)" in out, out[-2000:]


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE=3" in (p.stdout + p.stderr)
