"""bench.py's N > 1 path on the hardware that is there: two ranks (two processes) on the ONE GPU of the box through the host
fan-out -- rank 0's blocks in a ring in shared host memory, every rank page-locks its mapping and copies its own blocks H2D
(sdrxgpu_bank_process_pinned); gloo for the barrier; no communicator.  What RCCL cannot rehearse on one GPU (it refuses two
ranks on one device).  Checked: every block arrived intact in every rank's input ring (input_check), the JSON contract, n_gpus,
and that the line's throughput is that of 2 x 1 024 channels.  And the in-process form, bench.py --multibank."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(args, timeout=280):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       env=env, timeout=timeout)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]          # ONE line on stdout, whatever the libraries print
    return json.loads(lines[0])


@pytest.mark.timeout(300)
def test_two_ranks_on_the_one_gpu_through_the_host_fanout():
    d = _run(["--gpus", "2", "--config", "c4", "--fanout", "host", "--ranks-on-one-gpu", "--steps", "40", "--warmup", "5", "--no-cpu-baseline"])
    assert d["n_gpus"] == 2 and d["config"]["name"] == "c4" and d["config"]["fanout"] == "host"
    assert d["config"]["channels_total"] == 2048
    assert d["input_check"]["blocks_wrong"] == 0 and d["input_check"]["blocks_checked"] == 16
    assert d["value"] > 0 and abs(d["value"] * 1e6 - 2048 * 320000 * 40 / (d["ms_per_step"] * 40 / 1e3)) / (d["value"] * 1e6) < 1e-6
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("sdrx_bench_ring_")]


@pytest.mark.timeout(300)
def test_in_process_multibank_bench_line():
    d = _run(["--multibank", "2", "--channels", "600", "--steps", "30", "--warmup", "3"])
    assert d["config"]["name"] == "multibank" and d["config"]["shards"] == 2 and d["config"]["channels_total"] == 1200
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["host_pointer_entry"]["ms_per_block"] > 0
