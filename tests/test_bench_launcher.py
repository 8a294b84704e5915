"""bench.py's own launcher: `python bench.py --gpus N` must start N ranks (the flag used to be ignored).

Runs on CPU with AKO_BENCH_REHEARSE=1: the ranks then rendezvous over gloo and, with no GPU present, time EMPTY
steps -- what is under test is the parent / child split (the parent never touches the GPU), the rendezvous, the
barrier + max-over-ranks region and the relayed result line, not a throughput."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(args, extra_env=None, timeout=300):
    env = dict(os.environ, AKO_BENCH_REHEARSE="1", OMP_NUM_THREADS="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          timeout=timeout, env=env)


def test_gpus_flag_launches_that_many_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["ms_per_step"] >= 1.0  # empty steps sleep 1 ms: the region really ran its K steps
    assert "rehearsal" in out["data"] and out["value"] is None  # never to be mistaken for a measurement


def test_world_size_must_match_gpus():
    r = _run(["--gpus", "2", "--steps", "1"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr


def test_single_rank_is_not_relaunched():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--repeats", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 1
