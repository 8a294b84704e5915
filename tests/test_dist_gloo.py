"""CPU, world_size 2, gloo: the N > 1 harness of bench.py (ako_amd/dist.py).

The transform itself needs a GPU, so the per-rank "work" here is the CPU oracle on each rank's own
images; what is under test is the sharding (disjoint, complete, seeded as BASELINE configs[3] says),
the barrier + max-over-ranks timing and the checksum gather -- i.e. everything bench.py adds for N > 1.
"""
import os
import socket
import subprocess
import sys
import textwrap

import pytest

from conftest import ROOT

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import numpy as np
    import torch.distributed as dist
    from ako_amd import dist as ad
    from oracle import pyoracle as po

    rank, local_rank, world = ad.env_world()
    assert ad.init("gloo") == world == 2
    per_rank = 2
    seeds = ad.image_seeds(rank, per_rank)
    imgs = [po.gen_image(0, 96, 64, seed=s) for s in seeds]
    s = po.settings(wavelet=0, compression=2, q=16, g=16)
    sums = []
    def step():
        sums.clear()
        for im in imgs:
            blob, st = po.encode_image(s, im)
            assert st == 0
            sums.append(po.adler32(blob))
        if rank == 1:
            time.sleep(0.05)          # uneven ranks: the reported time must be the slow rank's
    elapsed = ad.timed_steps(step, steps=2, warmup=1)
    allsums = ad.gather_checksums(sums)
    strong = ad.shard_images(7, rank, world)
    gathered = [None, None]
    dist.all_gather_object(gathered, strong)
    if rank == 0:
        print(json.dumps({"elapsed": elapsed, "sums": allsums, "strong": gathered, "seeds0": seeds}))
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_gloo_harness(po, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    # slow rank decides: two timed steps with a 50 ms sleep on rank 1
    assert out["elapsed"] >= 0.1
    # sharding: weak -- rank r owns seeds base + 2r, base + 2r + 1; strong -- i mod world
    assert out["seeds0"] == [0x9E3779B9, 0x9E3779B9 + 1]
    assert out["strong"] == [[0, 2, 4, 6], [1, 3, 5]]
    # the gathered checksums are those of the four distinct images, in rank order
    s = po.settings(wavelet=0, compression=2, q=16, g=16)
    expect = []
    for i in range(4):
        blob, _ = po.encode_image(s, po.gen_image(0, 96, 64, seed=0x9E3779B9 + i))
        expect.append(po.adler32(blob))
    assert out["sums"] == [expect[0:2], expect[2:4]]
    assert len(set(expect)) == 4


def test_tile_band_sharding_reassembles_the_whole_image(po):
    """BASELINE configs[4] sharding rule on CPU: per-rank bands of tile rows, streams concatenated in rank
    order == the stream of the whole tiled image (ragged last band and edge tiles included)."""
    import numpy as np
    from ako_amd import dist as ad

    for (w, h, td, world) in [(200, 300, 64, 2), (131, 259, 32, 4), (96, 64, 16, 8), (70, 200, 64, 3)]:
        img = po.gen_image(1, w, h)
        s = po.settings(wavelet=1, compression=2, q=0, g=0, tiles=td)
        whole, st = po.encode_image(s, img)
        assert st == 0
        parts = []
        covered = 0
        for r in range(world):
            y0, rows = ad.tile_band(h, td, r, world)
            assert y0 == covered or rows == 0
            covered += rows
            if rows == 0:
                continue
            blob, st = po.encode_image(s, np.ascontiguousarray(img[y0:y0 + rows]))
            assert st == 0
            parts.append(blob[16:])
            dec, _, _ = po.decode_image(blob)
            assert np.array_equal(dec, img[y0:y0 + rows])       # lossless per band
        assert covered == h
        assert np.array_equal(np.concatenate(parts), whole[16:])
